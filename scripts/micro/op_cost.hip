// Developer microbenchmark for gfx950: issue cost (cycles per wave-instruction per SIMD at 8 waves/SIMD) of the VALU
// instructions that appear in the EM kernels next to v_fma_f64.  Each kernel runs 16 independent copies of one
// instruction per loop iteration (inline asm, so the compiler keeps exactly that instruction).
// Build: hipcc --offload-arch=gfx950 -O3 op_cost.hip -o op_cost
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define OP_KERNEL(NAME, DECL, BODY, SINK)                                                      \
	__global__ void NAME(double *out, int iters)                                           \
	{                                                                                      \
		DECL;                                                                          \
		for (int it = 0; it < iters; it++) {                                           \
			_Pragma("unroll") for (int i = 0; i < 16; i++) { BODY; }               \
		}                                                                              \
		double s = 0;                                                                  \
		for (int i = 0; i < 16; i++) s += SINK;                                        \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;                                \
	}

OP_KERNEL(k_fma, double x[16]; for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i,
	  asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x[i]) : "v"(0.999), "v"(0.001)), x[i])
OP_KERNEL(k_mul, double x[16]; for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i,
	  asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(0.999)), x[i])
OP_KERNEL(k_add, double x[16]; for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i,
	  asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(0.999)), x[i])
OP_KERNEL(k_cvt, double x[16]; unsigned u[16]; for (int i = 0; i < 16; i++) { u[i] = threadIdx.x + i; x[i] = 0; },
	  asm volatile("v_cvt_f64_u32_e32 %0, %1" : "=v"(x[i]) : "v"(u[i])), x[i])
OP_KERNEL(k_rcp, double x[16]; for (int i = 0; i < 16; i++) x[i] = 1.0 + threadIdx.x * 1e-3 + i,
	  asm volatile("v_rcp_f64_e32 %0, %0" : "+v"(x[i])), x[i])
OP_KERNEL(k_frexp_mant, double x[16]; for (int i = 0; i < 16; i++) x[i] = 1.0 + threadIdx.x * 1e-3 + i,
	  asm volatile("v_frexp_mant_f64_e32 %0, %0" : "+v"(x[i])), x[i])
OP_KERNEL(k_ldexp, double x[16]; for (int i = 0; i < 16; i++) x[i] = 1.0 + threadIdx.x * 1e-3 + i,
	  asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(1)), x[i])
OP_KERNEL(k_bfe, unsigned u[16]; for (int i = 0; i < 16; i++) u[i] = threadIdx.x * 2654435761u + i,
	  asm volatile("v_bfe_u32 %0, %0, 2, 30" : "+v"(u[i])), (double)u[i])
OP_KERNEL(k_and, unsigned u[16]; for (int i = 0; i < 16; i++) u[i] = threadIdx.x * 2654435761u + i,
	  asm volatile("v_and_b32_e32 %0, 0x7fffffff, %0" : "+v"(u[i])), (double)u[i])
OP_KERNEL(k_cndmask, unsigned u[16]; for (int i = 0; i < 16; i++) u[i] = threadIdx.x * 2654435761u + i,
	  asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(7u) : "vcc"), (double)u[i])
OP_KERNEL(k_mul_lo, unsigned u[16]; for (int i = 0; i < 16; i++) u[i] = threadIdx.x * 2654435761u + i,
	  asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(2654435761u)), (double)u[i])
OP_KERNEL(k_mul_hi, unsigned u[16]; for (int i = 0; i < 16; i++) u[i] = threadIdx.x * 2654435761u + i,
	  asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(u[i]) : "v"(2654435761u)), (double)u[i])
OP_KERNEL(k_cmp, double x[16]; for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i,
	  asm volatile("v_cmp_lt_f64_e32 vcc, %0, %1" : : "v"(x[i]), "v"(0.5) : "vcc"), x[i])
OP_KERNEL(k_fma_sgpr, double x[16]; for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i,
	  asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(x[i]) : "s"(0.999), "v"(0.001)), x[i])

template <typename F> static double time_ms(F f)
{
	hipEvent_t a, b;
	hipEventCreate(&a); hipEventCreate(&b);
	f();
	hipDeviceSynchronize();
	hipEventRecord(a);
	f();
	hipEventRecord(b);
	hipEventSynchronize(b);
	float ms;
	hipEventElapsedTime(&ms, a, b);
	return ms;
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	double *out;
	CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
	const int iters = 20000;
	printf("device: %s, %d CUs; cycles per wave-instruction per SIMD at 2.4 GHz (8 / 4 waves per SIMD)\n", prop.gcnArchName, cus);
#define RUN(NAME)                                                                                              \
	{                                                                                                      \
		double c[2];                                                                                   \
		int w = 0;                                                                                     \
		for (int wpc : {32, 16}) {                                                                     \
			dim3 grid(cus * wpc / 4), block(256);                                                  \
			double ms = time_ms([&] { hipLaunchKernelGGL(NAME, grid, block, 0, 0, out, iters); }); \
			c[w++] = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wpc / 4);                           \
		}                                                                                              \
		printf("%-14s %6.2f %6.2f\n", #NAME, c[0], c[1]);                                              \
	}
	RUN(k_fma) RUN(k_fma_sgpr) RUN(k_mul) RUN(k_add) RUN(k_cvt) RUN(k_rcp) RUN(k_frexp_mant) RUN(k_ldexp) RUN(k_cmp)
	RUN(k_bfe) RUN(k_and) RUN(k_cndmask) RUN(k_mul_lo) RUN(k_mul_hi)
	return 0;
}
