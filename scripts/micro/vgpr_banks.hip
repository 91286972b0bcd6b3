// Developer microbenchmark for gfx950: does the issue cost of v_fma_f64 depend on WHICH registers its three 64-bit sources sit in?
// Explicit physical registers (asm clobbers), 16 independent accumulators per iteration so that dependency never binds; variants
// differ only in the register numbers of the two multiplicands relative to the accumulator.  4 and 8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 vgpr_banks.hip -o vgpr_banks
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// accumulators v[32:63] (16 pairs); multiplicand A at v[A:A+1], B at v[B:B+1] (fixed for all 16 FMAs), or A stepping with the accumulator
#define FMA16(A, B) \
	"v_fma_f64 v[32:33], " #A ", " #B ", v[32:33]\n v_fma_f64 v[34:35], " #A ", " #B ", v[34:35]\n" \
	"v_fma_f64 v[36:37], " #A ", " #B ", v[36:37]\n v_fma_f64 v[38:39], " #A ", " #B ", v[38:39]\n" \
	"v_fma_f64 v[40:41], " #A ", " #B ", v[40:41]\n v_fma_f64 v[42:43], " #A ", " #B ", v[42:43]\n" \
	"v_fma_f64 v[44:45], " #A ", " #B ", v[44:45]\n v_fma_f64 v[46:47], " #A ", " #B ", v[46:47]\n" \
	"v_fma_f64 v[48:49], " #A ", " #B ", v[48:49]\n v_fma_f64 v[50:51], " #A ", " #B ", v[50:51]\n" \
	"v_fma_f64 v[52:53], " #A ", " #B ", v[52:53]\n v_fma_f64 v[54:55], " #A ", " #B ", v[54:55]\n" \
	"v_fma_f64 v[56:57], " #A ", " #B ", v[56:57]\n v_fma_f64 v[58:59], " #A ", " #B ", v[58:59]\n" \
	"v_fma_f64 v[60:61], " #A ", " #B ", v[60:61]\n v_fma_f64 v[62:63], " #A ", " #B ", v[62:63]\n"
#define CLOB "v32","v33","v34","v35","v36","v37","v38","v39","v40","v41","v42","v43","v44","v45","v46","v47","v48","v49","v50","v51","v52","v53","v54","v55","v56","v57","v58","v59","v60","v61","v62","v63","v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79"

#define KERNEL(NAME, BODY) \
	__global__ __launch_bounds__(256) void NAME(double *out, int iters) \
	{ \
		asm volatile("v_mov_b32 v64, 0\n v_mov_b32 v65, 0x3fefffff\n v_mov_b32 v66, 0\n v_mov_b32 v67, 0x3f500000\n" \
			     "v_mov_b32 v68, 0\n v_mov_b32 v69, 0x3fefffff\n v_mov_b32 v70, 0\n v_mov_b32 v71, 0x3f500000\n" \
			     "v_mov_b32 v72, 0\n v_mov_b32 v73, 0x3fefffff\n v_mov_b32 v74, 0\n v_mov_b32 v75, 0x3f500000\n" \
			     "s_mov_b32 s20, 0\n s_mov_b32 s21, 0x3fefffff\n" ::: CLOB, "s20", "s21"); \
		for (int i = 32; i < 64; i += 2) asm volatile("" ::: "memory"); \
		for (int it = 0; it < iters; it++) asm volatile(BODY ::: CLOB); \
		double s; \
		asm volatile("v_add_f64 %0, v[32:33], v[62:63]" : "=v"(s) :: CLOB); \
		out[blockIdx.x * blockDim.x + threadIdx.x] = s; \
	}

// source pairs: 64:65 (pair index 32 -> even "bank pair"), 66:67 (odd), 68:69 (even), 70:71 (odd), 72..75
KERNEL(k_even_even, FMA16(v[64:65], v[68:69]))     // both multiplicands in even pairs (reg/2 even); accumulators alternate even/odd pairs
KERNEL(k_even_odd, FMA16(v[64:65], v[66:67]))      // one even, one odd
KERNEL(k_same, FMA16(v[64:65], v[64:65]))          // the same register twice
KERNEL(k_sgpr, FMA16(s[20:21], v[66:67]))       // one multiplicand in scalar registers
// accumulators restricted to even pairs / odd pairs vs the sources: built from two half-sets
KERNEL(k_acc_even_src_odd_odd, "v_fma_f64 v[32:33], v[66:67], v[70:71], v[32:33]\n v_fma_f64 v[36:37], v[66:67], v[70:71], v[36:37]\n v_fma_f64 v[40:41], v[66:67], v[70:71], v[40:41]\n v_fma_f64 v[44:45], v[66:67], v[70:71], v[44:45]\n v_fma_f64 v[48:49], v[66:67], v[70:71], v[48:49]\n v_fma_f64 v[52:53], v[66:67], v[70:71], v[52:53]\n v_fma_f64 v[56:57], v[66:67], v[70:71], v[56:57]\n v_fma_f64 v[60:61], v[66:67], v[70:71], v[60:61]\n" "v_fma_f64 v[32:33], v[66:67], v[70:71], v[32:33]\n v_fma_f64 v[36:37], v[66:67], v[70:71], v[36:37]\n v_fma_f64 v[40:41], v[66:67], v[70:71], v[40:41]\n v_fma_f64 v[44:45], v[66:67], v[70:71], v[44:45]\n v_fma_f64 v[48:49], v[66:67], v[70:71], v[48:49]\n v_fma_f64 v[52:53], v[66:67], v[70:71], v[52:53]\n v_fma_f64 v[56:57], v[66:67], v[70:71], v[56:57]\n v_fma_f64 v[60:61], v[66:67], v[70:71], v[60:61]\n")
KERNEL(k_acc_even_src_even_even, "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[36:37], v[64:65], v[68:69], v[36:37]\n v_fma_f64 v[40:41], v[64:65], v[68:69], v[40:41]\n v_fma_f64 v[44:45], v[64:65], v[68:69], v[44:45]\n v_fma_f64 v[48:49], v[64:65], v[68:69], v[48:49]\n v_fma_f64 v[52:53], v[64:65], v[68:69], v[52:53]\n v_fma_f64 v[56:57], v[64:65], v[68:69], v[56:57]\n v_fma_f64 v[60:61], v[64:65], v[68:69], v[60:61]\n" "v_fma_f64 v[32:33], v[64:65], v[68:69], v[32:33]\n v_fma_f64 v[36:37], v[64:65], v[68:69], v[36:37]\n v_fma_f64 v[40:41], v[64:65], v[68:69], v[40:41]\n v_fma_f64 v[44:45], v[64:65], v[68:69], v[44:45]\n v_fma_f64 v[48:49], v[64:65], v[68:69], v[48:49]\n v_fma_f64 v[52:53], v[64:65], v[68:69], v[52:53]\n v_fma_f64 v[56:57], v[64:65], v[68:69], v[56:57]\n v_fma_f64 v[60:61], v[64:65], v[68:69], v[60:61]\n")
KERNEL(k_acc_even_src_even_odd, "v_fma_f64 v[32:33], v[64:65], v[66:67], v[32:33]\n v_fma_f64 v[36:37], v[64:65], v[66:67], v[36:37]\n v_fma_f64 v[40:41], v[64:65], v[66:67], v[40:41]\n v_fma_f64 v[44:45], v[64:65], v[66:67], v[44:45]\n v_fma_f64 v[48:49], v[64:65], v[66:67], v[48:49]\n v_fma_f64 v[52:53], v[64:65], v[66:67], v[52:53]\n v_fma_f64 v[56:57], v[64:65], v[66:67], v[56:57]\n v_fma_f64 v[60:61], v[64:65], v[66:67], v[60:61]\n" "v_fma_f64 v[32:33], v[64:65], v[66:67], v[32:33]\n v_fma_f64 v[36:37], v[64:65], v[66:67], v[36:37]\n v_fma_f64 v[40:41], v[64:65], v[66:67], v[40:41]\n v_fma_f64 v[44:45], v[64:65], v[66:67], v[44:45]\n v_fma_f64 v[48:49], v[64:65], v[66:67], v[48:49]\n v_fma_f64 v[52:53], v[64:65], v[66:67], v[52:53]\n v_fma_f64 v[56:57], v[64:65], v[66:67], v[56:57]\n v_fma_f64 v[60:61], v[64:65], v[66:67], v[60:61]\n")
// v_fmac (VOP2 form, what hipcc emits: dst = src2) with the same placements
KERNEL(k_fmac_even_odd, "v_fmac_f64_e32 v[32:33], v[64:65], v[66:67]\n v_fmac_f64_e32 v[36:37], v[64:65], v[66:67]\n v_fmac_f64_e32 v[40:41], v[64:65], v[66:67]\n v_fmac_f64_e32 v[44:45], v[64:65], v[66:67]\n v_fmac_f64_e32 v[48:49], v[64:65], v[66:67]\n v_fmac_f64_e32 v[52:53], v[64:65], v[66:67]\n v_fmac_f64_e32 v[56:57], v[64:65], v[66:67]\n v_fmac_f64_e32 v[60:61], v[64:65], v[66:67]\n" "v_fmac_f64_e32 v[34:35], v[64:65], v[66:67]\n v_fmac_f64_e32 v[38:39], v[64:65], v[66:67]\n v_fmac_f64_e32 v[42:43], v[64:65], v[66:67]\n v_fmac_f64_e32 v[46:47], v[64:65], v[66:67]\n v_fmac_f64_e32 v[50:51], v[64:65], v[66:67]\n v_fmac_f64_e32 v[54:55], v[64:65], v[66:67]\n v_fmac_f64_e32 v[58:59], v[64:65], v[66:67]\n v_fmac_f64_e32 v[62:63], v[64:65], v[66:67]\n")

template <typename F> static double time_ms(F f)
{
	hipEvent_t a, b;
	(void)hipEventCreate(&a); (void)hipEventCreate(&b);
	f();
	(void)hipDeviceSynchronize();
	(void)hipEventRecord(a);
	f();
	(void)hipEventRecord(b);
	(void)hipEventSynchronize(b);
	float ms;
	(void)hipEventElapsedTime(&ms, a, b);
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
	printf("device: %s, %d CUs; nominal cycles (2.4 GHz) per v_fma_f64 per SIMD at 8 / 4 waves per SIMD\n", prop.gcnArchName, cus);
#define RUN(NAME) \
	{ \
		double c[2]; int w = 0; \
		for (int wpc : {32, 16}) { \
			dim3 grid(cus * wpc / 4), block(256); \
			const double ms = time_ms([&] { hipLaunchKernelGGL(NAME, grid, block, 0, 0, out, iters); }); \
			c[w++] = ms * 1e-3 * 2.4e9 / ((double)iters * 16 * (wpc / 4)); \
		} \
		printf("%-28s %6.2f %6.2f\n", #NAME, c[0], c[1]); \
	}
	RUN(k_even_even) RUN(k_even_odd) RUN(k_same) RUN(k_sgpr)
	RUN(k_acc_even_src_odd_odd) RUN(k_acc_even_src_even_even) RUN(k_acc_even_src_even_odd) RUN(k_fmac_even_odd)
	return 0;
}
