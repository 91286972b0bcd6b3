// Developer microbenchmark for gfx950: what an LDS gather costs the FP64 vector unit BESIDE it.
// One loop iteration = NF independent v_fma_f64 (asm, so exactly these) + R ds_read_b128 into registers nothing reads
// (conflict-free addresses; R = 0, 2, 4, 8, 12, 16), at 4 or 5 waves per SIMD, the occupancy of the S-side pass.  The reads are
// requested at the top of the iteration and waited for at its end: their latency lies under the FMAs of the same wave and
// under the other waves' work.  If the gather were free beside the vector unit, the time would not depend on R until the LDS
// itself saturates (4 SIMDs x R x 4 LDS cycles against NF x 4 vector cycles per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 lds_valu.hip -o lds_valu
#include <hip/hip_runtime.h>
#include <stdio.h>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

template <int NF, int R, bool B64>
__global__ __launch_bounds__(128) void k_mix(double *out, int iters, unsigned long long *clk)
{
	__shared__ __attribute__((aligned(16))) double tile[32 * 8 * 2];	// 32 rows of 64 bytes, as a P tile of 8 loci x 4 alleles at K = 8
	for (int x = threadIdx.x; x < 32 * 8 * 2; x += 128) tile[x] = 1e-3 * x;
	__syncthreads();
	double acc[16];
	for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
	// each lane gathers "its" row, as the sparse pass does: rows differ between lanes (lane % 4), 64 bytes each
	const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) const char *)(const char *)tile + (threadIdx.x % 4) * 64;
	v2d sink[16];
	double sink64[16];
	// in-kernel clock: shader cycles (s_memtime) over the 100 MHz constant counter (s_memrealtime), MI355X_MICROARCH.md DVFS item 6
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int r = 0; r < R; r++) {
			if (B64) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(sink64[r]) : "v"(addr), "n"((r % 8) * 8 + (r / 8) * 256));
			else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(sink[r]) : "v"(addr), "n"((r % 4) * 16 + (r / 4) * 256));
		}
#pragma unroll
		for (int f = 0; f < NF; f++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[f % 16]) : "v"(0.999), "v"(0.001));
		asm volatile("s_waitcnt lgkmcnt(0)");
	}
	const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
	double s = 0;
	for (int i = 0; i < 16; i++) s += acc[i];
	for (int r = 0; r < R; r++) {
		if (B64) asm volatile("" :: "v"(sink64[r]));
		else asm volatile("" :: "v"(sink[r]));
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

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

template <int NF, int R, bool B64> static void run(double *out, int cus, int waves_per_simd)
{
	static unsigned long long *clk = nullptr;
	if (!clk) (void)hipMallocManaged(&clk, sizeof(unsigned long long) * 2 * 256 * 8 * 2 * 2);

	const int iters = 4000;
	// 128-thread workgroups = 2 waves; waves_per_simd x 4 SIMDs x CUs waves in all, one round
	dim3 grid(cus * waves_per_simd * 2), block(128);
	const double ms = time_ms([&] { hipLaunchKernelGGL((k_mix<NF, R, B64>), grid, block, 0, 0, out, iters, clk); });
	(void)hipDeviceSynchronize();
	double mhz = 0;
	const int nb = cus * waves_per_simd * 2;
	for (int b = 0; b < nb; b++) mhz += (double)clk[2 * b] / (double)clk[2 * b + 1] * 100.0;
	mhz /= nb;
	// per SIMD: waves_per_simd waves x iters iterations; "cycles" at 2.4 GHz nominal, as in the other microbenchmarks
	const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * waves_per_simd);
	printf("  %2d fma + %2d %s, %d waves/SIMD: %7.3f ms  %6.1f cycles per wave-iteration per SIMD  (%.2f per fma); in-kernel clock %.0f MHz -> %.2f shader cycles per fma\n", NF, R,
	       B64 ? "ds_read_b64 " : "ds_read_b128", waves_per_simd, ms, cyc, cyc / NF, mhz, NF ? cyc / NF * mhz / 2400.0 : 0.0);
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	double *out;
	CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 2 * 128 * 2));
	printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
	for (int w : {4, 5, 8}) {
		printf("%d waves per SIMD\n", w);
		run<40, 0, false>(out, cus, w);
		run<40, 2, false>(out, cus, w);
		run<40, 4, false>(out, cus, w);
		run<40, 8, false>(out, cus, w);
		run<40, 12, false>(out, cus, w);
		run<40, 16, false>(out, cus, w);
		run<40, 16, true>(out, cus, w);
		run<20, 8, false>(out, cus, w);
		run<0, 8, false>(out, cus, w);
		run<0, 16, false>(out, cus, w);
	}
	return 0;
}
