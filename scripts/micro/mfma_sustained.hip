// Developer microbenchmark for gfx950: does the FP64 matrix unit SUSTAIN a higher rate than the FP64 vector unit?
// r01 measured v_mfma_f64_16x16x4 at the vector rate in a short run; r03 found that a v_fma_f64 stream pulls the clock
// down to 1.8-1.9 GHz within a second.  Here each stream runs for several seconds; throughput is reported for the first
// 0.25 s and for the last second, with the in-kernel clock (s_memtime over s_memrealtime).
//   fma   : 32 independent v_fma_f64 per iteration
//   mfma  : 8 v_mfma_f64_16x16x4_f64 per iteration on 4 independent accumulator tiles (2048 flop each)
//   mixed : 4 mfma + 16 fma per iteration in the same wave (equal flops from each unit)
//   fmarnd: the fma stream on operands that keep changing (per-lane, per-register multipliers; the accumulators never settle),
//           to see whether the clock the stream holds depends on the data
// and, last, single 1.5-ms launches of the fma stream each after 100 ms of idle: the clock a short kernel sees.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_sustained.hip -o mfma_sustained
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#include <thread>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void k_stream(double *out, int iters, unsigned long long *clk)
{
	double acc[16];
	for (int i = 0; i < 16; i++) acc[i] = threadIdx.x * 1e-3 + i;
	v4d tile[4];
	for (int i = 0; i < 4; i++) tile[i] = v4d{0.0, 0.0, 0.0, 0.0};
	const double a = 1e-3 * (threadIdx.x % 7), b = 1e-3 * (threadIdx.x % 5);
	const unsigned long long c0 = __builtin_amdgcn_s_memtime(), w0 = __builtin_amdgcn_s_memrealtime();
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) {
#pragma unroll
			for (int f = 0; f < 32; f++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[f % 16]) : "v"(0.999), "v"(0.001));
		} else if (MODE == 3) {
			// x <- x * m + c with |m| slightly above and below 1 in turn and c of alternating sign: bounded, never settles
#pragma unroll
			for (int f = 0; f < 32; f++)
				asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[f % 16]) : "v"(f < 16 ? -1.0000001 - a : -0.9999999 + b), "v"(f < 16 ? 0.37 + b : -0.41 - a));
		} else if (MODE == 1) {
#pragma unroll
			for (int m = 0; m < 8; m++) tile[m % 4] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, tile[m % 4], 0, 0, 0);
		} else {
#pragma unroll
			for (int m = 0; m < 4; m++) {
				tile[m] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, tile[m], 0, 0, 0);
#pragma unroll
				for (int f = 0; f < 4; f++) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(acc[4 * m + f]) : "v"(0.999), "v"(0.001));
			}
		}
	}
	const unsigned long long c1 = __builtin_amdgcn_s_memtime(), w1 = __builtin_amdgcn_s_memrealtime();
	if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
	double s = 0;
	for (int i = 0; i < 16; i++) s += acc[i];
	for (int i = 0; i < 4; i++) s += tile[i][0] + tile[i][1] + tile[i][2] + tile[i][3];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double flops_per_wave_iter(int mode)
{
	return mode == 0 || mode == 3 ? 32 * 64 * 2.0 : mode == 1 ? 8 * 2048.0 : 4 * 2048.0 + 16 * 64 * 2.0;
}

template <int MODE> static int run(const char *name, double *out, unsigned long long *clk, int cus, int waves_per_simd, double seconds)
{
	const int iters = 20000;
	const int nb = cus * waves_per_simd;		// 256-thread workgroups = 4 waves = one per SIMD
	const double flop_launch = flops_per_wave_iter(MODE) * iters * 4.0 * nb;
	hipEvent_t e0, e1;
	CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
	double t_total = 0, first = 0, first_t = 0, last = 0, last_t = 0, first_mhz = 0, last_mhz = 0;
	int n = 0, n_first = 0, n_last = 0;
	while (t_total < seconds) {
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_stream<MODE>), dim3(nb), dim3(256), 0, 0, out, iters, clk);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		double mhz = 0;
		for (int b = 0; b < nb; b++) mhz += (double)clk[2 * b] / (double)clk[2 * b + 1] * 100.0;
		mhz /= nb;
		if (t_total < 0.25) { first += flop_launch; first_t += ms * 1e-3; first_mhz += mhz; n_first++; }
		if (t_total > seconds - 1.0) { last += flop_launch; last_t += ms * 1e-3; last_mhz += mhz; n_last++; }
		t_total += ms * 1e-3;
		n++;
	}
	const double tf_first = first / first_t * 1e-12, tf_last = last / last_t * 1e-12;
	printf("  %-6s %d waves/SIMD: first 0.25 s %6.1f TF/s at %4.0f MHz; last second of %.0f: %6.1f TF/s at %4.0f MHz  (%d launches)\n",
	       name, waves_per_simd, tf_first, first_mhz / n_first, seconds, tf_last, last_mhz / n_last, n);
	fflush(stdout);
	return 0;
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	double *out;
	unsigned long long *clk;
	CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 256));
	CHK(hipMallocManaged(&clk, sizeof(unsigned long long) * 2 * cus * 8));
	printf("device: %s, %d CUs\n", prop.gcnArchName, cus);
	for (int w : {4, 8}) {
		if (run<0>("fma", out, clk, cus, w, 5.0)) return 1;
		if (run<1>("mfma", out, clk, cus, w, 5.0)) return 1;
		if (run<2>("mixed", out, clk, cus, w, 5.0)) return 1;
		if (run<3>("fmarnd", out, clk, cus, w, 5.0)) return 1;
	}
	// short launches after idle
	for (int rep = 0; rep < 6; rep++) {
		std::this_thread::sleep_for(std::chrono::milliseconds(100));
		const int nb = cus * 4, iters = 6000;
		hipEvent_t e0, e1;
		CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
		CHK(hipEventRecord(e0));
		hipLaunchKernelGGL((k_stream<0>), dim3(nb), dim3(256), 0, 0, out, iters, clk);
		CHK(hipEventRecord(e1));
		CHK(hipEventSynchronize(e1));
		float ms;
		CHK(hipEventElapsedTime(&ms, e0, e1));
		double mhz = 0;
		for (int b = 0; b < nb; b++) mhz += (double)clk[2 * b] / (double)clk[2 * b + 1] * 100.0;
		printf("  fma, one %.2f-ms launch after 100 ms idle: %6.1f TF/s at %4.0f MHz\n", ms, flops_per_wave_iter(0) * iters * 4.0 * nb / (ms * 1e-3) * 1e-12, mhz / nb);
	}
	return 0;
}
