// Developer microbenchmarks for gfx950 FP64: v_fma_f64 / v_rcp_f64 / v_mfma_f64_16x16x4 rates, MFMA f64 layout,
// and VALU || MFMA co-execution.  Build: hipcc --offload-arch=gfx950 -O3 fp64_micro.hip -o fp64_micro
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#include <math.h>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_fma(double *out, int iters, double a, double b)
{
	double x[16];
	for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 16; i++) x[i] = __builtin_fma(x[i], a, b);
	}
	double s = 0;
	for (int i = 0; i < 16; i++) s += x[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_rcp(double *out, int iters)
{
	double x[8];
	for (int i = 0; i < 8; i++) x[i] = 1.0 + threadIdx.x * 1e-3 + i;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < 8; i++) x[i] = __builtin_amdgcn_rcp(x[i]) + 1.0;
	}
	double s = 0;
	for (int i = 0; i < 8; i++) s += x[i];
	out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ void k_mfma(double *out, int iters, int nacc)
{
	double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
	double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	for (int it = 0; it < iters; it++) {
		c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
		c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
		c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
		c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
	}
	double4_t c = c0 + c1 + c2 + c3;
	out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
}

// one chain only (dependent accumulator)
__global__ void k_mfma_chain(double *out, int iters)
{
	double4_t c0 = {0, 0, 0, 0};
	double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
	for (int it = 0; it < iters; it++) {
		c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
		c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
		c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
		c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
	}
	out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c0[1] + c0[2] + c0[3];
}

// waves with even wave id run FMAs, odd run MFMAs (co-execution on each SIMD when 2 waves/SIMD)
__global__ void k_mix(double *out, int iters)
{
	const int wave = threadIdx.x >> 6;
	if ((wave >> 2) & 1) {	// waves 4..7 of a 512-thread block: the second wave on each SIMD
		double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
		double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
		for (int it = 0; it < iters / 4; it++) {
			c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
			c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
			c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
			c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
		}
		double4_t c = c0 + c1 + c2 + c3;
		out[blockIdx.x * blockDim.x + threadIdx.x] = c[0] + c[1] + c[2] + c[3];
	} else {
		double x[16];
		for (int i = 0; i < 16; i++) x[i] = threadIdx.x * 1e-3 + i;
		for (int it = 0; it < iters; it++) {
#pragma unroll
			for (int i = 0; i < 16; i++) x[i] = __builtin_fma(x[i], 0.999, 0.001);
		}
		double s = 0;
		for (int i = 0; i < 16; i++) s += x[i];
		out[blockIdx.x * blockDim.x + threadIdx.x] = s;
	}
}

// accuracy of v_rcp_f64 and of 1 / 2 Newton steps on it, in ulps of the correctly rounded 1/t
__global__ void k_rcp_acc(const double *t, double *e0, double *e1, double *e2, int n)
{
	int i = blockIdx.x * blockDim.x + threadIdx.x;
	if (i >= n) return;
	double x = t[i];
	double exact = 1.0 / x;
	double r0 = __builtin_amdgcn_rcp(x);
	double e = __builtin_fma(-x, r0, 1.0);
	double r1 = __builtin_fma(r0, e, r0);
	e = __builtin_fma(-x, r1, 1.0);
	double r2 = __builtin_fma(r1, e, r1);
	double ulp = exact * 2.220446049250313e-16;
	e0[i] = fabs(r0 - exact) / ulp; e1[i] = fabs(r1 - exact) / ulp; e2[i] = fabs(r2 - exact) / ulp;
}

// layout check: D = A(16x4) * B(4x16), integer data
__global__ void k_layout(const double *A, const double *B, double *D)
{
	const int l = threadIdx.x;
	// hypothesis: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]
	double a = A[(l & 15) * 4 + (l >> 4)];
	double b = B[(l >> 4) * 16 + (l & 15)];
	double4_t c = {0, 0, 0, 0};
	c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
	// hypothesis: c[r] = D[row = (l >> 4) + 4 r][col = l & 15]
	for (int r = 0; r < 4; r++) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = c[r];
}

template <typename F> static double time_ms(F f, int reps = 5)
{
	hipEvent_t e0, e1;
	hipEventCreate(&e0); hipEventCreate(&e1);
	f();
	hipDeviceSynchronize();
	float best = 1e30f;
	for (int r = 0; r < reps; r++) {
		hipEventRecord(e0);
		f();
		hipEventRecord(e1);
		hipEventSynchronize(e1);
		float ms; hipEventElapsedTime(&ms, e0, e1);
		if (ms < best) best = ms;
	}
	return best;
}

int main()
{
	hipDeviceProp_t prop;
	CHK(hipGetDeviceProperties(&prop, 0));
	const int cus = prop.multiProcessorCount;
	printf("device: %s, %d CUs, clock %d kHz\n", prop.gcnArchName, cus, prop.clockRate);
	double *out;
	CHK(hipMalloc(&out, sizeof(double) * cus * 8 * 512));
	const int iters = 20000;
	for (int wpc : {4, 8, 16, 20, 24, 32}) {	// waves per CU
		dim3 grid(cus * wpc / 4), block(256);
		double ms = time_ms([&] { hipLaunchKernelGGL(k_fma, grid, block, 0, 0, out, iters, 0.999, 0.001); });
		double flops = (double)grid.x * 256 * iters * 16 * 2;
		printf("v_fma_f64: %2d waves/CU: %.3f ms, %.1f TFLOP/s (%.2f cycles/wave-instr/SIMD at 2.4 GHz)\n", wpc, ms, flops / ms / 1e9,
		       ms * 1e-3 * 2.4e9 / ((double)iters * 16 * wpc / 4));
	}
	for (int wpc : {4, 8, 16}) {
		dim3 grid(cus * wpc / 4), block(256);
		double ms = time_ms([&] { hipLaunchKernelGGL(k_rcp, grid, block, 0, 0, out, iters); });
		printf("v_rcp_f64+add: %2d waves/CU: %.3f ms (%.2f cycles per rcp+add pair/SIMD at 2.4 GHz)\n", wpc, ms,
		       ms * 1e-3 * 2.4e9 / ((double)iters * 8 * wpc / 4));
	}
	for (int wpc : {4, 8}) {
		dim3 grid(cus * wpc / 4), block(256);
		double ms = time_ms([&] { hipLaunchKernelGGL(k_mfma, grid, block, 0, 0, out, iters / 4, 4); });
		double flops = (double)grid.x * 4 * (iters / 4) * 4 * 2048.0;
		printf("mfma_f64_16x16x4 (4 acc): %2d waves/CU: %.3f ms, %.1f TFLOP/s (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wpc, ms, flops / ms / 1e9,
		       ms * 1e-3 * 2.4e9 / ((double)(iters / 4) * 4 * wpc / 4));
		ms = time_ms([&] { hipLaunchKernelGGL(k_mfma_chain, grid, block, 0, 0, out, iters / 4); });
		printf("mfma_f64_16x16x4 (1 chain): %2d waves/CU: %.3f ms (%.1f cycles/MFMA/SIMD at 2.4 GHz)\n", wpc, ms,
		       ms * 1e-3 * 2.4e9 / ((double)(iters / 4) * 4 * wpc / 4));
	}
	{
		dim3 grid(cus), block(512);
		double ms_mix = time_ms([&] { hipLaunchKernelGGL(k_mix, grid, block, 0, 0, out, iters); });
		dim3 g2(cus), b2(256);
		double ms_fma = time_ms([&] { hipLaunchKernelGGL(k_fma, g2, b2, 0, 0, out, iters, 0.999, 0.001); });
		double ms_mfma = time_ms([&] { hipLaunchKernelGGL(k_mfma, g2, b2, 0, 0, out, iters / 4, 4); });
		printf("co-execution (1 FMA wave + 1 MFMA wave per SIMD): %.3f ms; FMA alone %.3f ms; MFMA alone (same count) %.3f ms\n", ms_mix, ms_fma, ms_mfma);
	}
	{
		const int n = 1 << 20;
		std::vector<double> t(n), a(n), b(n), c(n);
		unsigned long long st = 88172645463325252ull;
		for (int i = 0; i < n; i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; double u = (st >> 11) / 9007199254740992.0; t[i] = pow(10.0, -9.0 * u); }
		double *dt, *d0, *d1, *d2;
		CHK(hipMalloc(&dt, n * 8)); CHK(hipMalloc(&d0, n * 8)); CHK(hipMalloc(&d1, n * 8)); CHK(hipMalloc(&d2, n * 8));
		CHK(hipMemcpy(dt, t.data(), n * 8, hipMemcpyHostToDevice));
		hipLaunchKernelGGL(k_rcp_acc, dim3(n / 256), dim3(256), 0, 0, dt, d0, d1, d2, n);
		CHK(hipMemcpy(a.data(), d0, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(b.data(), d1, n * 8, hipMemcpyDeviceToHost)); CHK(hipMemcpy(c.data(), d2, n * 8, hipMemcpyDeviceToHost));
		double m0 = 0, m1 = 0, m2 = 0;
		for (int i = 0; i < n; i++) { if (a[i] > m0) m0 = a[i]; if (b[i] > m1) m1 = b[i]; if (c[i] > m2) m2 = c[i]; }
		printf("v_rcp_f64 max error: raw %.3g ulp, +1 Newton %.3g ulp, +2 Newton %.3g ulp (t in [1e-9,1], 1M samples)\n", m0, m1, m2);
	}
	// layout
	std::vector<double> A(64), B(64), D(256), Dd(256);
	for (int i = 0; i < 16; i++) for (int k = 0; k < 4; k++) A[i * 4 + k] = i * 7 + k * 3 + 1;
	for (int k = 0; k < 4; k++) for (int j = 0; j < 16; j++) B[k * 16 + j] = k * 5 + j * 11 + 2;
	for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < 4; k++) s += A[i * 4 + k] * B[k * 16 + j]; D[i * 16 + j] = s; }
	double *dA, *dB, *dD;
	CHK(hipMalloc(&dA, 64 * 8)); CHK(hipMalloc(&dB, 64 * 8)); CHK(hipMalloc(&dD, 256 * 8));
	CHK(hipMemcpy(dA, A.data(), 64 * 8, hipMemcpyHostToDevice)); CHK(hipMemcpy(dB, B.data(), 64 * 8, hipMemcpyHostToDevice));
	hipLaunchKernelGGL(k_layout, dim3(1), dim3(64), 0, 0, dA, dB, dD);
	CHK(hipMemcpy(Dd.data(), dD, 256 * 8, hipMemcpyDeviceToHost));
	int bad = 0;
	for (int x = 0; x < 256; x++) if (Dd[x] != D[x]) bad++;
	printf("mfma_f64_16x16x4 layout hypothesis (A[l&15][l>>4], B[l>>4][l&15], D[(l>>4)+4r][l&15]): %s (%d mismatches)\n", bad ? "WRONG" : "OK", bad);
	return 0;
}
