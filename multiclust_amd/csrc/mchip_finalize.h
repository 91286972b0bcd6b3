/* mchip_finalize.h -- device code of the P-side finaliser that two translation units launch: mchip.hip (k_finalize_p_tile by
 * itself: mixture model, first M step, shared mixing proportions) and every mchip_kernels_k.hip (k_finalize_qp: the finalisers of
 * an EM step's two sides in one launch).  Internal to the library. */
#ifndef MCHIP_FINALIZE_H
#define MCHIP_FINALIZE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mchip_internal.h"

/* simplex.c:109-143 on a strided vector in memory; fixed-entry set kept in a 64-bit mask (len <= 64)
 * or in byte flags (longer) */
static __device__ void michelot_strided(double *x, int stride, int len, double mn, uint8_t *flags)
{
	unsigned long long fixed = 0ull;
	int n = len;
	if (len > 64) for (int j = 0; j < len; j++) flags[(size_t)j * stride] = 0;
	while (n) {
		double csum = 0.0;
		for (int j = 0; j < len; j++) csum += x[(size_t)j * stride];
		const double shift = (csum - 1.0) / (double)n;
		bool can_terminate = true;
		for (int j = 0; j < len; j++) {
			const bool is_fixed = (len > 64) ? (flags[(size_t)j * stride] != 0) : (((fixed >> j) & 1ull) != 0);
			if (is_fixed) continue;
			double v = x[(size_t)j * stride] - shift;
			if (v < mn) {
				v = mn;
				if (len > 64) flags[(size_t)j * stride] = 1; else fixed |= 1ull << j;
				n--;
				can_terminate = false;
			}
			x[(size_t)j * stride] = v;
		}
		if (can_terminate) break;
	}
}

/* simplex.c:109-143 on up to 8 values held in registers (static indices; entries past len are ignored): the arithmetic of
 * michelot_strided, operation for operation */
__device__ __forceinline__ void michelot_small(double (&x)[8], int len, double mn)
{
	unsigned fixed = 0u;
	int n = len;
	while (n) {
		double csum = 0.0;
#pragma unroll
		for (int j = 0; j < 8; j++)
			if (j < len) csum += x[j];
		const double shift = (csum - 1.0) / (double)n;
		bool can_terminate = true;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (j >= len || ((fixed >> j) & 1u)) continue;
			double v = x[j] - shift;
			if (v < mn) {
				v = mn;
				fixed |= 1u << j;
				n--;
				can_terminate = false;
			}
			x[j] = v;
		}
		if (can_terminate) break;
	}
}

/* The same with the memory side done by whole blocks.  k_finalize_p's (l, k) threads walk their M elements K doubles apart: every
 * load instruction of a wave touches eight half-used cache lines, and the slab reads ran at 2.9 TB/s (40 us per launch at config
 * 3, the largest of the small kernels of a cycle).  Here a block owns FP_LOCI consecutive loci, i.e. one contiguous run of
 * (c1 - c0) * K elements of every slab: phase 1 adds the slabs element by element, coalesced, into LDS (same sums, same order:
 * ordered_sum over the slabs, times P[from], plus the additive bound); phase 2 is k_finalize_p's arithmetic per (l, k) on the LDS
 * copy; phase 3 stores the run, coalesced.  Same bits as k_finalize_p.  Used while FP_LOCI loci of max_M alleles fit the tile. */
constexpr int FP_TILE = 1024;	/* doubles of LDS per block: four elements per thread, all of a thread's slab loads in flight at once */
/* (the body: block `block` of the launch, `tile` = FP_TILE doubles of LDS; k_finalize_p_tile in mchip.hip, and the second part of
 * the grid of k_finalize_qp in mchip_kernels_k.hip) */
__device__ __forceinline__ void finalize_p_tile_body(int block, double *tile, int L, int K, int T, const int32_t *__restrict__ toff, int loci_per_block,
		int n_slabs, const double *__restrict__ Apart, const double *Pfrom, double *Pto,
		int weighted, double add_lb, int do_projection, double lb)
{
	const int l0 = block * loci_per_block, l1 = min(L, l0 + loci_per_block);
	const int c0 = toff[l0], c1 = toff[l1];
	const int nel = (c1 - c0) * K;
	const size_t e0 = (size_t)c0 * K, slab = (size_t)T * K;
	if (nel > 0) {	/* a thread's (up to) four elements side by side: their loads of one slab go out together, the sums keep the slab order */
		double v[FP_TILE / MCHIP_BLOCK];
#pragma unroll
		for (int y = 0; y < FP_TILE / MCHIP_BLOCK; y++) v[y] = 0.0;
		for (int sl = 0; sl < n_slabs; sl++) {
			double w[FP_TILE / MCHIP_BLOCK];
#pragma unroll
			for (int y = 0; y < FP_TILE / MCHIP_BLOCK; y++) {
				const int x = min((int)threadIdx.x + y * MCHIP_BLOCK, nel - 1);
				w[y] = Apart[(size_t)sl * slab + e0 + x];
			}
#pragma unroll
			for (int y = 0; y < FP_TILE / MCHIP_BLOCK; y++) v[y] += w[y];
		}
#pragma unroll
		for (int y = 0; y < FP_TILE / MCHIP_BLOCK; y++) {
			const int x = threadIdx.x + y * MCHIP_BLOCK;
			if (x < nel) {
				double t = v[y];
				if (weighted) t *= Pfrom[e0 + x];
				tile[x] = t + add_lb;
			}
		}
	}
	__syncthreads();
	for (int y = threadIdx.x; y < (l1 - l0) * K; y += MCHIP_BLOCK) {
		const int k = y % K, l = l0 + y / K;
		const int m0 = toff[l] - c0, M = toff[l + 1] - toff[l];
		double *col = tile + (size_t)m0 * K + k;	/* the locus's M values of cluster k, K doubles apart */
		double temp = 0.0;
		if (M <= 8) {
			double v[8];
#pragma unroll
			for (int m = 0; m < 8; m++) {
				v[m] = 0.0;
				if (m < M) {
					v[m] = col[(size_t)m * K];
					temp += v[m];
				}
			}
#pragma unroll
			for (int m = 0; m < 8; m++)
				if (m < M) v[m] /= temp;
			if (do_projection) michelot_small(v, M, lb);
#pragma unroll
			for (int m = 0; m < 8; m++)
				if (m < M) col[(size_t)m * K] = v[m];
		} else {
			for (int m = 0; m < M; m++) temp += col[(size_t)m * K];
			for (int m = 0; m < M; m++) col[(size_t)m * K] /= temp;
			if (do_projection) michelot_strided(col, K, M, lb, nullptr);	/* (M <= 64 here: the fixed set is a bit mask) */
		}
	}
	__syncthreads();
	for (int x = threadIdx.x; x < nel; x += MCHIP_BLOCK) Pto[e0 + x] = tile[x];
}


#endif
