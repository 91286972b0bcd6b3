/*
 * mchip_internal.h -- private definitions shared by the HIP translation units of libmulticlust_hip.so.
 * Not part of the C-ABI (include/multiclust_hip.h is).
 *
 * Device data layout (all sized for one MI355X, 288 GB HBM3E):
 *   gtA  uint8 [ceil(I/8)][L][8][ploidy]   genotype blocked by 8 individuals: one 8*ploidy-byte load per lane
 *                                           (lane = allele column) serves 8 individuals of the column pass
 *   gtC  uint4 [ceil(I/G)][T]               packed allele counts n_ic (2 or 4 bits each, G = 64 or 32 individuals per
 *                                           16-byte word): the column pass reads one word per lane per G individuals
 *   gtS  uint8 [ceil(L/8)][I][8][ploidy]   genotype blocked by 8 loci: one load per lane (lane = individual)
 *                                           serves 8 loci of the individual pass
 *   P    double [T][K]  per slot           allele column c = T_off[l]+m, K contiguous doubles per column
 *   Q    double [I][K]  per slot           (or [K] when eta is shared: row stride 0)
 *   col_locus int32 [T], col_allele uint8 [T], toff int32 [L+1], ua int32 [L]
 */
#ifndef MCHIP_INTERNAL_H
#define MCHIP_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "multiclust_hip.h"

#define MCHIP_BLOCK 256
#define MCHIP_SPARSE_MAX_M 32	/* sparse individual pass is used when no locus has more alleles than this */
#ifndef MCHIP_QBLOCK
#define MCHIP_QBLOCK 128	/* individuals per workgroup of the individual-side kernels (lane = individual) */
#endif

/* Sparse individual pass: lanes that share one individual, each holding a range of k (its part of q, of the S-side sums and of
 * every gathered P row; the partial dot products are combined across the lanes).  One lane holds 4K doubles: above K = 27 that
 * leaves one or two waves per SIMD (292 registers at K = 32), above 32 it spills. */
#ifdef MCHIP_FORCE_SPLIT1
/* DIAGNOSTIC builds only (scripts/diag/k52_spill.sh): one lane per individual at every K, the form in which the K = 52 instance
 * of the tetraploid reciprocal-per-copy pass gave wrong results with hipcc's VGPR-to-AGPR spilling (profiles/r03_k52_spill.md) */
constexpr int mchip_ind_split(int) { return 1; }
#else
constexpr int mchip_ind_split(int K) { return K <= 27 ? 1 : (K <= 48 ? 2 : 4); }
#endif
/* lanes per workgroup of the individual-side kernels: the staged P tile of a workgroup grows with K (34 KB at K = 64), so the
 * largest K share it among four waves instead of two, or LDS capacity would leave two waves per SIMD */
#ifdef MCHIP_FORCE_SPLIT1
constexpr int mchip_qblock(int) { return MCHIP_QBLOCK; }
#else
constexpr int mchip_qblock(int K) { return K > 48 ? 2 * MCHIP_QBLOCK : MCHIP_QBLOCK; }
#endif
/* LDS row stride of the staged P tiles in doubles: rows stay 16-byte aligned and hold the lanes' k ranges (each padded to an even
 * count); a stride of 128 or 256 bytes would put the rows of a locus on the same banks (64 banks x 4 bytes), so multiples of 16
 * get two doubles of padding */
constexpr int mchip_kp(int K)
{
	const int split = mchip_ind_split(K);
	const int per_lane = ((((K + split - 1) / split) + 1) & ~1) * split;
	const int base = per_lane > ((K + 1) & ~1) ? per_lane : ((K + 1) & ~1);
	return base + ((base % 16 == 0) ? 2 : 0);
}

enum { MCHIP_KERN_ACCUM_P = 0, MCHIP_KERN_ACCUM_Q = 1, MCHIP_KERN_LOGLIK = 2, MCHIP_KERN_DUAL = 3, MCHIP_KERN_COUNT = 4 };

/* arguments of the two streaming passes over the genotype matrix */
struct mchip_pass_args {
	int I, L, T, ploidy, K;
	const uint8_t *gtA, *gtS;
	const uint8_t *gtC;	/* packed allele counts [ceil(I/G)][T] x 16 bytes, G = 128/count_bits; NULL if count_bits == 0 */
	int count_bits;		/* 2 (ploidy <= 3), 4 (ploidy <= 15) or 0 */
	int has_missing;	/* any 0xFF byte in the genotype matrix */
	const int32_t *ua, *toff, *col_locus;
	const uint8_t *col_allele;
	const double *P;	/* [T][K] slot read by the E step */
	const double *Q;	/* [I][K] or [K] */
	int qstride;		/* K, or 0 when eta is shared by all individuals */
	/* column pass (lane = allele column, loop over a chunk of individuals) */
	int ichunk, n_ichunks;	/* individuals per chunk (multiple of 8) */
	double *Apart;		/* [n_ichunks][T][K] sum_i q_ik r_ic over the chunk */
	double *llpart;		/* [gridDim.x*gridDim.y] block partial log-likelihoods */
	int flush_blocks;	/* blocks of 8 individuals between log-product checks; 0 = check after every individual */
	int safe_rcp;		/* 1: t may be 0 in zero-count cells or tiny (projection off, lower bound < 1e-75): no shared reciprocals */
	/* individual pass (lane = individual, loop over a chunk of loci) */
	int lchunk, n_lchunks;	/* loci per chunk (multiple of 8) */
	double *Spart;		/* [n_lchunks][I][K] sum_c P_kc r_ic over the chunk */
	/* sparse individual pass: P rows of 8 loci staged in LDS */
	int sparse;		/* 1: sparse individual pass + N-only column pass; 0: dense pair */
	int tile_cols;		/* LDS tile capacity in allele columns (8 * max alleles per locus) */
	int biallelic;		/* every locus has exactly two allele columns: column of (l, m) is 2l + m */
	/* batched runs: when non-null and *stop != 0 every kernel of the step returns at once */
	const int *stop;
	/* batched accelerated runs: when non-null and *skip_ind != 0 the S-side pass returns at once, because the pass that
	 * took the log likelihood of these very parameters (the accepted extrapolation) already left its sums in Spart */
	const int *skip_ind;
	/* dual individual pass: a second parameter set whose log likelihood the same pass takes */
	const double *P2, *Q2;
	double *llpart2;
	/* hard-partition first M step */
	const uint8_t *asA, *asS;	/* assignment bytes in the gtA / gtS layouts */
	int part_counts;		/* 0: d_iklm = 1 per (allele, cluster) pair, missing copies skipped (random_allele_partition);
					 * 1: one count per copy, missing copies count for the individual (initialize_parameters_admixture) */
};

/* per-K kernel table (one translation unit per K keeps each hipcc job small and `make -j` parallel) */
struct mchip_ktable {
	void (*accum_p)(const mchip_pass_args &a, hipStream_t s);	/* column pass: Apart + logL */
	void (*loglik)(const mchip_pass_args &a, hipStream_t s);	/* column pass: logL only */
	void (*accum_q)(const mchip_pass_args &a, hipStream_t s);	/* individual pass: Spart */
	void (*part_p)(const mchip_pass_args &a, hipStream_t s);	/* hard partition, column pass */
	void (*part_q)(const mchip_pass_args &a, hipStream_t s);	/* hard partition, individual pass */
	/* finalize: Q[to] from Spart (normalise + project), stores expected counts */
	void (*finalize_q)(int I, int K, int n_lchunks, const double *Spart, const double *Qfrom, int qstride_from,
			   double *Qto, double *sik, int do_mstep, int weighted, int do_projection, double lb, const int *stop, hipStream_t s,
			   double add);
	void (*project_q)(int nrows, int K, double *Q, double lb, const int *stop, hipStream_t s);
	/* mixture model */
	void (*mix_gather)(const mchip_pass_args &a, hipStream_t s);	/* a.P = log P table; Spart = per-chunk sums */
	void (*mix_finalize)(int I, int n_lchunks, const double *Vpart, const double *eta, double *vik, double *llpart, int mode,
			     const int *stop, hipStream_t s);
	void (*mix_column)(const mchip_pass_args &a, hipStream_t s);	/* a.Q = vik; Apart = sum_i vik n */
	/* individual pass of (a.Q, a.P) that also takes log L of (a.Q2, a.P2) into a.llpart2, where dual_available() says so */
	int (*dual_available)(const mchip_pass_args &a);
	void (*accum_q_dual)(const mchip_pass_args &a, hipStream_t s);
};

const mchip_ktable *mchip_get_ktable(int K);

#endif
