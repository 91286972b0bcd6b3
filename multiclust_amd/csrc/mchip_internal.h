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
/* Column pass: lanes that share one allele column (each holds a range of k of P_.c and of the N-side accumulators; the q rows of
 * the individuals come from LDS instead of scalar registers, the partial dot products are combined across the lanes).  One lane
 * holds 4K doubles and wants 2K scalar registers: neither fits at large K (67 SGPRs spilled at K = 64: 14.0 ms at the 5 000 x
 * 50 000 shape).  Measured (profiles/r03_k_sweep.txt): two lanes per column win from K = 37 (K = 38: 3.34 ms with one lane; K = 40: 2.91 against 3.06 ms) to K = 64
 * (4.83 against 13.95 ms; K = 48: 3.77 against 5.45 ms) and lose at K = 36 (2.64 against 2.11 ms); four lanes per column lose to
 * two at every K tried (K = 48: 4.12 ms, K = 64: 5.50 ms): every lane repeats the count extraction and the reciprocal. */
#ifndef MCHIP_COL_SPLIT2_ABOVE
#define MCHIP_COL_SPLIT2_ABOVE 36
#endif
#ifndef MCHIP_COL_SPLIT4_ABOVE
#define MCHIP_COL_SPLIT4_ABOVE 9999	/* experiments only */
#endif
constexpr int mchip_col_split(int K) { return K > MCHIP_COL_SPLIT4_ABOVE ? 4 : (K > MCHIP_COL_SPLIT2_ABOVE ? 2 : 1); }
/* LDS row stride of the staged P tiles in doubles: rows stay 16-byte aligned and hold the lanes' k ranges (each padded to an even
 * count); a stride of 128 or 256 bytes would put the rows of a locus on the same banks (64 banks x 4 bytes), so multiples of 16
 * get two doubles of padding */
/* LDS doubles between the lane parts of a staged row (sparse individual pass with a lane split): each part's KSP doubles, plus
 * padding where two parts would start on the same banks (64 banks x 4 bytes, a 16-byte read covers 4): KSP = 16 (K = 57 ... 64)
 * put parts 0 / 2 and 1 / 3 on the same banks and cost the S-side pass 40 % (6.0 against 4.2 ms at K = 56) */
constexpr bool mchip_parts_conflict(int split, int stride)
{
	for (int a = 0; a < split; a++)
		for (int b = a + 1; b < split; b++) {
			const int d = (2 * stride * (b - a)) % 64;
			if (d < 4 || d > 60) return true;
		}
	return false;
}
constexpr int mchip_ind_pstride(int K)
{
	const int split = mchip_ind_split(K);
	int stride = ((((K + split - 1) / split) + 1) & ~1);
	while (split > 1 && mchip_parts_conflict(split, stride)) stride += 2;
	return stride;
}
constexpr int mchip_kp(int K)
{
	const int split = mchip_ind_split(K);
	const int per_lane = mchip_ind_pstride(K) * split;
	const int base = per_lane > ((K + 1) & ~1) ? per_lane : ((K + 1) & ~1);
	return base + ((base % 16 == 0) ? 2 : 0);
}

/* Cooperating waves (round 4).  Both streaming passes reduce thread-privately over a chunk of the other axis and leave one slab
 * of partial sums per chunk; the slabs cost HBM traffic twice (written, then added up) and were 131 MB (S side, 205 slabs) and
 * 269 MB (N side, 14 slabs) per EM step at config 3.  The waves of a workgroup now take the SAME lanes' worth of individuals
 * (columns) and consecutive sub-chunks of loci (individuals), and add their sums through LDS in wave order before one of them
 * stores the slab: as many waves and as much work per wave as before, a quarter of the slabs.  For the sparse individual pass
 * this also makes the staged P tile private to a wave (each wave is at other loci): no workgroup barrier inside the loop. */
constexpr int MCHIP_COL_WAVES = MCHIP_BLOCK / 64;	/* column pass on packed counts: waves per workgroup = sub-chunks per slab */
#ifndef MCHIP_IND_WAVES_MAX
#define MCHIP_IND_WAVES_MAX 4
#endif
/* sparse individual pass: waves per workgroup, each with its own double-buffered tile of tile_cols staged rows (two such
 * pairs where the dual pass exists, K <= 12): the largest of 4, 2, 1 whose tiles fit 48 KiB */
inline int mchip_ind_waves(int K, int tile_cols);

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
	int xcd_rows;		/* the slab rows of the sparse cooperating kernel are a multiple of eight: workgroup id -> (tile, row) keeps a row
				 * on one XCD */
	int no_col_split;	/* MCHIP_NO_COL_SPLIT (experiments): one lane per allele column at every K */
	int ind_waves;		/* sparse / biallelic individual passes: waves per workgroup; wave w takes sub-chunk w of lchunk loci of the
				 * workgroup's ind_waves * lchunk loci, the slab index is blockIdx.y (mchip_ind_slabs() of them) */
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

/* what the P-side part of a launch needs (k_finalize_qp takes it by value beside the Q side's arguments; mchip_finalize.h has the code) */
struct mchip_finalize_p_args {
	int L, K, T, loci_per_block, n_slabs, weighted, do_projection;
	const int32_t *toff;
	const double *Apart, *Pfrom;
	double *Pto;
	double add_lb, lb;
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
	/* what the launchers above will leave for these arguments: N-side slabs of accum_p (mix = 0) / mix_column (mix = 1); S-side
	 * slabs of accum_q / accum_q_dual (= of loglik's partial sums), and partial log likelihoods of accum_q / loglik */
	int (*col_slabs)(const mchip_pass_args &a, int mix);
	int (*ind_slabs)(const mchip_pass_args &a);
	int (*ind_ll_parts)(const mchip_pass_args &a);
	/* both finalisers of an EM step in one launch (k_finalize_qp): individual mixing proportions, tiled P side */
	void (*finalize_qp)(int I, int n_slabs_q, const double *Spart, const double *Qfrom, int qstride_from, double *Qto, double *sik,
			    int do_projection_q, double lb_q, const mchip_finalize_p_args &p, const int *stop, hipStream_t s);
};

const mchip_ktable *mchip_get_ktable(int K);

inline int mchip_ind_waves(int K, int tile_cols)
{
	/* (a buffer is rounded up to whole 1 KiB pieces where tiles are copied straight into LDS) */
	const size_t per_wave = (size_t)(K <= 12 ? 4 : 2) * ((((size_t)tile_cols * (size_t)mchip_kp(K) + 127) / 128) * 128) * sizeof(double);
	for (int w = MCHIP_IND_WAVES_MAX; w > 1; w >>= 1)
		if ((size_t)w * per_wave <= 48 * 1024) return w;
	return 1;
}
/* slabs (and partial log likelihoods per individual tile) the cooperating forms of the passes leave */
__host__ __device__ inline int mchip_ind_slabs(const mchip_pass_args &a) { return (a.n_lchunks + a.ind_waves - 1) / a.ind_waves; }
__host__ __device__ inline int mchip_col_slabs(const mchip_pass_args &a) { return (a.n_ichunks + MCHIP_COL_WAVES - 1) / MCHIP_COL_WAVES; }

#endif
