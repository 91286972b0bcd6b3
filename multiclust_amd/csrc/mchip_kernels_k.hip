/*
 * mchip_kernels_k.hip -- the K-specialised gfx950 kernels of the admixture EM hot path.
 * Compiled once per K (hipcc -DMCHIP_K=<K>), see Makefile; each object exports mchip_ktable_<K>.
 *
 * Fused E+M without materialising d_iklm (reference em_alg.c:291-486 + 592-754):
 *   per non-empty cell (i, c=(l,m)), n = ILM[i][l][m]:
 *       t = sum_k q_ik P_kc ; r = n / t ; logL += n log t
 *       S_ik = q_ik * sum_c P_kc r_ic        (= sum_{l,m} d_iklm, em_alg.c:650-682)
 *       N_kc = P_kc * sum_i q_ik r_ic        (= sum_i d_iklm,      em_alg.c:706-731)
 * The two marginals reduce over opposite axes, so the genotype matrix is streamed by two passes whose
 * reductions are both thread-private (no cross-lane traffic, no atomics, bitwise reproducible):
 *   column pass     (lane = allele column c, loop over individuals): N-side sum_i q_ik r_ic, and logL;
 *                    q_i is wave-uniform -> scalar loads, P_kc and the accumulators live in VGPRs.
 *   individual pass (lane = individual i, loop over loci/alleles):   S-side sum_c P_kc r_ic;
 *                    P_.c is wave-uniform -> scalar loads, q_ik and the accumulators live in VGPRs.
 * All arithmetic is IEEE double (v_fma_f64); n log t is accumulated as log of a running product of t
 * (one v_log-free multiply per allele copy, one log per flush), see DESIGN.md.
 */
#include "mchip_internal.h"
#include "mchip_finalize.h"

#ifndef MCHIP_K
#error "compile with -DMCHIP_K=<K>"
#endif

namespace {

constexpr int K = MCHIP_K;

/* 1/t to full double precision for t in (0, 1]: v_rcp_f64 is good to ~2^-25 (measured 2.1e8 ulp,
 * scripts/micro/fp64_micro.hip); one second-order step r0 (1 + e + e^2), e = 1 - t r0, leaves e^3 ~ 1e-22 plus
 * rounding (measured <= 1 ulp), for 3 FMAs instead of the 4 of two Newton steps.  No scaling / fix-up: t is a
 * convex combination of probabilities >= the lower bound, never denormal or huge. */
__device__ __forceinline__ double rcp_full(double t)
{
	const double r = __builtin_amdgcn_rcp(t);
	const double e = __builtin_fma(-t, r, 1.0);
	const double p = __builtin_fma(e, e, e);
	return __builtin_fma(r, p, r);
}

/* Four reciprocals from one: 1/(t0 t1 t2 t3) by rcp_full, then back-multiplication (9 multiplies + 1 reciprocal
 * instead of 4 reciprocals; v_rcp_f64 issues at a quarter of the FMA rate).  Each t is in [p_lb/K, 1], so the
 * product of four stays far from underflow; every result carries <= 2.5 ulp. */
__device__ __forceinline__ void rcp4(const double (&t)[4], double (&rc)[4])
{
	const double p01 = t[0] * t[1], p23 = t[2] * t[3];
	const double rp = rcp_full(p01 * p23);
	const double r01 = rp * p23, r23 = rp * p01;
	rc[0] = r01 * t[1];
	rc[1] = r01 * t[0];
	rc[2] = r23 * t[3];
	rc[3] = r23 * t[2];
}

__device__ __forceinline__ void rcp_group(const double (&t)[4], double (&rc)[4]) { rcp4(t, rc); }
__device__ __forceinline__ void rcp_group(const double (&t)[2], double (&rc)[2])
{
	const double rp = rcp_full(t[0] * t[1]);
	rc[0] = rp * t[1];
	rc[1] = rp * t[0];
}
__device__ __forceinline__ void rcp_group(const double (&t)[1], double (&rc)[1]) { rc[0] = rcp_full(t[0]); }

/* genotype bytes of sub-entry j (0..7) of an 8-entry group; PL = 2 fast path keeps the group in a uint4 */
template <int PL> struct geno_group;

template <> struct geno_group<2> {
	uint4 g;
	__device__ __forceinline__ void load(const uint8_t *base, size_t group, int) { g = *reinterpret_cast<const uint4 *>(base + group * 16); }
	__device__ __forceinline__ unsigned field(int j) const
	{
		unsigned w = (j < 2) ? g.x : (j < 4) ? g.y : (j < 6) ? g.z : g.w;
		return (w >> ((j & 1) * 16)) & 0xFFFFu;
	}
	/* number of copies of entry j equal to allele m */
	__device__ __forceinline__ int count(int j, unsigned m, int) const
	{
		unsigned w = field(j);
		return (int)((w & 0xFFu) == m) + (int)((w >> 8) == m);
	}
	__device__ __forceinline__ unsigned copy(int j, int a, int) const { return (field(j) >> (8 * a)) & 0xFFu; }
	/* the loaded word is here from now on (an empty statement that reads it: hipcc puts its s_waitcnt in front of it).  Before
	 * a loop whose body starts with loads of its own: the in-order counter would otherwise make the body's first use of this
	 * group wait for those younger loads as well, every trip */
	__device__ __forceinline__ void arrive() const { asm volatile("" :: "v"(g.x), "v"(g.y), "v"(g.z), "v"(g.w)); }
	/* entries 4h..4h+3 moved to positions 0..3 (h wave-uniform) */
	__device__ __forceinline__ geno_group<2> half(int h) const
	{
		geno_group<2> r;
		r.g.x = h ? g.z : g.x;
		r.g.y = h ? g.w : g.y;
		r.g.z = 0xFFFFFFFFu;
		r.g.w = 0xFFFFFFFFu;
		return r;
	}
};

template <> struct geno_group<4> {	/* tetraploid: 8 entries x 4 bytes = two 16-byte loads */
	uint4 g0, g1;
	__device__ __forceinline__ void load(const uint8_t *base, size_t group, int)
	{
		const uint4 *p = reinterpret_cast<const uint4 *>(base + group * 32);
		g0 = p[0];
		g1 = p[1];
	}
	__device__ __forceinline__ unsigned field(int j) const
	{
		return (j == 0) ? g0.x : (j == 1) ? g0.y : (j == 2) ? g0.z : (j == 3) ? g0.w
		     : (j == 4) ? g1.x : (j == 5) ? g1.y : (j == 6) ? g1.z : g1.w;
	}
	__device__ __forceinline__ int count(int j, unsigned m, int) const
	{
		const unsigned w = field(j);
		return (int)((w & 0xFFu) == m) + (int)(((w >> 8) & 0xFFu) == m) + (int)(((w >> 16) & 0xFFu) == m) + (int)((w >> 24) == m);
	}
	__device__ __forceinline__ unsigned copy(int j, int a, int) const { return (field(j) >> (8 * a)) & 0xFFu; }
	__device__ __forceinline__ void arrive() const
	{
		asm volatile("" :: "v"(g0.x), "v"(g0.y), "v"(g0.z), "v"(g0.w), "v"(g1.x), "v"(g1.y), "v"(g1.z), "v"(g1.w));
	}
	__device__ __forceinline__ geno_group<4> half(int h) const
	{
		geno_group<4> r;
		r.g0 = h ? g1 : g0;
		r.g1 = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
		return r;
	}
};

template <> struct geno_group<0> {	/* any ploidy: byte loads */
	const uint8_t *p;
	__device__ __forceinline__ void load(const uint8_t *base, size_t group, int pl) { p = base + group * 8 * (size_t)pl; stride_pl = pl; }
	__device__ __forceinline__ int count(int j, unsigned m, int pl) const
	{
		int n = 0;
		for (int a = 0; a < pl; a++) n += (int)(p[j * pl + a] == m);
		return n;
	}
	__device__ __forceinline__ unsigned copy(int j, int a, int pl) const { return p[j * pl + a]; }
	__device__ __forceinline__ void arrive() const {}	/* (bytes are loaded where they are used) */
	__device__ __forceinline__ geno_group<0> half(int h) const
	{
		geno_group<0> r;
		r.p = p + h * 4 * stride_pl;
		r.stride_pl = stride_pl;
		return r;
	}
	int stride_pl;
};

template <int NT> __device__ __forceinline__ double block_sum(double v, double *red)
{
	const int tid = threadIdx.x;
	red[tid] = v;
	__syncthreads();
	for (int s = NT / 2; s > 0; s >>= 1) {
		if (tid < s) red[tid] += red[tid + s];
		__syncthreads();
	}
	return red[0];
}

/* ---------------------------------------------------------------- column pass */
/* SAFE = false: the log-product is checked once per `flush_blocks` blocks of 8 individuals (host guarantees that
 * 8*ploidy*flush_blocks multiplications cannot underflow a product that starts above 1e-100);
 * SAFE = true: checked after every individual (tiny lower bounds, projection disabled, high ploidy). */
template <int PL, bool ACCUM, bool SAFE, bool LL>
__global__ __launch_bounds__(MCHIP_BLOCK) void k_column_pass(mchip_pass_args a)
{
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	__shared__ double red[MCHIP_BLOCK];
	const int c_raw = blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	const bool valid = c_raw < a.T;
	const int c = valid ? c_raw : a.T - 1;
	const int l = a.col_locus[c];
	const unsigned m = valid ? (unsigned)a.col_allele[c] : 0xFEu;	/* 0xFE never matches a genotype byte */
	const int pl = PL ? PL : a.ploidy;

	double p[K], acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) {
		p[k] = a.P[(size_t)c * K + k];
		acc[k] = 0.0;
	}
	const int i0 = blockIdx.y * a.ichunk;
	const int i1 = min(a.I, i0 + a.ichunk);
	const int ib_end = (i1 + 7) >> 3;
	double ll = 0.0, prod = 1.0;
	int blk = 0;

	geno_group<PL> g, gn;
	g.load(a.gtA, (size_t)(i0 >> 3) * a.L + l, pl);
	for (int ib = i0 >> 3; ib < ib_end; ib++) {
		/* software prefetch of the next group (clamped: the last iteration re-reads its own group) */
		gn.load(a.gtA, (size_t)min(ib + 1, ib_end - 1) * a.L + l, pl);
		/* two halves of four individuals: four q rows (4*2K SGPRs) in flight at a time; unrolling all
		 * eight makes hipcc hoist eight s_load_dwordx16 and spill SGPRs through v_writelane */
#pragma unroll 1
		for (int h = 0; h < 2; h++) {
			const geno_group<PL> gh = g.half(h);
#pragma unroll
			for (int j = 0; j < 4; j++) {
				const int i = min(ib * 8 + h * 4 + j, a.I - 1);	/* padded individuals carry 0xFF bytes: n = 0 */
				const double *__restrict__ q = a.Q + (size_t)i * a.qstride;	/* wave-uniform: s_load */
				const int n = gh.count(j, m, pl);
				double t = q[0] * p[0];
#pragma unroll
				for (int k = 1; k < K; k++) t = __builtin_fma(q[k], p[k], t);
				if (ACCUM) {
					/* a zero-count cell contributes exactly 0 whatever t is (em_alg.c:338-342): a column whose P is 0 for
					 * every k (projection off) has t = 0 and 0 * (1/0) would be NaN */
					const double r = n ? (double)n * rcp_full(t) : 0.0;
#pragma unroll
					for (int k = 0; k < K; k++) acc[k] = __builtin_fma(q[k], r, acc[k]);
				}
				/* n log t as log of a product: t^n.  SAFE (projection off): a negative t in a cell that carries copies is
				 * NaN in the reference's log and must not square itself away in a homozygote (found on a 36-allele locus:
				 * tests/test_gpu_cli_differential.py) */
				if (LL) {
					const double tl = (SAFE && t < 0.0) ? __builtin_nan("") : t;
					if (PL == 2) {
						prod *= (n >= 1) ? tl : 1.0;
						prod *= (n >= 2) ? tl : 1.0;
					} else {
						for (int b = 1; b <= pl; b++) prod *= (n >= b) ? tl : 1.0;
					}
				}
				if (LL && SAFE) {
					if (prod < 1e-100) {
						ll += log(prod);
						prod = 1.0;
					}
				}
			}
		}
		if (LL && !SAFE) {
			if (++blk >= a.flush_blocks) {
				blk = 0;
				if (prod < 1e-100) {
					ll += log(prod);
					prod = 1.0;
				}
			}
		}
		g = gn;
	}
	if (ACCUM && valid) {
		double *out = a.Apart + ((size_t)blockIdx.y * a.T + c) * K;
#pragma unroll
		for (int k = 0; k < K; k++) out[k] = acc[k];
	}
	if (LL) {
		ll += log(prod);
		const double tot = block_sum<MCHIP_BLOCK>(ll, red);
		if (threadIdx.x == 0) a.llpart[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
	}
}

/* ---------------------------------------------------------------- column pass on packed allele counts
 * gtC[g][c] is one 16-byte word per (group of G = 128/BITS individuals, allele column): BITS-bit counts
 * n_ic = ILM[i][l][m] (BITS = 2 for ploidy <= 3, 4 for ploidy <= 15).  One coalesced 16-byte load per lane serves G
 * individuals and the count is a single v_bfe_u32, instead of two byte compares, a select and an add per cell.
 * MIX = false: admixture N-side sums  acc_k += q_ik * n / t;   MIX = true: mixture M step  acc_k += vik * n. */
template <int BITS, bool MIX, bool SAFE = false>
__global__ __launch_bounds__(MCHIP_BLOCK) void k_column_counts(mchip_pass_args a)
{
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	constexpr int PERWORD = 32 / BITS;		/* individuals per dword */
	constexpr int G = 4 * PERWORD;			/* individuals per 16-byte word */
	constexpr unsigned MASK = (1u << BITS) - 1u;
	/* individuals per inner step: their q rows must fit the scalar register file (2K SGPRs each, about 100 usable) */
	constexpr int NJ = K <= 12 ? 4 : (K <= 20 ? 2 : 1);
	/* the workgroup's MCHIP_COL_WAVES waves take the same 64 allele columns and consecutive sub-chunks of a.ichunk individuals;
	 * their sums meet in LDS below, in wave order, and wave 0 stores the slab (mchip_internal.h: cooperating waves) */
	__shared__ double wsum[K][64];
	const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;	/* wv is wave-uniform */
	const int c_raw = blockIdx.x * 64 + lane;
	const bool valid = c_raw < a.T;
	const int c = valid ? c_raw : a.T - 1;
	double p[K], acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) {
		p[k] = MIX ? 0.0 : a.P[(size_t)c * K + k];
		acc[k] = 0.0;
	}
	const int i0 = __builtin_amdgcn_readfirstlane((blockIdx.y * MCHIP_COL_WAVES + wv) * a.ichunk);	/* multiple of G */
	const int i1 = min(a.I, i0 + a.ichunk);
	const int g_end = (i1 + G - 1) / G;		/* (a wave whose sub-chunk starts behind the last individual has no groups) */
	const uint4 *gt = reinterpret_cast<const uint4 *>(a.gtC);
	uint4 w = gt[(size_t)min(i0 / G, (a.I - 1) / G) * a.T + c];
	for (int g = i0 / G; g < g_end; g++) {
		const uint4 wn = gt[(size_t)min(g + 1, g_end - 1) * a.T + c];	/* prefetch (clamped) */
#pragma unroll 1
		for (int wi = 0; wi < 4; wi++) {
			unsigned word = (wi == 0) ? w.x : (wi == 1) ? w.y : (wi == 2) ? w.z : w.w;
#pragma unroll 1
			for (int h = 0; h < PERWORD / NJ; h++) {
				const int ibase = g * G + wi * PERWORD + h * NJ;
				if (ibase >= i1) break;		/* wave-uniform; padded individuals have zero counts anyway */
				/* NJ individuals at a time: their q rows are wave-uniform (scalar loads, used as SGPR operands) */
				const double *__restrict__ q[NJ];
				double n[NJ], t[NJ], rc[NJ];
#pragma unroll
				for (int j = 0; j < NJ; j++) {
					q[j] = a.Q + (size_t)min(ibase + j, a.I - 1) * a.qstride;
					n[j] = (double)((word >> (BITS * j)) & MASK);
				}
				if (!MIX) {
#pragma unroll
					for (int j = 0; j < NJ; j++) {
						t[j] = q[j][0] * p[0];
#pragma unroll
						for (int k = 1; k < K; k++) t[j] = __builtin_fma(q[j][k], p[k], t[j]);
					}
					if (SAFE) {
						/* projection off / tiny lower bounds (mchip_set_model): t may be 0 in a zero-count cell (a column
						 * whose P is 0 for every k) and a product of four t's may underflow: one reciprocal per cell,
						 * and none where n = 0, so such cells contribute exactly 0 as in em_alg.c:338-342 */
#pragma unroll
						for (int j = 0; j < NJ; j++) rc[j] = (n[j] != 0.0) ? rcp_full(t[j]) : 0.0;
					} else {
						rcp_group(t, rc);
					}
				}
#pragma unroll
				for (int j = 0; j < NJ; j++) {
					const double r = MIX ? n[j] : n[j] * rc[j];
#pragma unroll
					for (int k = 0; k < K; k++) acc[k] = __builtin_fma(q[j][k], r, acc[k]);
				}
				word >>= NJ * BITS;
			}
		}
		w = wn;
	}
	/* (lane, wave and column are worked out again from the thread index, laundered so that hipcc does not keep the first copies
	 * alive across the loop: those four registers were the difference between 7 and 6 waves per SIMD at K = 8) */
	unsigned tid = threadIdx.x;
	asm volatile("" : "+v"(tid));
	const int lane2 = (int)(tid & 63u), wv2 = (int)(tid >> 6);
	for (int x = 1; x < MCHIP_COL_WAVES; x++) {	/* wave x hands its sums to wave 0: one wave's worth of LDS, taken in turns */
		if (wv2 == x) {
#pragma unroll
			for (int k = 0; k < K; k++) wsum[k][lane2] = acc[k];
		}
		__syncthreads();
		if (wv2 == 0) {
#pragma unroll
			for (int k = 0; k < K; k++) acc[k] += wsum[k][lane2];
		}
		__syncthreads();
	}
	const int c2 = (int)blockIdx.x * 64 + lane2;
	if (wv2 == 0 && c2 < a.T) {
		double *out = a.Apart + ((size_t)blockIdx.y * a.T + c2) * K;
#pragma unroll
		for (int k = 0; k < K; k++) out[k] = acc[k];
	}
}

/* ---------------------------------------------------------------- column pass on packed counts, k range split over CSPLIT lanes
 * K >= 37: CSPLIT = 2 adjacent lanes = one allele column, lane part s holds k in [s CKS, (s+1) CKS) of P_.c and of the
 * accumulators.  The q rows of one dword's worth of individuals (16 at 2 bits per count, 8 at 4) are staged in LDS
 * (double-buffered: the next batch waits in registers while this one is computed); a lane reads its part of a row with
 * ds_read_b128 (the parts of a row sit an odd number of 16-byte units apart: distinct banks), forms its partial dot product, and
 * the lanes of a column add their partials with DPP exchanges (every lane ends on the same bits: a + b = b + a at each level).
 * One reciprocal per cell. */
constexpr int CSPLIT = mchip_col_split(K);
constexpr int CKS = (K + CSPLIT - 1) / CSPLIT;		/* k per lane part */
constexpr int CKSE = (CKS + 1) & ~1;			/* rounded up to whole 16-byte reads (the padding multiplies p = 0) */
/* LDS doubles per (individual, part): the parts of a row must not start on the same banks (64 banks x 4 bytes; a 16-byte read
 * covers 4): K = 60 with a stride of 32 doubles = 256 bytes put both parts on banks 0-3 and cost 25 % */
constexpr int cqs_stride(int n) { return ((2 * n) % 64 < 4 || (2 * n) % 64 > 60) ? n + 2 : n; }
constexpr int CQS = cqs_stride(CKSE + 2);

template <int CTRL> __device__ __forceinline__ double quad_exchange(double v)
{
	return __hiloint2double(__builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, false),
				__builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, false));
}
/* sum over the four lanes of a quad: v + (the lane whose index differs in bit 0), then the same for bit 1
 * (quad_perm [1,0,3,2] = 0xB1 and [2,3,0,1] = 0x4E) */
__device__ __forceinline__ double quad_sum(double v)
{
	v += quad_exchange<0xB1>(v);
	if (CSPLIT == 4) v += quad_exchange<0x4E>(v);
	return v;
}

template <int BITS, bool SAFE>
__global__ __launch_bounds__(MCHIP_BLOCK) void k_column_counts_split(mchip_pass_args a)
{
	static_assert(CSPLIT == 4 || CSPLIT == 2 || CSPLIT == 1, "quad exchange");
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	constexpr int PERWORD = 32 / BITS;		/* individuals per dword = per staged batch */
	constexpr int G = 4 * PERWORD;
	constexpr unsigned MASK = (1u << BITS) - 1u;
	constexpr int ROW = CSPLIT * CQS;		/* LDS doubles per individual */
	constexpr int NST = (PERWORD * K + MCHIP_BLOCK - 1) / MCHIP_BLOCK;	/* staged doubles per thread */
	__shared__ __attribute__((aligned(16))) double qs[2][PERWORD * ROW];
	const int part = threadIdx.x % CSPLIT, k0 = part * CKS;
	const int c_raw = blockIdx.x * (MCHIP_BLOCK / CSPLIT) + threadIdx.x / CSPLIT;
	const bool valid = c_raw < a.T;
	const int c = valid ? c_raw : a.T - 1;
	double p[CKSE], acc[CKSE];
#pragma unroll
	for (int k = 0; k < CKSE; k++) {
		p[k] = (k < CKS && k0 + k < K) ? a.P[(size_t)c * K + k0 + k] : 0.0;
		acc[k] = 0.0;
	}
	const int i0 = blockIdx.y * a.ichunk;		/* multiple of G */
	const int i1 = min(a.I, i0 + a.ichunk);
	const int g_end = (i1 + G - 1) / G;
	/* the padding slots multiply p = 0: they must hold finite values, not whatever the last kernel left in LDS */
	{
		double *flat = &qs[0][0];	/* both buffers, as the one array they are */
		for (int x = threadIdx.x; x < 2 * PERWORD * ROW; x += MCHIP_BLOCK) flat[x] = 0.0;
	}
	__syncthreads();
	auto q_address = [&](int x, int ibase) __attribute__((always_inline)) {	/* element x of a batch: individual x / K, cluster x % K */
		const int j = x / K, k = x % K;
		return a.Q + (size_t)min(ibase + j, a.I - 1) * a.qstride + k;
	};
	auto q_slot = [&](int x) __attribute__((always_inline)) { const int j = x / K, k = x % K; return j * ROW + (k / CKS) * CQS + (k % CKS); };
	for (int y = 0; y < NST; y++) {
		const int x = threadIdx.x + y * MCHIP_BLOCK;
		if (x < PERWORD * K) qs[0][q_slot(x)] = *q_address(x, i0);
	}
	const uint4 *gt = reinterpret_cast<const uint4 *>(a.gtC);
	uint4 w = gt[(size_t)(i0 / G) * a.T + c];
	__syncthreads();
	int buf = 0;
	for (int g = i0 / G; g < g_end; g++) {
		const uint4 wn = gt[(size_t)min(g + 1, g_end - 1) * a.T + c];	/* prefetch (clamped) */
#pragma unroll 1
		for (int wi = 0; wi < 4; wi++) {
			const unsigned word = (wi == 0) ? w.x : (wi == 1) ? w.y : (wi == 2) ? w.z : w.w;
			const int ibase = g * G + wi * PERWORD;
			if (ibase >= i1) break;		/* block-uniform: the remaining batches of the chunk's last word are empty */
			/* the next batch's q rows: requested now, stored behind the arithmetic */
			const int inext = ibase + PERWORD;
			const bool more = inext < i1;	/* block-uniform */
			double st[NST];
			if (more) {
#pragma unroll
				for (int y = 0; y < NST; y++) {
					const int x = threadIdx.x + y * MCHIP_BLOCK;
					st[y] = *q_address(min(x, PERWORD * K - 1), inext);
				}
			}
			const double *rowp = &qs[buf][part * CQS];
#pragma unroll 1
			for (int j = 0; j < PERWORD; j++) {
				if (ibase + j >= i1) break;	/* wave-uniform; padded individuals have zero counts anyway */
				const double2 *qr = reinterpret_cast<const double2 *>(rowp + j * ROW);
				double q[CKSE];
#pragma unroll
				for (int k = 0; k < CKSE / 2; k++) {
					const double2 v = qr[k];
					q[2 * k] = v.x;
					q[2 * k + 1] = v.y;
				}
				double t = q[0] * p[0];
#pragma unroll
				for (int k = 1; k < CKSE; k++) t = __builtin_fma(q[k], p[k], t);
				t = quad_sum(t);
				const double n = (double)((word >> (BITS * j)) & MASK);
				/* SAFE: t may be 0 where n = 0 (projection off: a column whose P is 0 for every k) */
				const double r = SAFE ? ((n != 0.0) ? n * rcp_full(t) : 0.0) : n * rcp_full(t);
#pragma unroll
				for (int k = 0; k < CKSE; k++) acc[k] = __builtin_fma(q[k], r, acc[k]);
			}
			if (more) {
#pragma unroll
				for (int y = 0; y < NST; y++) {
					const int x = threadIdx.x + y * MCHIP_BLOCK;
					if (x < PERWORD * K) qs[buf ^ 1][q_slot(x)] = st[y];
				}
			}
			__syncthreads();	/* the next batch is complete, this one may be overwritten */
			buf ^= 1;
		}
		w = wn;
	}
	if (valid) {
		double *out = a.Apart + ((size_t)blockIdx.y * a.T + c) * K + k0;
#pragma unroll
		for (int k = 0; k < CKS; k++)
			if (k0 + k < K) out[k] = acc[k];
	}
}

/* ---------------------------------------------------------------- individual pass */
constexpr int QBLOCK = mchip_qblock(K);

template <int PL>
__global__ __launch_bounds__(QBLOCK) void k_individual_pass(mchip_pass_args a)
{
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	if (a.skip_ind && *a.skip_ind) return;	/* sums of these parameters are already in Spart */
	const int i_raw = blockIdx.x * QBLOCK + threadIdx.x;
	const bool active = i_raw < a.I;
	const int i = active ? i_raw : a.I - 1;
	const int pl = PL ? PL : a.ploidy;
	double q[K], acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) {
		q[k] = a.Q[(size_t)i * a.qstride + k];
		acc[k] = 0.0;
	}
	const int l0 = blockIdx.y * a.lchunk;
	const int l1 = min(a.L, l0 + a.lchunk);
	const int lb_end = (l1 + 7) >> 3;
	geno_group<PL> g, gn;
	g.load(a.gtS, (size_t)(l0 >> 3) * a.I + i, pl);
	for (int lb = l0 >> 3; lb < lb_end; lb++) {
		gn.load(a.gtS, (size_t)min(lb + 1, lb_end - 1) * a.I + i, pl);
#pragma unroll
		for (int j = 0; j < 8; j++) {
			const int l = lb * 8 + j;
			if (l >= l1) break;			/* wave-uniform */
			const int c0 = a.toff[l];
			const int M = a.ua[l];
			/* columns of one locus: P rows are contiguous ([T][K]); the next column's row is requested
			 * (scalar loads) before this column's arithmetic so its latency hides under it */
			double pn[K];
#pragma unroll
			for (int k = 0; k < K; k++) pn[k] = a.P[(size_t)c0 * K + k];
			for (int m = 0; m < M; m++) {
				double pc[K];
#pragma unroll
				for (int k = 0; k < K; k++) pc[k] = pn[k];
				const int cn = min(c0 + m + 1, a.T - 1);
#pragma unroll
				for (int k = 0; k < K; k++) pn[k] = a.P[(size_t)cn * K + k];
				const int n = active ? g.count(j, (unsigned)m, pl) : 0;
				if (__ballot(n > 0) == 0ull) continue;	/* nobody in this wave carries allele m */
				double t = q[0] * pc[0];
#pragma unroll
				for (int k = 1; k < K; k++) t = __builtin_fma(q[k], pc[k], t);
				const double r = n ? (double)n * rcp_full(t) : 0.0;	/* lanes that do not carry m: exactly 0, also when t = 0 */
#pragma unroll
				for (int k = 0; k < K; k++) acc[k] = __builtin_fma(pc[k], r, acc[k]);
			}
		}
		g = gn;
	}
	if (active) {
		double *out = a.Spart + ((size_t)blockIdx.y * a.I + i) * K;
#pragma unroll
		for (int k = 0; k < K; k++) out[k] = acc[k];
	}
}

/* ---------------------------------------------------------------- individual pass, sparse form
 * lane = individual; only the alleles the individual carries are visited: one term per allele copy
 * (a homozygote contributes its allele twice, which is the reference's n = 2 cell: same t, r and t^2).
 * The P rows of the current block of 8 loci are staged in LDS (double-buffered) and gathered per lane with
 * ds_read_b128; rows of one locus are consecutive, so lanes on different alleles hit different bank groups.
 * Produces the S-side sums and the log likelihood (one multiply per copy, no selects).  ACCUM = false is the
 * stand-alone log-likelihood pass (logL_admixture, log_likelihood.c:96-147). */
/* LDS row stride in doubles (mchip_internal.h) and the lane split of the sparse individual pass: SPLIT lanes per individual,
 * lane part s holds k in [s * KSP, (s + 1) * KSP) (KS values, KSP = KS rounded up to even; slots past K hold zeros) */
constexpr int KP = mchip_kp(K);
constexpr int SPLIT = mchip_ind_split(K);
constexpr int KS = (K + SPLIT - 1) / SPLIT;
constexpr int KSP = SPLIT == 1 ? K : ((KS + 1) & ~1);	/* doubles a lane works on */
constexpr int KGP = SPLIT == 1 ? KP / 2 : KSP / 2;	/* 16-byte LDS reads per gathered row part */
constexpr int PSTR = mchip_ind_pstride(K);		/* LDS doubles from one lane part of a row to the next (KSP, or padded: bank conflicts) */
/* where cluster k of a staged row sits in LDS */
__device__ __forceinline__ constexpr int lds_k(int k) { return SPLIT == 1 ? k : (k / KSP) * PSTR + (k % KSP); }
static_assert(SPLIT * PSTR <= KP || SPLIT == 1, "row stride holds every lane's range");

/* sum over the SPLIT adjacent lanes of an individual; every lane ends with the same bits (a + b = b + a at every level) */
__device__ __forceinline__ double split_sum(double v)
{
	if (SPLIT >= 2) v += __shfl_xor(v, 1);
	if (SPLIT >= 4) v += __shfl_xor(v, 2);
	return v;
}

#ifndef MCHIP_SPARSE_WAVES
#define MCHIP_SPARSE_WAVES 1
#endif
/* Keeps a running product of t values away from underflow without a log in the loop: below 1e-100 its binary
 * exponent moves into an integer (v_frexp_exp_i32_f64 / v_frexp_mant_f64); log(product) = ex*ln2 + log(prod). */
__device__ __forceinline__ void rescale(double &prod, int &ex)
{
	if (prod < 1e-100) {
		ex += __builtin_amdgcn_frexp_exp(prod);
		prod = __builtin_amdgcn_frexp_mant(prod);
	}
}

/* DUAL (with ACCUM, diploid, !SAFE): the same pass also takes the log likelihood of a second parameter set
 * (a.Q2, a.P2 -> a.llpart2), with the arithmetic of the ACCUM = false instance, lane for lane: an accelerated cycle needs
 * log L of its second EM iterate and the E step of the extrapolated point back to back (accel_em.c:53,544); the first is
 * bound by the LDS gather, the second by FP64 issue, and one kernel overlaps some of the two: 2.78-2.88 ms against 1.84 + 1.15
 * at config 3 (scripts/diag/dual.sh).  Tetraploid data loses (1.53 against 0.90 + 0.55 ms at config 5's shape: 170 registers,
 * three waves per SIMD): not instantiated */
/* minimum waves per SIMD the register allocation has to leave room for.  K = 57 ... 64 (four lanes per individual, 16 doubles of
 * q, of the sums and of each gathered row per lane): left alone the diploid S-side instance takes 174 registers = two waves per
 * SIMD, where K = 56 runs three on 150; held to three waves it spills a handful of registers to scratch in the block head
 * and runs 5.3-5.5 -> 4.3-4.6 ms (profiles/r03_k_sweep.txt) */
constexpr int sparse_min_waves(int PL, bool ACCUM)
{
	return MCHIP_SPARSE_WAVES > 1 ? MCHIP_SPARSE_WAVES : ((SPLIT == 4 && KSP == 16 && PL == 2 && ACCUM) ? 3 : 1);
}
template <int PL, bool ACCUM, bool SAFE, bool NOMISS, bool DUAL = false>
__global__ __launch_bounds__(QBLOCK, sparse_min_waves(PL, ACCUM)) void k_individual_sparse(mchip_pass_args a)
{
	static_assert(!DUAL || (ACCUM && !SAFE && PL == 2), "dual pass: ACCUM, shared reciprocals, diploid");
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	if (ACCUM && !DUAL && a.skip_ind && *a.skip_ind) return;	/* sums + logL partials of these parameters are already there */
	extern __shared__ __attribute__((aligned(16))) double lds[];	/* [2][tile_cols][KP] (DUAL: twice) then [QBLOCK] reduction scratch */
	double *lds2 = lds + 2 * (size_t)a.tile_cols * KP;
	double *red = lds + (DUAL ? 4 : 2) * (size_t)a.tile_cols * KP;
#ifdef MCHIP_EXP_SCATTER
	/* EXPERIMENT (never in the product build; scripts/diag/scatter_exp.sh): what it would cost this pass to also form the
	 * N-side sums by scattering q_ik * r into per-workgroup LDS accumulators [column of the tile][K] with hardware
	 * ds_add_f64 -- the "(f)" alternative of DESIGN.md 4.3.  Only the atomics are issued (the accumulators are never flushed to
	 * memory, which would cost more): a lower bound on that design's S-side pass. */
	double *nacc = red + QBLOCK;
	for (int x = threadIdx.x; x < a.tile_cols * KP; x += QBLOCK) nacc[x] = 0.0;
#endif
	static_assert(!DUAL || SPLIT == 1, "dual pass: one lane per individual");
	const int part = SPLIT == 1 ? 0 : (int)(threadIdx.x % SPLIT);	/* which k range this lane holds */
	const int k0 = part * KSP;		/* this lane's first cluster */
	const int kl0 = part * PSTR;		/* ... and where its part of a staged row starts in LDS */
	const int i_raw = blockIdx.x * (QBLOCK / SPLIT) + threadIdx.x / SPLIT;
	const bool active = i_raw < a.I;
	const int i = active ? i_raw : a.I - 1;
	const int pl = PL ? PL : a.ploidy;
	double q[KSP], acc[KSP], q2[DUAL ? K : 1];
#pragma unroll
	for (int k = 0; k < KSP; k++) {
		q[k] = (SPLIT == 1 || k0 + k < K) ? a.Q[(size_t)i * a.qstride + k0 + k] : 0.0;
		acc[k] = 0.0;
		if constexpr (DUAL) q2[k] = a.Q2[(size_t)i * a.qstride + k];
	}
	if constexpr (SPLIT > 1 || !NOMISS) {
		/* once per workgroup: every tile slot holds a finite value from here on.  SPLIT > 1: the row slots past K are read by the
		 * last lane part (staging writes k < K only).  Missing data: a missing copy borrows column 0 of its locus and multiplies
		 * it by r = 0; at a locus WITHOUT any allele column (every individual missing: uniquealleles = 0, golden allmiss_*) that
		 * row is the next locus's, or, behind the tile's last column, whatever the previous kernel left in LDS -- 0 x NaN */
		for (int x = threadIdx.x; x < (DUAL ? 4 : 2) * a.tile_cols * KP; x += QBLOCK) lds[x] = 0.0;
		__syncthreads();
	}
	const int l0 = blockIdx.y * a.lchunk;
	const int l1 = min(a.L, l0 + a.lchunk);
	const int lb0 = l0 >> 3, lb_end = (l1 + 7) >> 3;
	double prod = 1.0, prod2 = 1.0;
	int blk = 0, ex = 0, ex2 = 0;
	constexpr int STAGE = 2;
	const bool staged = a.tile_cols * K <= STAGE * QBLOCK;	/* wave-uniform */

	/* column offsets of the current block's and the next block's loci: E[x] = toff[8 lb + x], x = 0..16 (toff is padded by 24
	 * entries equal to T).  Its upper half for the NEXT block is requested at the head of this one and moved in at its end: the
	 * head of a block -- right behind the workgroup barrier, where the waves of a workgroup all stand at the same place -- then
	 * starts its prefetches at once instead of behind three dependent scalar loads (c_lo, then n_lo, then n_hi) */
	int E[17];
#pragma unroll
	for (int x = 0; x < 17; x++) E[x] = a.toff[lb0 * 8 + x];
	/* stage the first tile */
	{
		const int c_lo = E[0], c_hi = E[8];
		const int nel = (c_hi - c_lo) * K;
		for (int x = threadIdx.x; x < nel; x += QBLOCK) {
			lds[(x / K) * KP + lds_k(x % K)] = a.P[(size_t)c_lo * K + x];
			if (DUAL) lds2[(x / K) * KP + lds_k(x % K)] = a.P2[(size_t)c_lo * K + x];
		}
	}
	geno_group<PL> g, gn;
	g.load(a.gtS, (size_t)lb0 * a.I + i, pl);
	__syncthreads();
	for (int lb = lb0; lb < lb_end; lb++) {
		const int buf = (lb - lb0) & 1;
		const double *tile = lds + (size_t)buf * a.tile_cols * KP;
		const double *tile2 = lds2 + (size_t)buf * a.tile_cols * KP;
		const int c_lo = E[0];
		int En[8];
#pragma unroll
		for (int x = 0; x < 8; x++) En[x] = a.toff[(lb + 1) * 8 + 9 + x];
		/* prefetch the next tile and the next genotype group.  Tiles of up to STAGE * QBLOCK doubles (every data set with
		 * <= 4 alleles per locus at K <= 8) wait in registers while this block is computed and go to the other LDS buffer
		 * after it, so no s_waitcnt for them sits in front of the arithmetic; larger tiles are copied through at once */
		double stage[STAGE], stage2[DUAL ? STAGE : 1];
		int nel_next = 0;
		double *dst = lds + (size_t)(buf ^ 1) * a.tile_cols * KP;
		double *dst2 = lds2 + (size_t)(buf ^ 1) * a.tile_cols * KP;
		if (lb + 1 < lb_end) {
			const int n_lo = E[8], n_hi = E[16];
			nel_next = (n_hi - n_lo) * K;
			const double *src = a.P + (size_t)n_lo * K;
			const double *src2 = DUAL ? a.P2 + (size_t)n_lo * K : nullptr;
			if (staged) {
#pragma unroll
				for (int s2 = 0; s2 < STAGE; s2++) {
					const int x = threadIdx.x + s2 * QBLOCK;
					stage[s2] = src[min(x, nel_next - 1)];
					if constexpr (DUAL) stage2[s2] = src2[min(x, nel_next - 1)];
				}
			} else {
				for (int x = threadIdx.x; x < nel_next; x += QBLOCK) {
					dst[(x / K) * KP + lds_k(x % K)] = src[x];
					if (DUAL) dst2[(x / K) * KP + lds_k(x % K)] = src2[x];
				}
			}
		}
		gn.load(a.gtS, (size_t)min(lb + 1, lb_end - 1) * a.I + i, pl);
		int tb[8];
#pragma unroll
		for (int j = 0; j < 8; j++) tb[j] = E[j];
		/* one locus: all of its copies; a lambda so that full blocks run as straight-line code (no per-locus bound check:
		 * the scheduler can then start the next locus's LDS reads under this locus's arithmetic) */
		auto one_locus = [&](int j) __attribute__((always_inline)) {
			const int base = tb[j] - c_lo;
			if constexpr (PL == 2 || PL == 4) {
				/* all copies of the locus first (t), then one shared reciprocal for each pair of copies */
				double pc[PL ? PL : 1][2 * KGP], t[PL ? PL : 1];
				bool miss[PL ? PL : 1];
				unsigned row[PL ? PL : 1];
#pragma unroll
				for (int b = 0; b < PL; b++) {
					const unsigned mraw = g.copy(j, b, pl);
					/* NOMISS: the data set has no missing copy: no selects (idle lanes duplicate individual I-1) */
					miss[b] = NOMISS ? false : ((mraw == MCHIP_MISSING) || !active);
					const unsigned mm = miss[b] ? 0u : mraw;
					row[b] = (unsigned)base + mm;
					/* 16-byte LDS reads (ds_read_b128): twice the bytes per LDS cycle of ds_read2_b64 */
					const double2 *pr = reinterpret_cast<const double2 *>(tile + (size_t)(base + (int)mm) * KP + kl0);
#pragma unroll
					for (int k = 0; k < KGP; k++) {
						const double2 v = pr[k];
						pc[b][2 * k] = v.x;
						pc[b][2 * k + 1] = v.y;
					}
					t[b] = q[0] * pc[b][0];
#pragma unroll
					for (int k = 1; k < KSP; k++) t[b] = __builtin_fma(q[k], pc[b][k], t[b]);
					t[b] = split_sum(t[b]);	/* the lanes' partial dot products; identical in all of them afterwards */
					/* a missing copy borrowed column 0 of its locus; it takes t = 1 so that it leaves the log-product and
					 * its partner's shared reciprocal alone, whatever column 0 holds (all zeros when projection is off and
					 * a bootstrap replicate lacks that allele) */
					if (!NOMISS) t[b] = miss[b] ? 1.0 : t[b];
				}
				if constexpr (PL == 4 && !SAFE && ACCUM) {
					/* tetraploid: the four copies of the locus share one reciprocal (rcp4's scheme: 16 issue slots per
					 * locus instead of 20 for two pairs); the log-product takes the same two pair products as below */
					const double p01 = t[0] * t[1], p23 = t[2] * t[3];
					const double rp = rcp_full(p01 * p23);
					const double r01 = rp * p23, r23 = rp * p01;
					const double r[4] = { miss[0] ? 0.0 : r01 * t[1], miss[1] ? 0.0 : r01 * t[0],
							      miss[2] ? 0.0 : r23 * t[3], miss[3] ? 0.0 : r23 * t[2] };
#pragma unroll
					for (int b = 0; b < 4; b++)
#pragma unroll
						for (int k = 0; k < KSP; k++) acc[k] = __builtin_fma(pc[b][k], r[b], acc[k]);
					prod *= p01;
					prod *= p23;
				} else
#pragma unroll
				for (int b = 0; b < PL; b += 2) {
					const double pp = t[b] * t[b + 1];
					if (ACCUM) {
						double r0, r1;
						if (SAFE) {	/* tiny lower bounds: the product of two t's may underflow */
							r0 = miss[b] ? 0.0 : rcp_full(t[b]);
							r1 = miss[b + 1] ? 0.0 : rcp_full(t[b + 1]);
						} else {
							const double rp = rcp_full(pp);
							r0 = miss[b] ? 0.0 : rp * t[b + 1];
							r1 = miss[b + 1] ? 0.0 : rp * t[b];
						}
#pragma unroll
						for (int k = 0; k < KSP; k++) acc[k] = __builtin_fma(pc[b][k], r0, acc[k]);
#pragma unroll
						for (int k = 0; k < KSP; k++) acc[k] = __builtin_fma(pc[b + 1][k], r1, acc[k]);
#ifdef MCHIP_EXP_SCATTER
#pragma unroll
						for (int k = 0; k < K; k++) unsafeAtomicAdd(&nacc[(size_t)row[b] * KP + k], q[k] * r0);
#pragma unroll
						for (int k = 0; k < K; k++) unsafeAtomicAdd(&nacc[(size_t)row[b + 1] * KP + k], q[k] * r1);
#endif
					}
					if (SAFE) {
						/* projection off: an extrapolated point may have entries outside [0, 1] and t < 0 in a cell that carries a
						 * copy; the reference's log(t) is NaN there and ends the run (em_alg.c:106-110).  Two such factors
						 * must not cancel in the product: a negative t poisons it */
						prod *= t[b] < 0.0 ? __builtin_nan("") : t[b];
						rescale(prod, ex);
						prod *= t[b + 1] < 0.0 ? __builtin_nan("") : t[b + 1];
						rescale(prod, ex);
					} else {
						prod *= pp;
					}
					if constexpr (DUAL) {	/* the second set's t of the same two copies: gather, dot product, into its own log-product
								 * (requesting these rows ahead of the accumulation above costs 30 more registers and a wave
								 * of occupancy: measured slower) */
						double t2[2];
#pragma unroll
						for (int bb = 0; bb < 2; bb++) {
							const double2 *pr = reinterpret_cast<const double2 *>(tile2 + (size_t)row[b + bb] * KP);
							double pc2[KP];
#pragma unroll
							for (int k = 0; k < KP / 2; k++) {
								const double2 v = pr[k];
								pc2[2 * k] = v.x;
								pc2[2 * k + 1] = v.y;
							}
							t2[bb] = q2[0] * pc2[0];
#pragma unroll
							for (int k = 1; k < K; k++) t2[bb] = __builtin_fma(q2[k], pc2[k], t2[bb]);
							if (!NOMISS) t2[bb] = miss[b + bb] ? 1.0 : t2[bb];
						}
						prod2 *= t2[0] * t2[1];
					}
				}
			} else {
				for (int bb = 0; bb < pl; bb++) {	/* any other ploidy: one copy at a time */
					const unsigned mraw = g.copy(j, bb, pl);
					const bool miss = (mraw == MCHIP_MISSING) || !active;
					const unsigned mm = miss ? 0u : mraw;
					const double *pr = tile + (size_t)(base + (int)mm) * KP + kl0;
					double pc[KSP];
#pragma unroll
					for (int k = 0; k < KSP; k++) pc[k] = pr[k];
					double t = q[0] * pc[0];
#pragma unroll
					for (int k = 1; k < KSP; k++) t = __builtin_fma(q[k], pc[k], t);
					t = split_sum(t);
					if (ACCUM) {
						const double r = miss ? 0.0 : rcp_full(t);
#pragma unroll
						for (int k = 0; k < KSP; k++) acc[k] = __builtin_fma(pc[k], r, acc[k]);
					}
					prod *= miss ? 1.0 : (t < 0.0 ? __builtin_nan("") : t);		/* as above: a negative t must end in NaN */
					rescale(prod, ex);
				}
			}
		};
		/* !SAFE: the product is looked at every flush_blocks units of 16 multiplications: a block of 8 diploid loci,
		 * half a block of tetraploid ones (the host sizes flush_blocks for that, mchip_set_model) */
		auto tick = [&]() __attribute__((always_inline)) {
			if (!SAFE) {
				if (++blk >= a.flush_blocks) {
					blk = 0;
					rescale(prod, ex);
					if (DUAL) rescale(prod2, ex2);
				}
			}
		};
		/* unrolled with a scalar bound check per locus: static j keeps the genotype group and the offsets in registers (a
		 * dynamic index would put them in scratch, whose s_waitcnt would also wait for the prefetches) */
		const int nloc = l1 - lb * 8;
		/* (diploid: taking two loci at a time so that four copies share one reciprocal, as the tetraploid locus does, needs the
		 * four gathered rows at once -- 138 registers, three waves per SIMD -- and measured 8-10 % slower than this) */
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (j < nloc) one_locus(j);
			if (PL == 4 && j == 3) tick();
		}
		tick();
		if (staged) {
#pragma unroll
			for (int s2 = 0; s2 < STAGE; s2++) {
				const int x = threadIdx.x + s2 * QBLOCK;
				if (x < nel_next) dst[(x / K) * KP + lds_k(x % K)] = stage[s2];
				if constexpr (DUAL) { if (x < nel_next) dst2[(x / K) * KP + lds_k(x % K)] = stage2[s2]; }
			}
		}
		g = gn;
#pragma unroll
		for (int x = 0; x <= 8; x++) E[x] = E[x + 8];
#pragma unroll
		for (int x = 0; x < 8; x++) E[9 + x] = En[x];
		__syncthreads();	/* next tile is complete and this one may be overwritten */
	}
	const double ll = (double)ex * 0.693147180559945309417 + log(prod);
	if (ACCUM && active) {
		double *out = a.Spart + ((size_t)blockIdx.y * a.I + i) * K + k0;
#pragma unroll
		for (int k = 0; k < KSP; k++)
			if (SPLIT == 1 || k0 + k < K) out[k] = acc[k];
	}
	/* every lane of an individual carries the same log-product: part 0 speaks for it */
	const double tot = block_sum<QBLOCK>((active && part == 0) ? ll : 0.0, red);
	if (threadIdx.x == 0) a.llpart[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
	if (DUAL) {
		const double ll2 = (double)ex2 * 0.693147180559945309417 + log(prod2);
		__syncthreads();	/* red[] is read by thread 0 above */
		const double tot2 = block_sum<QBLOCK>(active ? ll2 : 0.0, red);
		if (threadIdx.x == 0) a.llpart2[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot2;
	}
}

/* ---------------------------------------------------------------- individual pass, sparse form, cooperating waves (K <= 27)
 * The same lane arithmetic as k_individual_sparse above (which stays for the lane-split K >= 28, whose tiles are too large to be
 * private), another division of labour: a workgroup is a.ind_waves waves over the SAME 64 individuals; wave w takes sub-chunk w
 * (a.lchunk loci) of the workgroup's a.ind_waves * a.lchunk loci, with its OWN double-buffered tile of staged P rows -- the waves
 * of a workgroup are at different loci, nothing in the loop is shared, and there is no workgroup barrier in it; they drift apart
 * and cover each other's block heads.  After the loop the waves hand their sums to wave 0 through LDS, in wave order, and wave 0
 * stores ONE slab of S-side sums and one partial log likelihood: a quarter of the slab traffic of one slab per wave pair
 * (mchip_internal.h: cooperating waves). */
__device__ __forceinline__ double wave_total(double v)
{
	/* fixed tree over the 64 lanes; lane 0 ends with the total */
	for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
	return v;
}

constexpr int WSTAGE = 4;	/* doubles of the next tile a lane keeps in registers: tiles of up to 256 doubles (32 columns at K = 8) */
/* Where a staged row is exactly the K doubles of a P row (KP == K: even K that is no multiple of 16) a tile is a contiguous copy
 * of P[c_lo .. c_hi) and goes from memory straight into LDS (global_load_lds_dwordx4: 64 lanes x 16 bytes per instruction, LDS
 * address = M0 + 16 * lane), with no staging registers, no ds_write and no address arithmetic per element.  The statement is
 * opaque to hipcc: an LDS-DMA it knew about would be waited for in front of the first LDS read that may alias it -- both
 * buffers live in one dynamic array, so that is the block's first gather -- which is the exposed latency the register
 * staging exists to avoid.  The wave waits for its copies itself, once, at the end of the block (s_waitcnt vmcnt(0)). */
#ifdef MCHIP_NO_GLDS
constexpr bool GLDS = false;
#else
constexpr bool GLDS = (KP == K) && (K % 2 == 0) && SPLIT == 1;
#endif
/* LDS doubles from one tile buffer to the next: with the direct copy a buffer is written in whole 1 KiB pieces */
__device__ __host__ __forceinline__ constexpr size_t tile_stride(int tile_cols)
{
	return GLDS ? (((size_t)tile_cols * KP + 127) / 128) * 128 : (size_t)tile_cols * KP;
}
typedef __attribute__((address_space(3))) double lds_double;
__device__ __forceinline__ void glds_copy(const double *src, lds_double *dst, int nel, int lane)
{
	/* nel doubles (even; src 16-byte aligned) to dst[0 .. nel); lanes past the end re-read the last 16 bytes into slots of the
	 * buffer's padding.  No "memory" clobber: it would cover toff[] / Q / P too, and hipcc then reads those through the vector
	 * memory path with a wait at the head of every block instead of through the scalar cache (seen in the ISA); the AMDGPU
	 * backend takes no "m" operands either.  The statements are volatile (kept, and in order among themselves and the wait);
	 * what the compiler has to know about memory it is told by the fence in glds_wait */
	const int units = nel >> 1;
	for (int u0 = 0; u0 < units; u0 += 64) {
		const double *g = src + 2 * (size_t)min(u0 + lane, units - 1);
		const unsigned l = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(dst + 2 * (size_t)u0));
		unsigned keep;
		asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
			     : "=&s"(keep) : "v"(g), "s"(l));
	}
}
/* the wave's copies have landed; the fence (wavefront scope: no instruction) keeps every later LDS read behind this point and
 * tells the optimiser that LDS may have changed */
__device__ __forceinline__ void glds_wait()
{
	asm volatile("s_waitcnt vmcnt(0)");
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <int PL, bool ACCUM, bool SAFE, bool NOMISS, bool DUAL = false>
__global__ __launch_bounds__(64 * MCHIP_IND_WAVES_MAX) void k_individual_sparse_w(mchip_pass_args a)
{
	static_assert(SPLIT == 1, "wave-private tiles: one lane per individual");
	static_assert(!DUAL || (ACCUM && !SAFE && PL == 2), "dual pass: ACCUM, shared reciprocals, diploid");
	if (a.stop && *a.stop) return;		/* batched run already stopped (uniform over the grid) */
	if (ACCUM && !DUAL && a.skip_ind && *a.skip_ind) return;	/* sums + logL partials of these parameters are already there */
	extern __shared__ __attribute__((aligned(16))) double lds[];	/* [wave][buffer][set][tile_cols][KP]; after the loop: [K + 2][64] hand-over */
	constexpr int SETS = DUAL ? 2 : 1;
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int W = a.ind_waves;
	/* which (tile of 64 individuals, slab row) this workgroup is: coop_rows() */
	const int n_tiles = (a.I + 63) / 64;
	const int slot = a.xcd_rows ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
	const int bx = slot % n_tiles, by = a.xcd_rows ? (slot / n_tiles) * 8 + (int)(blockIdx.x & 7) : slot / n_tiles;
	const size_t tile_doubles = tile_stride(a.tile_cols);
	double *mine = lds + (size_t)wv * 2 * SETS * tile_doubles;
	const int i_raw = bx * 64 + lane;
	const bool active = i_raw < a.I;
	const int i = active ? i_raw : a.I - 1;
	const int pl = PL ? PL : a.ploidy;
	double q[K], acc[K], q2[DUAL ? K : 1];
#pragma unroll
	for (int k = 0; k < K; k++) {
		q[k] = a.Q[(size_t)i * a.qstride + k];
		acc[k] = 0.0;
		if constexpr (DUAL) q2[k] = a.Q2[(size_t)i * a.qstride + k];
	}
	const int l0 = (by * W + wv) * a.lchunk;
	const int l1 = min(a.L, l0 + a.lchunk);
	double prod = 1.0, prod2 = 1.0;
	int blk = 0, ex = 0, ex2 = 0;
	if (l0 < a.L) {		/* (wave-uniform; the last workgroup's trailing waves may have nothing) */
		if constexpr (!NOMISS) {
			/* a missing copy borrows column 0 of its locus and multiplies it by r = 0; at a locus WITHOUT any allele column (every
			 * individual missing: uniquealleles = 0, golden allmiss_*) that row is the next locus's, or, behind the tile's last
			 * column, whatever the previous kernel left in LDS -- 0 x NaN.  Once per wave: every slot of its tiles is finite */
			for (size_t x = lane; x < 2 * SETS * tile_doubles; x += 64) mine[x] = 0.0;
		}
		const int lb0 = l0 >> 3, lb_end = (l1 + 7) >> 3;
		const bool staged = a.tile_cols * K <= WSTAGE * 64;	/* wave-uniform */
		/* column offsets of the current block's and the next block's loci: E[x] = toff[8 lb + x], x = 0..16 (toff is padded by 24
		 * entries equal to T); the upper half for the NEXT block is requested at the head of this one and moved in at its end */
		/* (read through the constant address space: toff[] does not change while a kernel runs, and the copy statements above are
		 * opaque to hipcc -- it would otherwise assume they may have written it and re-read it through the vector memory path) */
		const __attribute__((address_space(4))) int *toffc = (const __attribute__((address_space(4))) int *)a.toff;
		int E[17];
#pragma unroll
		for (int x = 0; x < 17; x++) E[x] = toffc[lb0 * 8 + x];
		{	/* stage the first tile */
			const int c_lo = E[0], c_hi = E[8];
			const int nel = (c_hi - c_lo) * K;
			for (int x = lane; x < nel; x += 64) {
				mine[(x / K) * KP + (x % K)] = a.P[(size_t)c_lo * K + x];
				if (DUAL) mine[2 * tile_doubles + (x / K) * KP + (x % K)] = a.P2[(size_t)c_lo * K + x];
			}
		}
		geno_group<PL> g, gn;
		g.load(a.gtS, (size_t)lb0 * a.I + i, pl);
		g.arrive();
		for (int lb = lb0; lb < lb_end; lb++) {
			const int buf = (lb - lb0) & 1;
			const double *tile = mine + (size_t)buf * tile_doubles;
			const double *tile2 = mine + (size_t)(2 + buf) * tile_doubles;
			const int c_lo = E[0];
			int En[8];
#pragma unroll
			for (int x = 0; x < 8; x++) En[x] = toffc[(lb + 1) * 8 + 9 + x];
			/* the next tile: requested now; tiles of up to WSTAGE * 64 doubles wait in registers while this block is computed and go
			 * to the wave's other buffer after it (no s_waitcnt for them in front of the arithmetic); larger ones are copied through */
			double stage[WSTAGE], stage2[DUAL ? WSTAGE : 1];
			int nel_next = 0;
			double *dst = mine + (size_t)(buf ^ 1) * tile_doubles;
			double *dst2 = mine + (size_t)(2 + (buf ^ 1)) * tile_doubles;
			if (lb + 1 < lb_end) {
				const int n_lo = E[8], n_hi = E[16];
				nel_next = (n_hi - n_lo) * K;
				const double *src = a.P + (size_t)n_lo * K;
				const double *src2 = DUAL ? a.P2 + (size_t)n_lo * K : nullptr;
				if constexpr (GLDS) {
					if (nel_next > 0) {
						glds_copy(src, (lds_double *)dst, nel_next, lane);
						if constexpr (DUAL) glds_copy(src2, (lds_double *)dst2, nel_next, lane);
					}
				} else if (staged) {
#pragma unroll
					for (int s2 = 0; s2 < WSTAGE; s2++) {
						const int x = lane + s2 * 64;
						stage[s2] = src[min(x, nel_next - 1)];
						if constexpr (DUAL) stage2[s2] = src2[min(x, nel_next - 1)];
					}
				} else {
					for (int x = lane; x < nel_next; x += 64) {
						dst[(x / K) * KP + (x % K)] = src[x];
						if (DUAL) dst2[(x / K) * KP + (x % K)] = src2[x];
					}
				}
			}
			gn.load(a.gtS, (size_t)min(lb + 1, lb_end - 1) * a.I + i, pl);
			int tb[8];
#pragma unroll
			for (int j = 0; j < 8; j++) tb[j] = E[j];
			auto one_locus = [&](int j) __attribute__((always_inline)) {
				const int base = tb[j] - c_lo;
				if constexpr (PL == 2 || PL == 4) {
					/* all copies of the locus first (t), then one shared reciprocal for each pair of copies */
					double pc[PL ? PL : 1][KP], t[PL ? PL : 1];
					bool miss[PL ? PL : 1];
					unsigned row[PL ? PL : 1];
#pragma unroll
					for (int b = 0; b < PL; b++) {
						const unsigned mraw = g.copy(j, b, pl);
						/* NOMISS: the data set has no missing copy: no selects (idle lanes duplicate individual I-1) */
						miss[b] = NOMISS ? false : ((mraw == MCHIP_MISSING) || !active);
						const unsigned mm = miss[b] ? 0u : mraw;
						row[b] = (unsigned)base + mm;
						/* 16-byte LDS reads (ds_read_b128): twice the bytes per LDS cycle of ds_read2_b64 */
						const double2 *pr = reinterpret_cast<const double2 *>(tile + (size_t)(base + (int)mm) * KP);
#pragma unroll
						for (int k = 0; k < KP / 2; k++) {
							const double2 v = pr[k];
							pc[b][2 * k] = v.x;
							pc[b][2 * k + 1] = v.y;
						}
						t[b] = q[0] * pc[b][0];
#pragma unroll
						for (int k = 1; k < K; k++) t[b] = __builtin_fma(q[k], pc[b][k], t[b]);
						/* a missing copy borrowed column 0 of its locus; it takes t = 1 so that it leaves the log-product and
						 * its partner's shared reciprocal alone, whatever column 0 holds */
						if (!NOMISS) t[b] = miss[b] ? 1.0 : t[b];
					}
					if constexpr (PL == 4 && !SAFE && ACCUM) {
						/* tetraploid: the four copies of the locus share one reciprocal (rcp4's scheme) */
						const double p01 = t[0] * t[1], p23 = t[2] * t[3];
						const double rp = rcp_full(p01 * p23);
						const double r01 = rp * p23, r23 = rp * p01;
						const double r[4] = { miss[0] ? 0.0 : r01 * t[1], miss[1] ? 0.0 : r01 * t[0],
								      miss[2] ? 0.0 : r23 * t[3], miss[3] ? 0.0 : r23 * t[2] };
#pragma unroll
						for (int b = 0; b < 4; b++)
#pragma unroll
							for (int k = 0; k < K; k++) acc[k] = __builtin_fma(pc[b][k], r[b], acc[k]);
						prod *= p01;
						prod *= p23;
					} else
#pragma unroll
					for (int b = 0; b < PL; b += 2) {
						const double pp = t[b] * t[b + 1];
						if (ACCUM) {
							double r0, r1;
							if (SAFE) {	/* tiny lower bounds: the product of two t's may underflow */
								r0 = miss[b] ? 0.0 : rcp_full(t[b]);
								r1 = miss[b + 1] ? 0.0 : rcp_full(t[b + 1]);
							} else {
								const double rp = rcp_full(pp);
								r0 = miss[b] ? 0.0 : rp * t[b + 1];
								r1 = miss[b + 1] ? 0.0 : rp * t[b];
							}
#pragma unroll
							for (int k = 0; k < K; k++) acc[k] = __builtin_fma(pc[b][k], r0, acc[k]);
#pragma unroll
							for (int k = 0; k < K; k++) acc[k] = __builtin_fma(pc[b + 1][k], r1, acc[k]);
						}
						if (SAFE) {
							/* projection off: a negative t in a cell that carries a copy is NaN in the reference's log and ends the
							 * run (em_alg.c:106-110); two such factors must not cancel in the product */
							prod *= t[b] < 0.0 ? __builtin_nan("") : t[b];
							rescale(prod, ex);
							prod *= t[b + 1] < 0.0 ? __builtin_nan("") : t[b + 1];
							rescale(prod, ex);
						} else {
							prod *= pp;
						}
						if constexpr (DUAL) {	/* the second set's t of the same two copies: gather, dot product, into its own log-product */
							double t2[2];
#pragma unroll
							for (int bb = 0; bb < 2; bb++) {
								const double2 *pr = reinterpret_cast<const double2 *>(tile2 + (size_t)row[b + bb] * KP);
								double pc2[KP];
#pragma unroll
								for (int k = 0; k < KP / 2; k++) {
									const double2 v = pr[k];
									pc2[2 * k] = v.x;
									pc2[2 * k + 1] = v.y;
								}
								t2[bb] = q2[0] * pc2[0];
#pragma unroll
								for (int k = 1; k < K; k++) t2[bb] = __builtin_fma(q2[k], pc2[k], t2[bb]);
								if (!NOMISS) t2[bb] = miss[b + bb] ? 1.0 : t2[bb];
							}
							prod2 *= t2[0] * t2[1];
						}
					}
				} else {
					for (int bb = 0; bb < pl; bb++) {	/* any other ploidy: one copy at a time */
						const unsigned mraw = g.copy(j, bb, pl);
						const bool miss = (mraw == MCHIP_MISSING) || !active;
						const unsigned mm = miss ? 0u : mraw;
						const double *pr = tile + (size_t)(base + (int)mm) * KP;
						double pc[K];
#pragma unroll
						for (int k = 0; k < K; k++) pc[k] = pr[k];
						double t = q[0] * pc[0];
#pragma unroll
						for (int k = 1; k < K; k++) t = __builtin_fma(q[k], pc[k], t);
						if (ACCUM) {
							const double r = miss ? 0.0 : rcp_full(t);
#pragma unroll
							for (int k = 0; k < K; k++) acc[k] = __builtin_fma(pc[k], r, acc[k]);
						}
						prod *= miss ? 1.0 : (t < 0.0 ? __builtin_nan("") : t);		/* as above: a negative t must end in NaN */
						rescale(prod, ex);
					}
				}
			};
			/* !SAFE: the product is looked at every flush_blocks units of 16 multiplications: a block of 8 diploid loci,
			 * half a block of tetraploid ones (the host sizes flush_blocks for that, mchip_set_model) */
			auto tick = [&]() __attribute__((always_inline)) {
				if (!SAFE) {
					if (++blk >= a.flush_blocks) {
						blk = 0;
						rescale(prod, ex);
						if (DUAL) rescale(prod2, ex2);
					}
				}
			};
			const int nloc = l1 - lb * 8;
#pragma unroll
			for (int j = 0; j < 8; j++) {
				if (j < nloc) one_locus(j);
				if (PL == 4 && j == 3) tick();
			}
			tick();
			if constexpr (GLDS) {
				glds_wait();	/* this wave's direct copies of the next tile have landed */
			} else if (staged) {
#pragma unroll
				for (int s2 = 0; s2 < WSTAGE; s2++) {
					const int x = lane + s2 * 64;
					if (x < nel_next) dst[(x / K) * KP + (x % K)] = stage[s2];
					if constexpr (DUAL) { if (x < nel_next) dst2[(x / K) * KP + (x % K)] = stage2[s2]; }
				}
			}
			g = gn;
#pragma unroll
			for (int x = 0; x <= 8; x++) E[x] = E[x + 8];
#pragma unroll
			for (int x = 0; x < 8; x++) E[9 + x] = En[x];
			/* (no barrier: the tile is this wave's own, and a wave's LDS operations complete in the order it issued them) */
		}
	}
	double ll = (double)ex * 0.693147180559945309417 + log(prod);
	double ll2 = DUAL ? (double)ex2 * 0.693147180559945309417 + log(prod2) : 0.0;
	if (W > 1) {
		/* waves 1 .. W-1 hand their sums and log likelihoods to wave 0, in wave order, through one wave's worth of LDS taken in turns
		 * (the tiles have served their purpose once every wave has left its loop) */
		__syncthreads();
		for (int x = 1; x < W; x++) {
			if (wv == x) {
#pragma unroll
				for (int k = 0; k < K; k++) lds[k * 64 + lane] = acc[k];
				lds[K * 64 + lane] = ll;
				if (DUAL) lds[(K + 1) * 64 + lane] = ll2;
			}
			__syncthreads();
			if (wv == 0) {
#pragma unroll
				for (int k = 0; k < K; k++) acc[k] += lds[k * 64 + lane];
				ll += lds[K * 64 + lane];
				if (DUAL) ll2 += lds[(K + 1) * 64 + lane];
			}
			__syncthreads();
		}
	}
	if (wv == 0) {
		if (ACCUM && active) {
			double *out = a.Spart + ((size_t)by * a.I + i) * K;
#pragma unroll
			for (int k = 0; k < K; k++) out[k] = acc[k];
		}
		const double tot = wave_total(active ? ll : 0.0);
		if (lane == 0) a.llpart[(size_t)by * n_tiles + bx] = tot;
		if (DUAL) {
			const double tot2 = wave_total(active ? ll2 : 0.0);
			if (lane == 0) a.llpart2[(size_t)by * n_tiles + bx] = tot2;
		}
	}
}

/* ---------------------------------------------------------------- individual pass, every locus biallelic (SNP data)
 * When every locus has exactly two allele columns (diploid data without missing-data phantom slots) column c of locus l is
 * 2l + m and the two rows of a locus are wave-uniform: they come through the scalar cache as SGPR operands of the FMAs, no
 * LDS tile, no per-lane gather.  Both alleles' t are computed for every individual (2K FMAs, the same count as two copies),
 * the genotype only selects which of them enter the log-product, and the S-side sums take P_0 n_0/t_0 + P_1 n_1/t_1 with one
 * reciprocal.  The stand-alone log-likelihood pass, LDS-bound in the general kernel, becomes FP64-bound here. */
template <bool ACCUM, bool NOMISS>
__global__ __launch_bounds__(64 * MCHIP_IND_WAVES_MAX) void k_individual_bial(mchip_pass_args a)
{
	if (a.stop && *a.stop) return;		/* batched run already stopped (uniform over the grid) */
	if (ACCUM && a.skip_ind && *a.skip_ind) return;
	/* cooperating waves, as k_individual_sparse_w: a.ind_waves waves over the same 64 individuals, wave w at sub-chunk w */
	__shared__ double xch[K + 1][64];
	const int lane = threadIdx.x & 63;
	const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
	const int W = a.ind_waves;
	const int i_raw = blockIdx.x * 64 + lane;
	const bool active = i_raw < a.I;
	const int i = active ? i_raw : a.I - 1;
	double q[K], acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) {
		q[k] = a.Q[(size_t)i * a.qstride + k];
		acc[k] = 0.0;
	}
	const int l0raw = (int)(blockIdx.y * W + wv) * a.lchunk;
	const bool any = l0raw < a.L;		/* (wave-uniform; a trailing wave of the last workgroup may have nothing: empty loop) */
	const int l0 = any ? l0raw : 0;
	const int l1 = any ? min(a.L, l0 + a.lchunk) : 0;
	const int lb0 = l0 >> 3, lb_end = any ? (l1 + 7) >> 3 : lb0;
	double prod = 1.0;
	int blk = 0, ex = 0;
	geno_group<2> g, gn;
	g.load(a.gtS, (size_t)lb0 * a.I + i, 2);
	for (int lb = lb0; lb < lb_end; lb++) {
		gn.load(a.gtS, (size_t)min(lb + 1, lb_end - 1) * a.I + i, 2);
		const int nloc = l1 - lb * 8;
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (j < nloc) {		/* wave-uniform */
				const double *__restrict__ r0 = a.P + (size_t)(lb * 8 + j) * 2 * K;	/* rows 2l, 2l+1: wave-uniform, s_load */
				double t0 = q[0] * r0[0], t1 = q[0] * r0[K];
#pragma unroll
				for (int k = 1; k < K; k++) {
					t0 = __builtin_fma(q[k], r0[k], t0);
					t1 = __builtin_fma(q[k], r0[K + k], t1);
				}
				const unsigned ma = g.copy(j, 0, 2), mb = g.copy(j, 1, 2);
				double ta, tb;
				if (NOMISS) {		/* idle lanes duplicate individual I-1 and are dropped at the end */
					ta = ma ? t1 : t0;
					tb = mb ? t1 : t0;
				} else {		/* a missing copy (0xFF) leaves the product alone */
					ta = !active ? 1.0 : (ma == 0u ? t0 : (ma == 1u ? t1 : 1.0));
					tb = !active ? 1.0 : (mb == 0u ? t0 : (mb == 1u ? t1 : 1.0));
				}
				if (ACCUM) {
					const double rp = rcp_full(t0 * t1);
					const double n0 = (double)((int)(ma == 0u) + (int)(mb == 0u)), n1 = (double)((int)(ma == 1u) + (int)(mb == 1u));
					const double R0 = n0 * (rp * t1), R1 = n1 * (rp * t0);
#pragma unroll
					for (int k = 0; k < K; k++) {
						acc[k] = __builtin_fma(r0[k], R0, acc[k]);
						acc[k] = __builtin_fma(r0[K + k], R1, acc[k]);
					}
				}
				prod *= ta * tb;
			}
		}
		if (++blk >= a.flush_blocks) {
			blk = 0;
			rescale(prod, ex);
		}
		g = gn;
	}
	double ll = (double)ex * 0.693147180559945309417 + log(prod);
	for (int x = 1; x < W; x++) {	/* waves 1 .. W-1 hand their sums to wave 0, in wave order */
		if (wv == x) {
#pragma unroll
			for (int k = 0; k < K; k++) xch[k][lane] = acc[k];
			xch[K][lane] = ll;
		}
		__syncthreads();
		if (wv == 0) {
#pragma unroll
			for (int k = 0; k < K; k++) acc[k] += xch[k][lane];
			ll += xch[K][lane];
		}
		__syncthreads();
	}
	if (wv == 0) {
		if (ACCUM && active) {
			double *out = a.Spart + ((size_t)blockIdx.y * a.I + i) * K;
#pragma unroll
			for (int k = 0; k < K; k++) out[k] = acc[k];
		}
		const double tot = wave_total(active ? ll : 0.0);
		if (lane == 0) a.llpart[(size_t)blockIdx.y * gridDim.x + blockIdx.x] = tot;
	}
}

/* ---------------------------------------------------------------- mixture model (em_alg.c:763-1011)
 * E step: v_ik = log eta_k + sum_{l,m: n>0} n log p_klm.  log P is tabulated once per step ([T][K], k_logp in
 * mchip.hip); the per-individual sum is a gather-add over the alleles the individual carries (one add per
 * allele copy; n copies of the same allele add the same value n times), lane = individual, rows staged in LDS
 * exactly like the sparse admixture pass.  Vpart (= Spart) holds the per-locus-chunk sums. */
template <int PL, bool STAGED>
__global__ __launch_bounds__(QBLOCK) void k_mix_gather(mchip_pass_args a)
{
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	extern __shared__ __attribute__((aligned(16))) double lds[];
	const int i_raw = blockIdx.x * QBLOCK + threadIdx.x;
	const bool active = i_raw < a.I;
	const int i = active ? i_raw : a.I - 1;
	const int pl = PL ? PL : a.ploidy;
	double acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) acc[k] = 0.0;
	const int l0 = blockIdx.y * a.lchunk;
	const int l1 = min(a.L, l0 + a.lchunk);
	const int lb0 = l0 >> 3, lb_end = (l1 + 7) >> 3;
	if (STAGED) {
		const int c_lo = a.toff[lb0 * 8], c_hi = a.toff[min(lb0 * 8 + 8, a.L)];
		const int nel = (c_hi - c_lo) * K;
		for (int x = threadIdx.x; x < nel; x += QBLOCK)
			lds[(x / K) * KP + (x % K)] = a.P[(size_t)c_lo * K + x];
		__syncthreads();
	}
	for (int lb = lb0; lb < lb_end; lb++) {
		const int buf = (lb - lb0) & 1;
		const int c_lo = a.toff[lb * 8];
		const double *tile = STAGED ? lds + (size_t)buf * a.tile_cols * KP : a.P + (size_t)c_lo * K;
		if (STAGED && lb + 1 < lb_end) {
			const int n_lo = a.toff[(lb + 1) * 8], n_hi = a.toff[min((lb + 1) * 8 + 8, a.L)];
			const int nel = (n_hi - n_lo) * K;
			double *dst = lds + (size_t)(buf ^ 1) * a.tile_cols * KP;
			for (int x = threadIdx.x; x < nel; x += QBLOCK)
				dst[(x / K) * KP + (x % K)] = a.P[(size_t)n_lo * K + x];
		}
		geno_group<PL> g;
		g.load(a.gtS, (size_t)lb * a.I + i, pl);
		for (int j = 0; j < 8; j++) {
			const int l = lb * 8 + j;
			if (l >= l1) break;
			const int base = a.toff[l] - c_lo;
			for (int b = 0; b < pl; b++) {
				const unsigned mraw = g.copy(j, b, pl);
				if (mraw == MCHIP_MISSING || !active) continue;
				const double *pr = tile + (size_t)(base + (int)mraw) * (STAGED ? KP : K);
#pragma unroll
				for (int k = 0; k < K; k++) acc[k] += pr[k];
			}
		}
		if (STAGED) __syncthreads();
	}
	if (active) {
		double *out = a.Spart + ((size_t)blockIdx.y * a.I + i) * K;
#pragma unroll
		for (int k = 0; k < K; k++) out[k] = acc[k];
	}
}

/* per individual: combine the chunk sums, then
 *   mode 0 (e_step_mixture, em_alg.c:828-882): v = log eta + sum; max; vik = exp(v - max) / sum; ll_i = log(sum) + max
 *   mode 1 (logL_mixture, log_likelihood.c:203-228): v = sum + log eta; scale only if exp(max) under/overflows */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_mix_finalize(int I, int n_lchunks, const double *__restrict__ Vpart,
		const double *__restrict__ eta, double *vik, double *llpart, int mode, const int *stop)
{
	__shared__ double red[MCHIP_BLOCK];
	if (stop && *stop) return;
	const int i = blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	double ll = 0.0;
	if (i < I) {
		double v[K];
		if (mode == 0) {
#pragma unroll
			for (int k = 0; k < K; k++) v[k] = log(eta[k]);
		} else {
#pragma unroll
			for (int k = 0; k < K; k++) v[k] = 0.0;
		}
		for (int ch = 0; ch < n_lchunks; ch++) {
			const double *src = Vpart + ((size_t)ch * I + i) * K;
#pragma unroll
			for (int k = 0; k < K; k++) v[k] += src[k];
		}
		if (mode == 1) {
#pragma unroll
			for (int k = 0; k < K; k++) v[k] += log(eta[k]);
		}
		double mx = -INFINITY;
#pragma unroll
		for (int k = 0; k < K; k++) if (v[k] > mx) mx = v[k];
		if (mode == 0) {
			double temp = 0.0;
#pragma unroll
			for (int k = 0; k < K; k++) {
				v[k] = exp(v[k] - mx);
				temp += v[k];
			}
#pragma unroll
			for (int k = 0; k < K; k++) vik[(size_t)i * K + k] = v[k] / temp;
			ll = log(temp) + mx;
		} else {
			double temp_exp = exp(mx), scale_exp = 0.0;
			/* The reference halves scale_exp until exp() stops overflowing (log_likelihood.c:212-221).  When no v[k] is an
			 * ordinary number -- every one NaN (a negative extrapolated p with the projection off: log() = NaN in k_logp) or
			 * -inf, or one +inf -- mx is infinite, scale_exp = inf, inf / 2 = inf, and its loop never ends: on the CPU a
			 * process that spins, here a wave the stream never gets back from.  Such an individual's term is NaN instead: the
			 * sum is NaN, and stop() ends the run on "nan" (em_alg.c:106-110), which is where a NaN leads the reference whenever
			 * its loop does end.  A finite mx needs at most log2(DBL_MAX / 709) < 1 016 halvings (a dozen for any sum of logs a
			 * data set can produce); the cap states that bound. */
			if (!(mx > -INFINITY && mx < INFINITY)) {
				ll = __builtin_nan("");
			} else {
				if (temp_exp == 0.0 || temp_exp == HUGE_VAL) {
					scale_exp = (temp_exp == HUGE_VAL) ? mx : -mx;
					int halvings = 0;
					do {
						scale_exp *= 0.5;
						temp_exp = exp(scale_exp);
					} while (temp_exp == HUGE_VAL && ++halvings < 2048);
					scale_exp = mx - scale_exp;
#pragma unroll
					for (int k = 0; k < K; k++) v[k] -= scale_exp;
				}
				temp_exp = 0.0;
#pragma unroll
				for (int k = 0; k < K; k++) temp_exp = temp_exp + exp(v[k]);
				ll = log(temp_exp) + scale_exp;
			}
		}
	}
	const double tot = block_sum<MCHIP_BLOCK>(ll, red);
	if (threadIdx.x == 0) llpart[blockIdx.x] = tot;
}

/* M step numerators: sum_i vik * n_ic (em_alg.c:972-986), lane = allele column, vik row wave-uniform */
template <int PL>
__global__ __launch_bounds__(MCHIP_BLOCK) void k_mix_column(mchip_pass_args a)
{
	if (a.stop && *a.stop) return;		/* batched run already stopped (wave-uniform) */
	const int c_raw = blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	const bool valid = c_raw < a.T;
	const int c = valid ? c_raw : a.T - 1;
	const int l = a.col_locus[c];
	const unsigned m = valid ? (unsigned)a.col_allele[c] : 0xFEu;
	const int pl = PL ? PL : a.ploidy;
	double acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) acc[k] = 0.0;
	const int i0 = blockIdx.y * a.ichunk;
	const int i1 = min(a.I, i0 + a.ichunk);
	for (int ib = i0 >> 3; ib < ((i1 + 7) >> 3); ib++) {
		geno_group<PL> g;
		g.load(a.gtA, (size_t)ib * a.L + l, pl);
#pragma unroll 4
		for (int j = 0; j < 8; j++) {
			const int i = min(ib * 8 + j, a.I - 1);
			const double *__restrict__ v = a.Q + (size_t)i * K;	/* vik row, wave-uniform */
			const double n = (double)g.count(j, m, pl);
#pragma unroll
			for (int k = 0; k < K; k++) acc[k] = __builtin_fma(v[k], n, acc[k]);
		}
	}
	if (valid) {
		double *out = a.Apart + ((size_t)blockIdx.y * a.T + c) * K;
#pragma unroll
		for (int k = 0; k < K; k++) out[k] = acc[k];
	}
}

/* ---------------------------------------------------------------- hard partition (first M step)
 * rnd_init.c:456-482: d[i][k][l][m] = 1 for every copy a of allele m assigned to cluster k. */
template <int PL>
__global__ __launch_bounds__(MCHIP_BLOCK) void k_partition_columns(mchip_pass_args a)
{
	const int c_raw = blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	const bool valid = c_raw < a.T;
	const int c = valid ? c_raw : a.T - 1;
	const int l = a.col_locus[c];
	const unsigned m = valid ? (unsigned)a.col_allele[c] : 0xFEu;
	const int pl = PL ? PL : a.ploidy;
	double acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) acc[k] = 0.0;
	const int i0 = blockIdx.y * a.ichunk;
	const int i1 = min(a.I, i0 + a.ichunk);
	for (int ib = i0 >> 3; ib < ((i1 + 7) >> 3); ib++) {
		geno_group<PL> g, s;
		g.load(a.gtA, (size_t)ib * a.L + l, pl);
		s.load(a.asA, (size_t)ib * a.L + l, pl);
#pragma unroll
		for (int j = 0; j < 8; j++) {
			if (a.part_counts) {	/* initialize_parameters_admixture (rnd_init.c:661-684): every copy counts */
				for (int b = 0; b < pl; b++)
					if (g.copy(j, b, pl) == m) {
						const unsigned kb = s.copy(j, b, pl);
#pragma unroll
						for (int k = 0; k < K; k++) acc[k] += (kb == (unsigned)k) ? 1.0 : 0.0;
					}
				continue;
			}
			unsigned long long flags = 0;
			for (int b = 0; b < pl; b++)
				if (g.copy(j, b, pl) == m) flags |= 1ull << (s.copy(j, b, pl) & 63u);
#pragma unroll
			for (int k = 0; k < K; k++) acc[k] += (double)((flags >> k) & 1ull);
		}
	}
	if (valid) {
		double *out = a.Apart + ((size_t)blockIdx.y * a.T + c) * K;
#pragma unroll
		for (int k = 0; k < K; k++) out[k] = acc[k];
	}
}

template <int PL>
__global__ __launch_bounds__(QBLOCK) void k_partition_individuals(mchip_pass_args a)
{
	const int i = blockIdx.x * QBLOCK + threadIdx.x;
	if (i >= a.I) return;
	const int pl = PL ? PL : a.ploidy;
	double acc[K];
#pragma unroll
	for (int k = 0; k < K; k++) acc[k] = 0.0;
	const int l0 = blockIdx.y * a.lchunk;
	const int l1 = min(a.L, l0 + a.lchunk);
	for (int lb = l0 >> 3; lb < ((l1 + 7) >> 3); lb++) {
		geno_group<PL> g, s;
		g.load(a.gtS, (size_t)lb * a.I + i, pl);
		s.load(a.asS, (size_t)lb * a.I + i, pl);
		for (int j = 0; j < 8; j++) {
			if (lb * 8 + j >= l1) break;
			for (int b = 0; b < pl; b++) {
				const unsigned mb = g.copy(j, b, pl), kb = s.copy(j, b, pl);
				if (!a.part_counts) {
					if (mb == MCHIP_MISSING) continue;
					bool dup = false;
					for (int b2 = 0; b2 < b; b2++)
						dup |= (g.copy(j, b2, pl) == mb) && (s.copy(j, b2, pl) == kb);
					if (dup) continue;
				}	/* else initialize_parameters_admixture (rnd_init.c:617-648): every copy of the individual, missing ones too */
#pragma unroll
				for (int k = 0; k < K; k++) acc[k] += (kb == (unsigned)k) ? 1.0 : 0.0;
			}
		}
	}
	double *out = a.Spart + ((size_t)blockIdx.y * a.I + i) * K;
#pragma unroll
	for (int k = 0; k < K; k++) out[k] = acc[k];
}

/* ---------------------------------------------------------------- simplex.c:109-143 on K registers */
__device__ __forceinline__ void michelot_k(double (&x)[K], double mn)
{
	unsigned long long fixed = 0;
	int n = K;
	while (n) {
		double csum = 0.0;
#pragma unroll
		for (int j = 0; j < K; j++) csum += x[j];
		const double shift = (csum - 1.0) / (double)n;
		bool can_terminate = true;
#pragma unroll
		for (int j = 0; j < K; j++)
			if (!((fixed >> j) & 1ull)) {
				x[j] -= shift;
				if (x[j] < mn) {
					x[j] = mn;
					fixed |= 1ull << j;
					n--;
					can_terminate = false;
				}
			}
		if (can_terminate) break;
	}
}

/* Q[to][i][.] = normalise(q_ik * sum_chunks Spart) then project (em_alg.c:685-701); also stores the
 * expected counts S_ik the writers need (write_file.c:359-381). */
constexpr int FQ_LANES = 8, FQ_IND = MCHIP_BLOCK / FQ_LANES;	/* threads per individual / individuals per block of k_finalize_q */
constexpr int FQ_XCH = (K <= 16 ? FQ_LANES - 1 : 1) * K * FQ_IND;	/* doubles of LDS a block exchanges its partial sums through */
__device__ __forceinline__ void finalize_q_body(int block, double *xch_raw, int I, int n_lchunks, const double *__restrict__ Spart,
		const double *__restrict__ Qfrom, int qstride_from, double *Qto, double *sik,
		int do_mstep, int weighted, int do_projection, double lb, double add)
{
	/* FQ_IND individuals per block, FQ_LANES threads each: thread g of an individual adds slabs g, g + 8, ... (four loads in
	 * flight at a time), the partial sums meet in LDS in the order g = 0 .. 7, and thread 0 of the individual carries on.  Every
	 * slab count is handled here: there is no separate slab-sum launch in front (k_sum_slabs took this scheme's place) */
	constexpr int TURNS = K <= 16 ? 1 : FQ_LANES - 1;	/* K <= 16: all partial sums at once (28 KB of LDS at most); above: in turns */
	double (*xch)[K][FQ_IND] = reinterpret_cast<double (*)[K][FQ_IND]>(xch_raw);	/* [K <= 16 ? FQ_LANES - 1 : 1][K][FQ_IND] */
	const int il = threadIdx.x % FQ_IND, g = threadIdx.x / FQ_IND;
	const int i = block * FQ_IND + il;
	const bool active = i < I;
	double s[K];
#pragma unroll
	for (int k = 0; k < K; k++) s[k] = 0.0;
	if (active) {
		const size_t row = (size_t)I * K;
		const double *src = Spart + (size_t)g * row + (size_t)i * K;
		int ch = g;
		for (; ch + 3 * FQ_LANES < n_lchunks; ch += 4 * FQ_LANES, src += 4 * FQ_LANES * row) {
			double v[4][K];
#pragma unroll
			for (int y = 0; y < 4; y++)
#pragma unroll
				for (int k = 0; k < K; k++) v[y][k] = src[(size_t)y * FQ_LANES * row + k];
#pragma unroll
			for (int y = 0; y < 4; y++)
#pragma unroll
				for (int k = 0; k < K; k++) s[k] += v[y][k];
		}
		for (; ch < n_lchunks; ch += FQ_LANES, src += FQ_LANES * row) {
#pragma unroll
			for (int k = 0; k < K; k++) s[k] += src[k];
		}
	}
	if constexpr (TURNS == 1) {
		if (g) {
#pragma unroll
			for (int k = 0; k < K; k++) xch[g - 1][k][il] = s[k];
		}
		__syncthreads();
		if (g == 0) {
#pragma unroll
			for (int x = 0; x < FQ_LANES - 1; x++)
#pragma unroll
				for (int k = 0; k < K; k++) s[k] += xch[x][k][il];
		}
	} else {
		for (int x = 1; x < FQ_LANES; x++) {
			if (g == x) {
#pragma unroll
				for (int k = 0; k < K; k++) xch[0][k][il] = s[k];
			}
			__syncthreads();
			if (g == 0) {
#pragma unroll
				for (int k = 0; k < K; k++) s[k] += xch[0][k][il];
			}
			__syncthreads();
		}
	}
	if (g || !active) return;
	if (weighted) {
#pragma unroll
		for (int k = 0; k < K; k++) s[k] *= Qfrom[(size_t)i * qstride_from + k];
	}
#pragma unroll
	for (int k = 0; k < K; k++) sik[(size_t)i * K + k] = s[k];
	if (!do_mstep) return;
	double temp = 0.0;
#pragma unroll
	for (int k = 0; k < K; k++) {
		s[k] += add;		/* 0, or the 1 every count of initialize_parameters_admixture starts from (rnd_init.c:624) */
		temp += s[k];
	}
	if (temp == 0.0) {
		/* no observed copy at all: 0 / 0 in the reference.  A finite row is kept here (mchip_get_q reports it as the reference has
		 * it); everything this individual's row is multiplied into carries a zero count */
#pragma unroll
		for (int k = 0; k < K; k++) Qto[(size_t)i * K + k] = 1.0 / K;
		return;
	}
#pragma unroll
	for (int k = 0; k < K; k++) s[k] /= temp;
	if (do_projection) michelot_k(s, lb);
#pragma unroll
	for (int k = 0; k < K; k++) Qto[(size_t)i * K + k] = s[k];
}
__global__ __launch_bounds__(MCHIP_BLOCK) void k_finalize_q(int I, int n_lchunks, const double *__restrict__ Spart,
		const double *__restrict__ Qfrom, int qstride_from, double *Qto, double *sik,
		int do_mstep, int weighted, int do_projection, double lb, const int *stop, double add)
{
	__shared__ double xch[FQ_XCH];
	if (stop && *stop) return;		/* (uniform over the grid) */
	finalize_q_body(blockIdx.x, xch, I, n_lchunks, Spart, Qfrom, qstride_from, Qto, sik, do_mstep, weighted, do_projection, lb, add);
}
/* Both finalisers of an EM step in ONE launch: blocks 0 .. q_blocks-1 are k_finalize_q's, the others k_finalize_p_tile's
 * (mchip_finalize.h).  The two read different sums and write different parameters; a small data set's step is five launches of
 * a few microseconds around its two passes, and this is one of them less (and the two run beside each other).  Same bodies,
 * same bits as the two launches. */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_finalize_qp(int q_blocks, int I, int n_lchunks, const double *__restrict__ Spart,
		const double *__restrict__ Qfrom, int qstride_from, double *Qto, double *sik,
		int do_projection_q, double lb_q, mchip_finalize_p_args p, const int *stop)
{
	__shared__ double lds[FQ_XCH > FP_TILE ? FQ_XCH : FP_TILE];
	if (stop && *stop) return;		/* (uniform over the grid) */
	if ((int)blockIdx.x < q_blocks)		/* (uniform over the block) */
		finalize_q_body(blockIdx.x, lds, I, n_lchunks, Spart, Qfrom, qstride_from, Qto, sik, 1, 1, do_projection_q, lb_q, 0.0);
	else
		finalize_p_tile_body((int)blockIdx.x - q_blocks, lds, p.L, p.K, p.T, p.toff, p.loci_per_block, p.n_slabs, p.Apart, p.Pfrom, p.Pto,
				     p.weighted, p.add_lb, p.do_projection, p.lb);
}

/* projection of nrows rows of K (accelerated updates, accel_em.c:510-511; shared eta row) */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_project_q(int nrows, double *Q, double lb, const int *stop)
{
	const int i = blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	if (i >= nrows || (stop && *stop)) return;
	double s[K];
#pragma unroll
	for (int k = 0; k < K; k++) s[k] = Q[(size_t)i * K + k];
	michelot_k(s, lb);
#pragma unroll
	for (int k = 0; k < K; k++) Q[(size_t)i * K + k] = s[k];
}

/* ---------------------------------------------------------------- launchers */
inline dim3 column_grid(const mchip_pass_args &a) { return dim3((a.T + MCHIP_BLOCK - 1) / MCHIP_BLOCK, a.n_ichunks); }
/* packed-count column pass (cooperating waves): 64 columns per workgroup, one slab per workgroup row */
inline dim3 counts_grid(const mchip_pass_args &a) { return dim3((a.T + 63) / 64, mchip_col_slabs(a)); }
inline dim3 indiv_grid(const mchip_pass_args &a) { return dim3((a.I + QBLOCK - 1) / QBLOCK, a.n_lchunks); }
inline dim3 sparse_grid(const mchip_pass_args &a) { return dim3((a.I + QBLOCK / SPLIT - 1) / (QBLOCK / SPLIT), a.n_lchunks); }
/* the cooperating forms (K <= 27): 64 individuals per workgroup, a.ind_waves waves, one slab per workgroup row */
inline dim3 coop_grid(const mchip_pass_args &a) { return dim3((a.I + 63) / 64, mchip_ind_slabs(a)); }
/* The sparse kernel takes a one-dimensional grid and works out (tile, row) itself.  Consecutive workgroup ids go to consecutive
 * XCDs (eight of them, each with its own L2), and every workgroup of a slab row reads the same P rows: with id -> (id / 8 mod
 * tiles, id / 8 / tiles * 8 + id mod 8) a row lives on ONE XCD and its P rows come into one L2 instead of eight (154 MB less
 * fabric traffic per pass at config 3; the pass is not bandwidth-bound, so this is about bytes, not time).  Only where the rows
 * come in whole groups of eight (a.xcd_rows; mchip_set_model rounds the chunk count for that): with 52 rows four XCDs would have
 * seven rows and four six, and the pass ran 5 % longer than with the plain order. */
inline dim3 coop_rows(const mchip_pass_args &a) { return dim3((unsigned)(((a.I + 63) / 64) * mchip_ind_slabs(a))); }
inline size_t coop_lds_bytes(const mchip_pass_args &a, int sets)
{
	const size_t tiles = (size_t)a.ind_waves * 2 * sets * tile_stride(a.tile_cols), hand_over = (size_t)(K + 2) * 64;
	return (tiles > hand_over ? tiles : hand_over) * sizeof(double);
}

/* a.sparse: the column pass only accumulates the N-side sums; S-side sums and logL come from the sparse
 * individual pass.  Otherwise (loci with more alleles than the LDS tile is sized for) the dense pair is used:
 * the column pass also produces logL and the individual pass loops over every allele of every locus. */
void launch_accum_p(const mchip_pass_args &a, hipStream_t s)
{
	if constexpr (CSPLIT > 1) {
		if (a.sparse && a.count_bits && !a.no_col_split) {
			const dim3 grid((a.T + MCHIP_BLOCK / CSPLIT - 1) / (MCHIP_BLOCK / CSPLIT), a.n_ichunks);
			if (a.count_bits == 2 && a.safe_rcp) hipLaunchKernelGGL((k_column_counts_split<2, true>), grid, dim3(MCHIP_BLOCK), 0, s, a);
			else if (a.count_bits == 2) hipLaunchKernelGGL((k_column_counts_split<2, false>), grid, dim3(MCHIP_BLOCK), 0, s, a);
			else if (a.safe_rcp) hipLaunchKernelGGL((k_column_counts_split<4, true>), grid, dim3(MCHIP_BLOCK), 0, s, a);
			else hipLaunchKernelGGL((k_column_counts_split<4, false>), grid, dim3(MCHIP_BLOCK), 0, s, a);
			return;
		}
	}
	if (a.sparse && a.count_bits == 2 && a.safe_rcp) { hipLaunchKernelGGL((k_column_counts<2, false, true>), counts_grid(a), dim3(MCHIP_BLOCK), 0, s, a); return; }
	if (a.sparse && a.count_bits == 4 && a.safe_rcp) { hipLaunchKernelGGL((k_column_counts<4, false, true>), counts_grid(a), dim3(MCHIP_BLOCK), 0, s, a); return; }
	if (a.sparse && a.count_bits == 2) { hipLaunchKernelGGL((k_column_counts<2, false>), counts_grid(a), dim3(MCHIP_BLOCK), 0, s, a); return; }
	if (a.sparse && a.count_bits == 4) { hipLaunchKernelGGL((k_column_counts<4, false>), counts_grid(a), dim3(MCHIP_BLOCK), 0, s, a); return; }
	if (a.sparse) {
		if (a.ploidy == 2) hipLaunchKernelGGL((k_column_pass<2, true, false, false>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
		else hipLaunchKernelGGL((k_column_pass<0, true, false, false>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
		return;
	}
	if (a.ploidy == 2 && a.flush_blocks >= 1) hipLaunchKernelGGL((k_column_pass<2, true, false, true>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else if (a.ploidy == 2) hipLaunchKernelGGL((k_column_pass<2, true, true, true>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else hipLaunchKernelGGL((k_column_pass<0, true, true, true>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
}
#ifdef MCHIP_EXP_SCATTER
inline size_t sparse_lds_bytes(const mchip_pass_args &a) { return (3 * (size_t)a.tile_cols * KP + QBLOCK) * sizeof(double); }
#else
inline size_t sparse_lds_bytes(const mchip_pass_args &a) { return (2 * (size_t)a.tile_cols * KP + QBLOCK) * sizeof(double); }
#endif
template <bool ACCUM> void launch_sparse(const mchip_pass_args &a, hipStream_t s)
{
	const size_t lds = sparse_lds_bytes(a);
	const bool nomiss = !a.has_missing;	/* idle lanes of the last block recompute individual I-1 and are dropped at the end */
	const bool safe = a.flush_blocks < 1;
	/* measured (profiles/r02_biallelic_variant.txt): the stand-alone log-likelihood pass gains 6-21 % from K = 6 up (it is
	 * LDS-bound in the general kernel), the S-side pass only from K = 10 (it is FP64-bound, and the dense-over-two-alleles
	 * form costs more instructions per locus: +8-19 % below that) */
	constexpr bool bial_pays = SPLIT == 1 && (ACCUM ? (K >= 10) : (K >= 6));	/* (its grid and partials are one lane per individual) */
	if constexpr (SPLIT == 1) {
		const dim3 grid = coop_grid(a), rows = coop_rows(a), block(64 * a.ind_waves);
		const size_t cl = coop_lds_bytes(a, 1);
		if (bial_pays && a.biallelic && a.ploidy == 2 && !safe) {
			if (nomiss) hipLaunchKernelGGL((k_individual_bial<ACCUM, true>), grid, block, 0, s, a);
			else hipLaunchKernelGGL((k_individual_bial<ACCUM, false>), grid, block, 0, s, a);
			return;
		}
#define MCHIP_SPARSE_W(PLV) \
		do { \
			if (nomiss && !safe) hipLaunchKernelGGL((k_individual_sparse_w<PLV, ACCUM, false, true>), rows, block, cl, s, a); \
			else if (!safe) hipLaunchKernelGGL((k_individual_sparse_w<PLV, ACCUM, false, false>), rows, block, cl, s, a); \
			else hipLaunchKernelGGL((k_individual_sparse_w<PLV, ACCUM, true, false>), rows, block, cl, s, a); \
		} while (0)
		if (a.ploidy == 2) MCHIP_SPARSE_W(2);
		else if (a.ploidy == 4) MCHIP_SPARSE_W(4);
		else hipLaunchKernelGGL((k_individual_sparse_w<0, ACCUM, true, false>), rows, block, cl, s, a);
#undef MCHIP_SPARSE_W
	} else {
#define MCHIP_SPARSE(PLV) \
		do { \
			if (nomiss && !safe) hipLaunchKernelGGL((k_individual_sparse<PLV, ACCUM, false, true>), sparse_grid(a), dim3(QBLOCK), lds, s, a); \
			else if (!safe) hipLaunchKernelGGL((k_individual_sparse<PLV, ACCUM, false, false>), sparse_grid(a), dim3(QBLOCK), lds, s, a); \
			else hipLaunchKernelGGL((k_individual_sparse<PLV, ACCUM, true, false>), sparse_grid(a), dim3(QBLOCK), lds, s, a); \
		} while (0)
		if (a.ploidy == 2) MCHIP_SPARSE(2);
		else if (a.ploidy == 4) MCHIP_SPARSE(4);
		else hipLaunchKernelGGL((k_individual_sparse<0, ACCUM, true, false>), sparse_grid(a), dim3(QBLOCK), lds, s, a);
#undef MCHIP_SPARSE
	}
}
/* the dual pass exists where both of its halves would run the sparse kernel with shared reciprocals (so that its results are
 * theirs bit for bit) and the second set's registers still fit */
int dual_available(const mchip_pass_args &a)
{
	if (K > 12 || !a.sparse || a.flush_blocks < 1 || a.ploidy != 2) return 0;
	if (a.biallelic && K >= 6) return 0;	/* launch_sparse takes k_individual_bial for one or both passes */
	return coop_lds_bytes(a, 2) <= 64 * 1024;
}
void launch_accum_q_dual(const mchip_pass_args &a, hipStream_t s)
{
	if constexpr (K <= 12) {
		const size_t lds = coop_lds_bytes(a, 2);
		if (!a.has_missing) hipLaunchKernelGGL((k_individual_sparse_w<2, true, false, true, true>), coop_rows(a), dim3(64 * a.ind_waves), lds, s, a);
		else hipLaunchKernelGGL((k_individual_sparse_w<2, true, false, false, true>), coop_rows(a), dim3(64 * a.ind_waves), lds, s, a);
	}
}
/* what the launchers leave behind (mchip_ktable) */
int col_slabs(const mchip_pass_args &a, int mix)
{
	if (mix) return a.count_bits ? mchip_col_slabs(a) : a.n_ichunks;
	return (a.sparse && a.count_bits && (CSPLIT == 1 || a.no_col_split)) ? mchip_col_slabs(a) : a.n_ichunks;
}
int ind_slabs(const mchip_pass_args &a) { return (a.sparse && SPLIT == 1) ? mchip_ind_slabs(a) : a.n_lchunks; }
int ind_ll_parts(const mchip_pass_args &a)
{
	if (a.sparse && SPLIT == 1) return ((a.I + 63) / 64) * mchip_ind_slabs(a);
	if (a.sparse) return ((a.I + QBLOCK / SPLIT - 1) / (QBLOCK / SPLIT)) * a.n_lchunks;
	return ((a.T + MCHIP_BLOCK - 1) / MCHIP_BLOCK) * a.n_ichunks;	/* dense pair: the column pass takes the log likelihood */
}
void launch_loglik(const mchip_pass_args &a, hipStream_t s)
{
	if (a.sparse) { launch_sparse<false>(a, s); return; }
	if (a.ploidy == 2 && a.flush_blocks >= 1) hipLaunchKernelGGL((k_column_pass<2, false, false, true>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else if (a.ploidy == 2) hipLaunchKernelGGL((k_column_pass<2, false, true, true>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else hipLaunchKernelGGL((k_column_pass<0, false, true, true>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
}
void launch_accum_q(const mchip_pass_args &a, hipStream_t s)
{
	if (a.sparse) { launch_sparse<true>(a, s); return; }
	if (a.ploidy == 2) hipLaunchKernelGGL((k_individual_pass<2>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
	else hipLaunchKernelGGL((k_individual_pass<0>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
}
void launch_mix_gather(const mchip_pass_args &a, hipStream_t s)
{
	const size_t lds = 2 * (size_t)a.tile_cols * KP * sizeof(double);
	if (a.sparse) {
		if (a.ploidy == 2) hipLaunchKernelGGL((k_mix_gather<2, true>), indiv_grid(a), dim3(QBLOCK), lds, s, a);
		else hipLaunchKernelGGL((k_mix_gather<0, true>), indiv_grid(a), dim3(QBLOCK), lds, s, a);
	} else {
		if (a.ploidy == 2) hipLaunchKernelGGL((k_mix_gather<2, false>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
		else hipLaunchKernelGGL((k_mix_gather<0, false>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
	}
}
void launch_mix_finalize(int I, int n_lchunks, const double *Vpart, const double *eta, double *vik, double *llpart, int mode, const int *stop, hipStream_t s)
{
	hipLaunchKernelGGL(k_mix_finalize, dim3((I + MCHIP_BLOCK - 1) / MCHIP_BLOCK), dim3(MCHIP_BLOCK), 0, s, I, n_lchunks, Vpart, eta, vik, llpart, mode, stop);
}
void launch_mix_column(const mchip_pass_args &a, hipStream_t s)
{
	if (a.count_bits == 2) { hipLaunchKernelGGL((k_column_counts<2, true>), counts_grid(a), dim3(MCHIP_BLOCK), 0, s, a); return; }
	if (a.count_bits == 4) { hipLaunchKernelGGL((k_column_counts<4, true>), counts_grid(a), dim3(MCHIP_BLOCK), 0, s, a); return; }
	if (a.ploidy == 2) hipLaunchKernelGGL((k_mix_column<2>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else hipLaunchKernelGGL((k_mix_column<0>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
}
void launch_part_p(const mchip_pass_args &a, hipStream_t s)
{
	if (a.ploidy == 2) hipLaunchKernelGGL((k_partition_columns<2>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else if (a.ploidy == 4) hipLaunchKernelGGL((k_partition_columns<4>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
	else hipLaunchKernelGGL((k_partition_columns<0>), column_grid(a), dim3(MCHIP_BLOCK), 0, s, a);
}
void launch_part_q(const mchip_pass_args &a, hipStream_t s)
{
	if (a.ploidy == 2) hipLaunchKernelGGL((k_partition_individuals<2>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
	else if (a.ploidy == 4) hipLaunchKernelGGL((k_partition_individuals<4>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
	else hipLaunchKernelGGL((k_partition_individuals<0>), indiv_grid(a), dim3(QBLOCK), 0, s, a);
}
void launch_finalize_q(int I, int, int n_lchunks, const double *Spart, const double *Qfrom, int qstride_from,
		       double *Qto, double *sik, int do_mstep, int weighted, int do_projection, double lb, const int *stop, hipStream_t s,
		       double add)
{
	hipLaunchKernelGGL(k_finalize_q, dim3((I + FQ_IND - 1) / FQ_IND), dim3(MCHIP_BLOCK), 0, s,
			   I, n_lchunks, Spart, Qfrom, qstride_from, Qto, sik, do_mstep, weighted, do_projection, lb, stop, add);
}
void launch_finalize_qp(int I, int n_lchunks, const double *Spart, const double *Qfrom, int qstride_from, double *Qto, double *sik,
			int do_projection_q, double lb_q, const mchip_finalize_p_args &p, const int *stop, hipStream_t s)
{
	const int q_blocks = (I + FQ_IND - 1) / FQ_IND, p_blocks = (p.L + p.loci_per_block - 1) / p.loci_per_block;
	hipLaunchKernelGGL(k_finalize_qp, dim3(q_blocks + p_blocks), dim3(MCHIP_BLOCK), 0, s,
			   q_blocks, I, n_lchunks, Spart, Qfrom, qstride_from, Qto, sik, do_projection_q, lb_q, p, stop);
}
void launch_project_q(int nrows, int, double *Q, double lb, const int *stop, hipStream_t s)
{
	hipLaunchKernelGGL(k_project_q, dim3((nrows + MCHIP_BLOCK - 1) / MCHIP_BLOCK), dim3(MCHIP_BLOCK), 0, s, nrows, Q, lb, stop);
}

}  // namespace

#define MCHIP_CAT2(a, b) a##b
#define MCHIP_CAT(a, b) MCHIP_CAT2(a, b)
/* a function, not a const global: hipcc would try to emit a const table of host pointers on the device side */
const mchip_ktable *MCHIP_CAT(mchip_ktable_get_, MCHIP_K)()
{
	static mchip_ktable t = {
		launch_accum_p, launch_loglik, launch_accum_q, launch_part_p, launch_part_q, launch_finalize_q, launch_project_q,
		launch_mix_gather, launch_mix_finalize, launch_mix_column, dual_available, launch_accum_q_dual,
		col_slabs, ind_slabs, ind_ll_parts, launch_finalize_qp,
	};
	return &t;
}
