/*
 * mchip.hip -- context management, K-independent kernels and the C-ABI entry points of
 * libmulticlust_hip.so (include/multiclust_hip.h).  gfx950 only; no CPU fallback: every entry point
 * fails with MCHIP_ERR_NO_DEVICE / MCHIP_ERR_HIP when the GPU path cannot run.
 */
#include "mchip_internal.h"
#include "mchip_finalize.h"
#include "mchip_progress.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <unistd.h>

/* ------------------------------------------------------------------ progress record (mchip_progress.h) */
mchip_progress_slot mchip_progress_slots[MCHIP_PROGRESS_SLOTS];
std::atomic<unsigned long long> mchip_progress_events{0};
static std::atomic<int> progress_next_slot{0};

mchip_progress_slot *mchip_progress_my_slot(void)
{
	static thread_local mchip_progress_slot *mine = nullptr;
	if (!mine) {
		int x = progress_next_slot.fetch_add(1, std::memory_order_relaxed);
		if (x >= MCHIP_PROGRESS_SLOTS) x = MCHIP_PROGRESS_SLOTS - 1;	/* more threads than slots: the last one is shared */
		mine = &mchip_progress_slots[x];
		mine->tid.store((long)syscall(SYS_gettid), std::memory_order_relaxed);
	}
	return mine;
}

/* ------------------------------------------------------------------ per-K tables */
#define DECL_KT(n) const mchip_ktable *mchip_ktable_get_##n();
DECL_KT(1) DECL_KT(2) DECL_KT(3) DECL_KT(4) DECL_KT(5) DECL_KT(6) DECL_KT(7) DECL_KT(8)
DECL_KT(9) DECL_KT(10) DECL_KT(11) DECL_KT(12) DECL_KT(13) DECL_KT(14) DECL_KT(15) DECL_KT(16)
DECL_KT(17) DECL_KT(18) DECL_KT(19) DECL_KT(20) DECL_KT(21) DECL_KT(22) DECL_KT(23) DECL_KT(24)
DECL_KT(25) DECL_KT(26) DECL_KT(27) DECL_KT(28) DECL_KT(29) DECL_KT(30) DECL_KT(31) DECL_KT(32)
DECL_KT(33) DECL_KT(34) DECL_KT(35) DECL_KT(36) DECL_KT(37) DECL_KT(38) DECL_KT(39) DECL_KT(40)
DECL_KT(41) DECL_KT(42) DECL_KT(43) DECL_KT(44) DECL_KT(45) DECL_KT(46) DECL_KT(47) DECL_KT(48)
DECL_KT(49) DECL_KT(50) DECL_KT(51) DECL_KT(52) DECL_KT(53) DECL_KT(54) DECL_KT(55) DECL_KT(56)
DECL_KT(57) DECL_KT(58) DECL_KT(59) DECL_KT(60) DECL_KT(61) DECL_KT(62) DECL_KT(63) DECL_KT(64)

const mchip_ktable *mchip_get_ktable(int K)
{
	typedef const mchip_ktable *(*getter)();
	static const getter tabs[MCHIP_MAX_K + 1] = {
		nullptr,
		mchip_ktable_get_1, mchip_ktable_get_2, mchip_ktable_get_3, mchip_ktable_get_4, mchip_ktable_get_5, mchip_ktable_get_6, mchip_ktable_get_7, mchip_ktable_get_8,
		mchip_ktable_get_9, mchip_ktable_get_10, mchip_ktable_get_11, mchip_ktable_get_12, mchip_ktable_get_13, mchip_ktable_get_14, mchip_ktable_get_15, mchip_ktable_get_16,
		mchip_ktable_get_17, mchip_ktable_get_18, mchip_ktable_get_19, mchip_ktable_get_20, mchip_ktable_get_21, mchip_ktable_get_22, mchip_ktable_get_23, mchip_ktable_get_24,
		mchip_ktable_get_25, mchip_ktable_get_26, mchip_ktable_get_27, mchip_ktable_get_28, mchip_ktable_get_29, mchip_ktable_get_30, mchip_ktable_get_31, mchip_ktable_get_32,
		mchip_ktable_get_33, mchip_ktable_get_34, mchip_ktable_get_35, mchip_ktable_get_36, mchip_ktable_get_37, mchip_ktable_get_38, mchip_ktable_get_39, mchip_ktable_get_40,
		mchip_ktable_get_41, mchip_ktable_get_42, mchip_ktable_get_43, mchip_ktable_get_44, mchip_ktable_get_45, mchip_ktable_get_46, mchip_ktable_get_47, mchip_ktable_get_48,
		mchip_ktable_get_49, mchip_ktable_get_50, mchip_ktable_get_51, mchip_ktable_get_52, mchip_ktable_get_53, mchip_ktable_get_54, mchip_ktable_get_55, mchip_ktable_get_56,
		mchip_ktable_get_57, mchip_ktable_get_58, mchip_ktable_get_59, mchip_ktable_get_60, mchip_ktable_get_61, mchip_ktable_get_62, mchip_ktable_get_63, mchip_ktable_get_64,
	};
	static_assert(MCHIP_MAX_K == 64, "one kernel translation unit per K (Makefile: KS)");
	return (K >= 1 && K <= MCHIP_MAX_K) ? tabs[K]() : nullptr;
}

/* ------------------------------------------------------------------ context */
struct mchip_context {
	int device;
	hipStream_t stream;
	char err[512];
	int n_cu;
	/* data set */
	int I, L, ploidy, T, max_M, min_M;
	std::vector<int32_t> h_ua;	/* host copy of uniquealleles: a data set of the same shape reuses every buffer */
	int parked_K;			/* model buffers kept allocated for this K (and the signature below) while no model is set */
	int sig_admixture, sig_constrained, sig_projection, sig_nsec;
	double sig_eta_lb, sig_p_lb;
	int init_geno_set;		/* d_initA / d_initS hold the observed haplotypes (mchip_set_init_genotypes) */
	int32_t *d_ua, *d_toff, *d_col_locus;
	uint8_t *d_col_allele;
	uint8_t *d_gtA, *d_gtS, *d_gtC;
	int count_bits, has_missing;
	int first_empty;	/* first individual without a single observed copy, or -1 */
	std::vector<int> empty_rows;	/* all of them: their mixing proportions are 0 / 0 in the reference (em_alg.c:685-690) */
	int empty_rows_nan[3];	/* per slot: the slot's rows of such individuals stand for the reference's 0 / 0 (an M step wrote the slot, or
				 * something computed from such a slot did, or NaN rows were uploaded): mchip_get_q reports them as NaN */
	unsigned long long nnz_cells, n_copies;	/* cells with n_ic > 0, non-missing allele copies (mchip_data_counts) */
	int counts_valid;
	size_t geno_bytes_A, geno_bytes_S;
	uint8_t *d_asA, *d_asS;		/* hard-partition scratch, allocated on first use */
	uint8_t *d_initA, *d_initS;	/* genotype the hard-partition M step reads when it is not the data set itself (bootstrap) */
	uint8_t *d_draw;		/* device-drawn partition in stream order [I][L][ploidy], padded to whole chunks */
	/* Rand-EM candidates (mchip_init_from_allele_centers), kept from one candidate to the next and grown when needed: the
	 * rand() % K values of a candidate's span of the stream, its center alleles [L][K], its per-locus draw offsets */
	uint8_t *d_cand_span, *d_cand_centers;
	unsigned long long *d_cand_off;
	size_t cand_span_bytes, cand_loci, cand_center_bytes;
	uint32_t *d_jump_hi, *d_jump_lo;	/* jump polynomials of the rand() stream (mchip_mstep_from_rand_partition) */
	size_t n_jump_hi;
	/* jump polynomials of the tiled generators (0: bootstrap data set, 1: random partition): x^(A*i), i < nA, stored [31][nA]
	 * (one per individual) and x^(B*r), r < nB, stored [nB][31] (one per locus tile); kept while A, nA, B, nB stay the same */
	struct lattice { uint32_t *d_i, *d_r; uint64_t A, B; unsigned nA, nB; } lat[2];
	uint32_t *d_part_slabs;		/* tiled random partition: packed 16-bit N-side counts per block of 256 individuals */
	size_t part_slab_bytes;
	/* model */
	int K, admixture, constrained, do_projection, nsec, nq, qstride;
	double eta_lb, p_lb;
	const mchip_ktable *kt;
	double *d_p[3], *d_q[3];
	double *d_up[MCHIP_MAX_SECANTS], *d_vp[MCHIP_MAX_SECANTS], *d_uq[MCHIP_MAX_SECANTS], *d_vq[MCHIP_MAX_SECANTS];
	double *d_sik;			/* [I][K] expected counts / vik */
	double *d_stage;		/* K*T staging for the [K][T] <-> [T][K] transposes */
	double *d_logp;			/* mixture model: log P table [T][K] */
	/* workspaces */
	int ichunk, n_ichunks, lchunk, n_lchunks, n_llpart, flush_blocks, safe_rcp, sparse;
	int ind_waves;			/* waves per workgroup of the cooperating individual-side kernels (mchip_internal.h) */
	int xcd_rows;			/* their slab rows come in whole groups of eight: one row, one XCD */
	/* testing / tuning knobs of the environment (README), read when a context is created and again with every data set and every
	 * model -- never on a launch path */
	struct {
		int no_bial, no_counts, force_dense, force_safe, no_graph, no_dual, no_slab_sum, no_col_split, part_no_tile, sim_no_tile;
		int no_fused_finalize;
		int per_cu_col, per_cu_ind, geometry_given, no_roundup;
		double slab_frac;
	} knob;
	double *d_Apart, *d_Spart, *d_llpart, *d_scalars;	/* d_scalars: [0]=logL, [1..3]=dots, [4..]=eta sums */
	double *d_llpart2;		/* partial log likelihoods of the second parameter set of a dual individual pass */
	double *d_redpart;		/* block partials of the dot products / column sums */
	uint8_t *d_flags;		/* michelot "fixed" flags for loci with more than 64 alleles */
	double *h_pinned;		/* 64 doubles */
	mchip_run_state *d_run;		/* batched-run state (mchip_em_run) */
	hipGraphExec_t step_graph[3];	/* one captured {EM step + stop check} per slot; rebuilt when the model changes */
	hipGraphExec_t cycle_graph[3][5];	/* one captured accelerated cycle per (start slot, scheme) */
	int *d_cyc;			/* batched accelerated runs: [0] no update this cycle, [1] extrapolation accepted */
	int have_ll;
	int ll_parts;			/* partial log likelihoods the last E step left in d_llpart */
	int s_cache_slot;		/* slot whose S-side sums + logL are held in Spart / d_scalars[2] (mchip_loglik_prefetch), or -1 */
	/* profiling */
	int profiling;
	hipEvent_t ev_begin, ev_end;
	std::vector<hipEvent_t> ev_pool;
	std::vector<int> ev_kind;	/* kernel kind of pair p = events 2p, 2p+1 */
	size_t ev_used;
};

static void read_knobs(mchip_context *ctx)
{
	auto on = [](const char *name) { return getenv(name) != nullptr ? 1 : 0; };
	auto num = [](const char *name, int dflt) { const char *e = getenv(name); return (e && atoi(e) > 0) ? atoi(e) : dflt; };
	ctx->knob.no_bial = on("MCHIP_NO_BIAL");
	ctx->knob.no_counts = on("MCHIP_NO_COUNTS");
	ctx->knob.force_dense = on("MCHIP_FORCE_DENSE");
	ctx->knob.force_safe = on("MCHIP_FORCE_SAFE");
	ctx->knob.no_graph = on("MCHIP_NO_GRAPH");
	ctx->knob.no_dual = on("MCHIP_NO_DUAL");
	ctx->knob.no_slab_sum = on("MCHIP_NO_SLAB_SUM");
	ctx->knob.no_fused_finalize = on("MCHIP_NO_FUSED_FINALIZE");
	ctx->knob.no_col_split = on("MCHIP_NO_COL_SPLIT");
	ctx->knob.part_no_tile = on("MCHIP_PART_NO_TILE");
	ctx->knob.sim_no_tile = on("MCHIP_SIM_NO_TILE");
	const int both = num("MCHIP_BLOCKS_PER_CU", 64);
	ctx->knob.per_cu_col = num("MCHIP_BLOCKS_PER_CU_COL", both);
	ctx->knob.per_cu_ind = num("MCHIP_BLOCKS_PER_CU_IND", both);
	const char *f = getenv("MCHIP_SLAB_FRAC");
	ctx->knob.slab_frac = (f && atof(f) > 0) ? atof(f) : 0.3;
	ctx->knob.no_roundup = on("MCHIP_NO_CHUNK_ROUNDUP");
	ctx->knob.geometry_given = on("MCHIP_BLOCKS_PER_CU") || on("MCHIP_BLOCKS_PER_CU_COL") || on("MCHIP_SLAB_FRAC");
}

static int fail(mchip_context *ctx, int code, const char *fmt, const char *detail)
{
	if (ctx) snprintf(ctx->err, sizeof ctx->err, fmt, detail ? detail : "");
	return code;
}

#define HIPCHK(call)                                                                                  \
	do {                                                                                          \
		hipError_t e_ = MCHIP_WAIT(call);                                                     \
		if (e_ != hipSuccess) {                                                               \
			snprintf(ctx->err, sizeof ctx->err, "%s failed: %s (%s:%d)", #call,           \
				 hipGetErrorString(e_), __FILE__, __LINE__);                          \
			return MCHIP_ERR_HIP;                                                         \
		}                                                                                     \
	} while (0)

template <typename Tp> static void dfree(Tp *&p)
{
	if (p) (void)MCHIP_WAIT(hipFree(p));
	p = nullptr;
}

/* temporary device allocation released on every exit path of the function that owns it */
template <typename Tp> struct scoped_dev {
	Tp *p = nullptr;
	scoped_dev() = default;
	scoped_dev(const scoped_dev &) = delete;
	scoped_dev &operator=(const scoped_dev &) = delete;
	~scoped_dev() { if (p) (void)MCHIP_WAIT(hipFree(p)); }
	hipError_t alloc(size_t count) { return hipMalloc((void **)&p, count * sizeof(Tp)); }
	operator Tp *() const { return p; }
};

/* ------------------------------------------------------------------ K-independent kernels */

/* raw [I][L][pl] bytes -> gtA [ceil(I/8)][L][8][pl] and gtS [ceil(L/8)][I][8][pl]; pads with 0xFF;
 * validates allele indices when ua != nullptr (limit = ua[l]) or against `limit` otherwise.
 * A workgroup moves one tile of RT_I individuals x RT_L loci through LDS: the rows of the tile are read once, contiguously
 * (RT_L * pl bytes per row), and both layouts are written in runs of whole 8 * pl-byte groups that are contiguous in memory
 * (gtA: consecutive loci of one block of 8 individuals; gtS: consecutive individuals of one block of 8 loci).  The first
 * version read raw[] once per OUTPUT byte in output order -- 8 rows per group, a few bytes of every 64-byte line at a time:
 * 28.5 GB of HBM traffic for a 2 GB genotype (profiles/r01_v5_c3_traffic.json); it is paid per initialisation (the
 * partition takes the same route) and per bootstrap replicate. */
constexpr int RT_I = 64;	/* individuals per tile; loci per tile: 64 up to ploidy 8, fewer above (32 KiB of LDS) */

__global__ __launch_bounds__(256) void k_relayout(const uint8_t *__restrict__ raw, int I, int L, int pl, const int32_t *__restrict__ ua,
			   int limit, uint8_t *gtA, uint8_t *gtS, int n_ltiles, int RT_L, int *bad)
{
	extern __shared__ __attribute__((aligned(16))) uint8_t tile[];	/* [RT_I][RT_L * pl] */
	const int rowb = RT_L * pl;		/* bytes per tile row: a multiple of 8 */
	const int l0 = (blockIdx.x % n_ltiles) * RT_L, i0 = (blockIdx.x / n_ltiles) * RT_I;
	int flags = 0;
	const int roww = rowb / 4;		/* dwords per tile row */
	/* rows are read four bytes at a time when every row of the tile starts on a 4-byte boundary of raw[] and lies inside it */
	const bool words = ((((size_t)L * pl) | ((size_t)l0 * pl) | (size_t)raw) & 3) == 0 && l0 + RT_L <= L;
	for (int x = threadIdx.x; x < RT_I * roww; x += 256) {
		const int r = x / roww, w = x % roww;
		const int i = i0 + r;
		uint32_t v4 = 0xFFFFFFFFu;
		if (i < I) {
			const uint8_t *src = raw + ((size_t)i * L + l0) * pl + (size_t)w * 4;
			if (words) {
				v4 = *reinterpret_cast<const uint32_t *>(src);
			} else {
				v4 = 0;
				for (int y = 0; y < 4; y++) {
					const int l = l0 + (w * 4 + y) / pl;
					const uint32_t v = l < L ? src[y] : 0xFFu;
					v4 |= v << (8 * y);
				}
			}
			for (int y = 0; y < 4; y++) {
				const int l = l0 + (w * 4 + y) / pl;
				const uint32_t v = (v4 >> (8 * y)) & 0xFFu;
				if (l >= L) continue;
				const int lim = ua ? ua[l] : limit;
				if (v != 0xFFu && (int)v >= lim) flags |= 1;
				if (v == 0xFFu) flags |= 2;	/* bit 1: the data set has missing copies */
			}
		}
		reinterpret_cast<uint32_t *>(tile)[x] = v4;
	}
	if (flags) atomicOr(bad, flags);
	__syncthreads();
	const int grp = 8 * pl, grpw = 2 * pl;	/* bytes / dwords per group of 8 entries */
	/* gtA: block ib of 8 individuals, loci l0 .. l0+RT_L-1: RT_L groups, contiguous; one dword (4 of the group's bytes) per thread */
	for (int x = threadIdx.x; x < (RT_I / 8) * RT_L * grpw; x += 256) {
		const int w = x % grpw, ll = (x / grpw) % RT_L, ibl = x / (grpw * RT_L);
		const int l = l0 + ll;
		const size_t ib = (size_t)(i0 / 8) + ibl;
		if (l >= L || ib * 8 >= (size_t)I) continue;
		uint32_t v4 = 0;
		for (int y = 0; y < 4; y++) {
			const int e = w * 4 + y, j = e / pl, a = e % pl;
			v4 |= (uint32_t)tile[(ibl * 8 + j) * rowb + ll * pl + a] << (8 * y);
		}
		reinterpret_cast<uint32_t *>(gtA + (ib * L + l) * (size_t)grp)[w] = v4;
	}
	/* gtS: block lb of 8 loci, individuals i0 .. i0+RT_I-1: RT_I groups, contiguous; the group's bytes are contiguous in the tile row */
	for (int x = threadIdx.x; x < (RT_L / 8) * RT_I * grpw; x += 256) {
		const int w = x % grpw, r = (x / grpw) % RT_I, lbl = x / (grpw * RT_I);
		const int i = i0 + r;
		const size_t lb = (size_t)(l0 / 8) + lbl;
		if (i >= I || lb * 8 >= (size_t)L) continue;
		reinterpret_cast<uint32_t *>(gtS + (lb * I + i) * (size_t)grp)[w] = reinterpret_cast<const uint32_t *>(tile + r * rowb + lbl * grp)[w];
	}
}

/* launch: one workgroup per tile; LDS = RT_I * RT_L * pl <= 32 KiB */
static int launch_relayout(hipStream_t stream, const uint8_t *raw, int I, int L, int pl, const int32_t *ua, int limit,
			   uint8_t *gtA, uint8_t *gtS, int *bad)
{
	int RT_L = 64;
	while (RT_L > 8 && RT_I * RT_L * pl > 32 * 1024) RT_L -= 8;
	const int n_ltiles = (L + RT_L - 1) / RT_L, n_itiles = (I + RT_I - 1) / RT_I;
	const size_t blocks = (size_t)n_ltiles * n_itiles;
	if (blocks >= ((size_t)1 << 31) || (size_t)RT_I * RT_L * pl > 64 * 1024) return -1;
	hipLaunchKernelGGL(k_relayout, dim3((unsigned)blocks), dim3(256), (size_t)RT_I * RT_L * pl, stream, raw, I, L, pl, ua, limit, gtA, gtS,
			   n_ltiles, RT_L, bad);
	return 0;
}

/* ------------------------------------------------------------------ libc-compatible rand() on the device
 * glibc's TYPE_3 generator is the additive lagged Fibonacci recurrence x_j = x_{j-31} + x_{j-3} (mod 2^32) with
 * rand() = x_j >> 1.  It is linear, so x^n mod (x^31 - x^28 - 1) over Z/2^32 carries a 31-word window n draws
 * ahead.  Thread c draws the RNG_CHUNK consecutive values that start RNG_CHUNK*c draws into the stream: its jump
 * polynomial is the product of two tabulated ones (hi = c / 256, lo = c % 256), so the stream is the serial one
 * whatever the launch geometry (random_allele_partition, rnd_init.c:456-467: one rand() % K per allele copy). */
constexpr int RNG_LAG = 31;
constexpr int RNG_CHUNK = 4 * RNG_LAG * 32;	/* draws (= bytes written) per thread: 32 rounds of 31 packed words */

struct rng_window { uint32_t s[2 * RNG_LAG - 1]; };	/* x_{j-31} .. x_{j+29}: window, then its next 30 values */

/* window (31 words behind the first draw) of a thread whose first draw lies n draws into the stream, x^n = hi(x) * lo(x)
 * mod (x^31 - x^28 - 1): hi is wave-uniform (scalar loads), lo is stored [31][lo_stride] so that lanes read consecutive words */
__device__ __forceinline__ void rng_window_at(const rng_window &base, const uint32_t *__restrict__ hi,
		const uint32_t *__restrict__ lo_table, size_t lo_stride, size_t lo_index, uint32_t (&w)[RNG_LAG])
{
	uint32_t t[2 * RNG_LAG - 1];
#pragma unroll
	for (int d = 0; d < 2 * RNG_LAG - 1; d++) t[d] = 0;
	{
		uint32_t lo[RNG_LAG];
#pragma unroll
		for (int j = 0; j < RNG_LAG; j++) lo[j] = lo_table[j * lo_stride + lo_index];
#pragma unroll
		for (int i = 0; i < RNG_LAG; i++) {
			const uint32_t h = hi[i];
#pragma unroll
			for (int j = 0; j < RNG_LAG; j++) t[i + j] += h * lo[j];
		}
#pragma unroll
		for (int d = 2 * RNG_LAG - 2; d >= RNG_LAG; d--) {	/* x^d = x^(d-3) + x^(d-31) */
			t[d - 3] += t[d];
			t[d - RNG_LAG] += t[d];
		}
	}
	/* w[e] = sum_j poly[j] * s[e + j] */
#pragma unroll
	for (int e = 0; e < RNG_LAG; e++) {
		uint32_t v = 0;
#pragma unroll
		for (int j = 0; j < RNG_LAG; j++) v += t[j] * base.s[e + j];
		w[e] = v;
	}
}

/* the thread whose chunk starts 256*blockIdx.x + threadIdx.x chunks of RNG_CHUNK draws into the stream: hi = x^(256*CHUNK*block),
 * lo = x^(CHUNK*thread) */
__device__ __forceinline__ void rng_thread_window(const rng_window &base, const uint32_t *__restrict__ jump_hi,
		const uint32_t *__restrict__ jump_lo, uint32_t (&w)[RNG_LAG])
{
	rng_window_at(base, jump_hi + (size_t)blockIdx.x * RNG_LAG, jump_lo, 256, threadIdx.x, w);
}

__global__ __launch_bounds__(256) void k_draw_partition(rng_window base, const uint32_t *__restrict__ jump_hi,
		const uint32_t *__restrict__ jump_lo, size_t n_chunks, uint32_t K, uint32_t magic, uint32_t shift, uint32_t *out)
{
	const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (c >= n_chunks) return;
	uint32_t w[RNG_LAG];
	rng_thread_window(base, jump_hi, jump_lo, w);
	uint32_t *dst = out + c * (RNG_CHUNK / 4);
	for (int round = 0; round < RNG_CHUNK / (4 * RNG_LAG); round++) {
		uint32_t word[RNG_LAG];
#pragma unroll
		for (int d = 0; d < 4 * RNG_LAG; d++) {
			const int e = d % RNG_LAG;
			const uint32_t x = w[e] + w[(e + RNG_LAG - 3) % RNG_LAG];	/* x_{j-31} + x_{j-3} */
			w[e] = x;
			const uint32_t v = x >> 1;
			/* v % K by multiplication: floor(v / K) = (v * magic) >> (31 + ceil(log2 K)), exact for v < 2^31 */
			const uint32_t r = v - (__umulhi(v, magic) >> shift) * K;
			if (d % 4 == 0) word[d / 4] = r;
			else word[d / 4] |= r << (8 * (d % 4));
		}
#pragma unroll
		for (int x = 0; x < RNG_LAG; x++) dst[round * RNG_LAG + x] = word[x];
	}
}

/* parametric_bootstrap_admixture (bootstrap.c:84-124) on the device.  Allele copy j (i, l, n order) takes draws 2j
 * (source cluster: inverse-CDF walk over the individual's eta) and 2j+1 (allele: walk over p[k][l][.]) of the rand()
 * stream, r = rand() / RAND_MAX in double.  The reference's walk `while (k < K && r > sum) sum += eta[k++]; if (k) k--;`
 * stops at the largest k whose left-to-right partial sum c_k = eta[0] + ... + eta[k-1] lies below r (partial sums of
 * non-negative terms never decrease), and r > c_k is a statement about the integer the generator returned: v / RAND_MAX,
 * rounded, never decreases in v, so r > c_k  <=>  v >= V(c_k) with V(c) the smallest v whose quotient exceeds c.
 * k_walk_tables finds every V once per call -- with the same double additions and the same division the reference executes
 * per copy -- and the per-copy work is integer compares against those thresholds: no running sums, no divisions, and none
 * of a copy's loads waits for another.  Every copy is simulated (in the default build the "missing stays missing" count is
 * overwritten before it is used, bootstrap.c:87-96). */
constexpr int SIM_QW = 8, SIM_PW = 4;	/* widths of the padded threshold rows of the FAST form (K <= 8, at most 4 alleles per locus) */

/* smallest v in [0, 2^31] with (double)v / RAND_MAX > c (2^31: no value of rand() does) */
__device__ uint32_t walk_threshold(double c)
{
	const double D = 2147483647.0;
	if (!(c < 2.0)) return 0x80000000u;	/* v / D <= 1 < 2; also +inf (padding) and NaN */
	double g = c * D - 2.0;
	uint32_t v = g > 0.0 ? (uint32_t)g : 0u;	/* c < 2: g < 2^32 */
	if (v > 0x80000000u) v = 0x80000000u;
	while (v > 0u && (double)(v - 1u) / D > c) v--;
	while (v < 0x80000000u && !((double)v / D > c)) v++;
	return v;
}

/* thq[i][j] = V(q[i][0] + ... + q[i][j-1]) (j < K; rows of SIM_QW padded with 2^31 when fast); thp[k][c0 + m] =
 * V(p[k][c0] + ... + p[k][c0 + m - 1]), or rows of SIM_PW per (k, l), padded, when fast */
__global__ void k_walk_tables(int n_qrows, int K, int L, int T, const int32_t *__restrict__ toff, const double *__restrict__ q,
			      const double *__restrict__ p, int fast, uint32_t *thq, uint32_t *thp)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx < (size_t)n_qrows) {
		const double *src = q + idx * K;
		uint32_t *dst = thq + idx * (size_t)(fast ? SIM_QW : K);
		double sum = 0.0;
		for (int j = 0; j < K; j++) { dst[j] = walk_threshold(sum); sum += src[j]; }
		if (fast) for (int j = K; j < SIM_QW; j++) dst[j] = 0x80000000u;
	}
	if (idx < (size_t)K * L) {
		const int l = (int)(idx % L), k = (int)(idx / L);
		const int c0 = toff[l], M = toff[l + 1] - c0;
		const double *src = p + (size_t)k * T + c0;
		uint32_t *dst = fast ? thp + idx * SIM_PW : thp + (size_t)k * T + c0;
		double sum = 0.0;
		for (int m = 0; m < M; m++) { dst[m] = walk_threshold(sum); sum += src[m]; }
		if (fast) for (int m = M; m < SIM_PW; m++) dst[m] = 0x80000000u;
	}
}

/* largest j in [0, n) with v >= tab[j], 0 if there is none (tab never decreases) */
__device__ __forceinline__ int walk_search(const uint32_t *__restrict__ tab, int n, uint32_t v)
{
	int lo = 0;
	while (n > 1) {
		const int half = n >> 1;
		if (v >= tab[lo + half]) lo += half;
		n -= half;
	}
	return lo;
}

/* Thread c owns RNG_CHUNK consecutive draws = RNG_CHUNK/2 copies; its generator window lives in LDS ([31][256]: word e of
 * all lanes is one conflict-free row).  Four copies (one output word) are in flight at a time: eight draws, then the four
 * cluster searches, then the four allele searches, so that the loads of the four overlap. */
template <bool FAST>
__global__ __launch_bounds__(256) void k_simulate_admixture(rng_window base, const uint32_t *__restrict__ jump_hi,
		const uint32_t *__restrict__ jump_lo, size_t n_chunks, int I, int L, int pl, int K, int T,
		const int32_t *__restrict__ toff, const uint32_t *__restrict__ thq, int per_individual, const uint32_t *__restrict__ thp,
		uint32_t *out)
{
	__shared__ uint32_t ring[RNG_LAG * 256];
	const size_t c = (size_t)blockIdx.x * 256 + threadIdx.x;
	if (c >= n_chunks) return;
	{
		uint32_t w[RNG_LAG];
		rng_thread_window(base, jump_hi, jump_lo, w);
#pragma unroll
		for (int e = 0; e < RNG_LAG; e++) ring[e * 256 + threadIdx.x] = w[e];
	}
	int e = 0;
	auto next = [&]() -> uint32_t {
		const int e3 = e >= 3 ? e - 3 : e + RNG_LAG - 3;
		const uint32_t x = ring[e * 256 + threadIdx.x] + ring[e3 * 256 + threadIdx.x];	/* x_{j-31} + x_{j-3} */
		ring[e * 256 + threadIdx.x] = x;
		e = e + 1 == RNG_LAG ? 0 : e + 1;
		return x >> 1;		/* rand() */
	};
	constexpr int COPIES = RNG_CHUNK / 2;
	const size_t j0 = c * COPIES;
	const size_t per_i = (size_t)L * pl;
	int i = (int)(j0 / per_i);	/* <= I; copies past the data set's end are clamped to the last individual */
	int l = (int)((j0 % per_i) / pl), a = (int)(j0 % pl);
	uint32_t *dst = out + c * (COPIES / 4);
	const int qrow = FAST ? SIM_QW : K;
	for (int x = 0; x < COPIES / 4; x++) {
		uint32_t v1[4], v2[4];
		int ci[4], cl[4], k[4];
#pragma unroll
		for (int y = 0; y < 4; y++) {
			v1[y] = next();
			v2[y] = next();
			ci[y] = i < I ? i : I - 1;
			cl[y] = l;
			if (++a == pl) {
				a = 0;
				if (++l == L) { l = 0; i++; }
			}
		}
#pragma unroll
		for (int y = 0; y < 4; y++) {
			const uint32_t *tab = thq + (per_individual ? (size_t)ci[y] * qrow : 0);
			if (FAST) {
				const uint4 t0 = *reinterpret_cast<const uint4 *>(tab), t1 = *reinterpret_cast<const uint4 *>(tab + 4);
				k[y] = (v1[y] >= t0.y) + (v1[y] >= t0.z) + (v1[y] >= t0.w) + (v1[y] >= t1.x) + (v1[y] >= t1.y) + (v1[y] >= t1.z) +
				       (v1[y] >= t1.w);		/* t0.x = V(0): k = 0 either way */
			} else {
				k[y] = walk_search(tab, K, v1[y]);
			}
		}
		uint32_t word = 0;
#pragma unroll
		for (int y = 0; y < 4; y++) {
			int m;
			if (FAST) {
				const uint4 t = *reinterpret_cast<const uint4 *>(thp + ((size_t)k[y] * L + cl[y]) * SIM_PW);
				m = (v2[y] >= t.y) + (v2[y] >= t.z) + (v2[y] >= t.w);
			} else {
				const int c0 = toff[cl[y]], M = toff[cl[y] + 1] - c0;
				m = walk_search(thp + (size_t)k[y] * T + c0, M, v2[y]);
			}
			word |= (uint32_t)m << (8 * y);
		}
		dst[x] = word;
	}
}

/* out[j * sj + n * sn] = coefficient j of x^(A*n) mod (x^31 - x^28 - 1), n < count, from pow2[b] = x^(A * 2^b): the jump
 * polynomials of a regular lattice of stream positions, computed once per data-set shape */
__global__ void k_jump_powers(const uint32_t *__restrict__ pow2, unsigned count, size_t sj, size_t sn, uint32_t *out)
{
	const unsigned n = blockIdx.x * blockDim.x + threadIdx.x;
	if (n >= count) return;
	uint32_t acc[RNG_LAG];
	for (int j = 0; j < RNG_LAG; j++) acc[j] = j == 0;
	for (int b = 0; b < 32 && (n >> b); b++) {
		if (!((n >> b) & 1u)) continue;
		uint32_t t[2 * RNG_LAG - 1];
		for (int d = 0; d < 2 * RNG_LAG - 1; d++) t[d] = 0;
		for (int i = 0; i < RNG_LAG; i++)
			for (int j = 0; j < RNG_LAG; j++) t[i + j] += acc[i] * pow2[b * RNG_LAG + j];
		for (int d = 2 * RNG_LAG - 2; d >= RNG_LAG; d--) {
			t[d - 3] += t[d];
			t[d - RNG_LAG] += t[d];
		}
		for (int j = 0; j < RNG_LAG; j++) acc[j] = t[j];
	}
	for (int j = 0; j < RNG_LAG; j++) out[j * sj + n * sn] = acc[j];
}

/* The same data set generated tile by tile (FAST form: K <= 8, at most 4 alleles per locus, ploidy <= 8): workgroup = 256
 * individuals x the `tile_loci` loci from blockIdx.x * tile_loci on.  Thread = individual i: its draws for the tile start
 * 2 * (i * L * PL + l0 * PL) into the stream, x^that = jump_i[i] * jump_r[blockIdx.x] (k_jump_powers).  The allele
 * thresholds of the tile's loci are staged in LDS ([locus][k][3]: the 64 lanes are at the same locus and differ in k only), the
 * individual's own cluster thresholds sit in registers, and the copies go straight into the two device layouts (gtS: the
 * thread's 8 loci x PL bytes are one contiguous run, runs of consecutive individuals adjacent; gtA: PL bytes per locus, 8
 * consecutive lanes adjacent) -- no raw [I][L][PL] intermediate and no gather from global memory.  tile_loci % 8 == 0. */
template <int PL>
__global__ __launch_bounds__(256) void k_simulate_tile(rng_window base, const uint32_t *__restrict__ jump_i,
		const uint32_t *__restrict__ jump_r, int I, int L, int K, int tile_loci, const uint32_t *__restrict__ thq,
		int per_individual, const uint32_t *__restrict__ thp, uint8_t *__restrict__ gtA, uint8_t *__restrict__ gtS)
{
	extern __shared__ uint32_t sim_lds[];
	uint32_t *ring = sim_lds, *tile = sim_lds + RNG_LAG * 256;
	const int l0 = blockIdx.x * tile_loci;
	const int nl = L - l0 < tile_loci ? L - l0 : tile_loci;
	const int i = blockIdx.y * 256 + threadIdx.x;
	for (int idx = threadIdx.x; idx < K * nl; idx += 256) {
		const int k = idx / nl, ll = idx - k * nl;
		const uint4 t = *reinterpret_cast<const uint4 *>(thp + ((size_t)k * L + l0 + ll) * SIM_PW);
		uint32_t *dst = tile + ((size_t)ll * K + k) * 3;
		dst[0] = t.y; dst[1] = t.z; dst[2] = t.w;	/* t.x = V(0): allele 0 either way */
	}
	__syncthreads();
	if (i >= ((I + 7) & ~7)) return;
	uint8_t *colA = gtA + ((((size_t)(i >> 3) * L + l0) * 8) + (i & 7)) * PL;	/* + ll * 8 * PL per locus */
	if (i >= I) {		/* rows that pad the last block of 8 individuals */
		for (int ll = 0; ll < nl; ll++)
#pragma unroll
			for (int a = 0; a < PL; a++) colA[(size_t)ll * 8 * PL + a] = MCHIP_MISSING;
		return;
	}
	{
		uint32_t w[RNG_LAG];
		rng_window_at(base, jump_r + (size_t)blockIdx.x * RNG_LAG, jump_i, (size_t)I, (size_t)i, w);
#pragma unroll
		for (int e = 0; e < RNG_LAG; e++) ring[e * 256 + threadIdx.x] = w[e];
	}
	int e = 0;
	auto next = [&]() -> uint32_t {
		const int e3 = e >= 3 ? e - 3 : e + RNG_LAG - 3;
		const uint32_t x = ring[e * 256 + threadIdx.x] + ring[e3 * 256 + threadIdx.x];	/* x_{j-31} + x_{j-3} */
		ring[e * 256 + threadIdx.x] = x;
		e = e + 1 == RNG_LAG ? 0 : e + 1;
		return x >> 1;		/* rand() */
	};
	uint32_t tq[SIM_QW - 1];
	{
		const uint32_t *row = thq + (per_individual ? (size_t)i * SIM_QW : 0);
		const uint4 t0 = *reinterpret_cast<const uint4 *>(row), t1 = *reinterpret_cast<const uint4 *>(row + 4);
		tq[0] = t0.y; tq[1] = t0.z; tq[2] = t0.w; tq[3] = t1.x; tq[4] = t1.y; tq[5] = t1.z; tq[6] = t1.w;
	}
	for (int g = 0; g * 8 < nl; g++) {
		unsigned long long run[PL];	/* the 8 * PL bytes of this individual's gtS group */
#pragma unroll
		for (int x = 0; x < PL; x++) run[x] = ~0ull;
#pragma unroll
		for (int t = 0; t < 8; t++) {
			const int ll = g * 8 + t;
			if (ll < nl) {		/* uniform */
				uint32_t v1[PL], v2[PL];
#pragma unroll
				for (int a = 0; a < PL; a++) { v1[a] = next(); v2[a] = next(); }
				unsigned long long bytes = 0;
#pragma unroll
				for (int a = 0; a < PL; a++) {
					int k = 0;
#pragma unroll
					for (int x = 0; x < SIM_QW - 1; x++) k += v1[a] >= tq[x];
					const uint32_t *th = tile + ((size_t)ll * K + k) * 3;
					const unsigned m = (v2[a] >= th[0]) + (v2[a] >= th[1]) + (v2[a] >= th[2]);
					bytes |= (unsigned long long)m << (8 * a);
				}
				uint8_t *dst = colA + (size_t)ll * 8 * PL;
				if (PL == 4) *reinterpret_cast<uint32_t *>(dst) = (uint32_t)bytes;
				else if (PL == 2) *reinterpret_cast<uint16_t *>(dst) = (uint16_t)bytes;
				else if (PL == 8) *reinterpret_cast<unsigned long long *>(dst) = bytes;
				else {
#pragma unroll
					for (int a = 0; a < PL; a++) dst[a] = (uint8_t)(bytes >> (8 * a));
				}
#pragma unroll
				for (int a = 0; a < PL; a++) {
					constexpr unsigned long long ff = 0xFFull;
					const int b = t * PL + a;
					run[b >> 3] = (run[b >> 3] & ~(ff << (8 * (b & 7)))) | (((bytes >> (8 * a)) & ff) << (8 * (b & 7)));
				}
			}
		}
		unsigned long long *dstS = reinterpret_cast<unsigned long long *>(gtS + (((size_t)(l0 / 8 + g) * I + i) * 8) * PL);
#pragma unroll
		for (int x = 0; x < PL; x++) dstS[x] = run[x];
	}
}

template <int PL>
static void launch_simulate_tile(mchip_context *ctx, const rng_window &base, int K, int tile, const uint32_t *thq, int per_individual,
				 const uint32_t *thp)
{
	const unsigned R = (unsigned)((ctx->L + tile - 1) / tile), by = (unsigned)((((ctx->I + 7) & ~7) + 255) / 256);
	const size_t lds = ((size_t)RNG_LAG * 256 + (size_t)tile * K * 3) * sizeof(uint32_t);
	hipLaunchKernelGGL(k_simulate_tile<PL>, dim3(R, by), dim3(256), lds, ctx->stream, base, ctx->lat[0].d_i, ctx->lat[0].d_r, ctx->I, ctx->L, K,
			   tile, thq, per_individual, thp, ctx->d_gtA, ctx->d_gtS);
}

/* random_allele_partition + the counts of the first M step (rnd_init.c:349-357,456-482) in one pass, nothing stored per copy.
 * Workgroup = 256 individuals x one locus chunk of the S-side pass (blockIdx.x; its loci taken `tile` at a time); thread =
 * individual i: copy (i, l, b) takes draw i*L*PL + l*PL + b of the stream, so the thread's draws for the chunk are consecutive
 * from x^(L*PL*i) * x^(lchunk*PL*blockIdx.x) on (k_jump_powers).  d[i][k][l][m] = 1 for every (allele m, cluster k) pair
 * some non-missing copy of the individual at the locus was given: the thread drops the repeats among its PL copies, then
 *   S side: one count per pair into its own packed 16-bit counters in LDS (at most lchunk*PL <= 65535), written as doubles
 *           to Spart[chunk][i][.] at the end -- what k_partition_individuals writes;
 *   N side: one LDS atomic per pair into the tile's packed 16-bit counters [column][KP] (at most 256 per workgroup), stored
 *           per tile into this workgroup's slab; k_partition_finish adds the slabs.  Integer counts: the order of the
 *           atomics does not matter, the sums are exact.
 * The genotype comes from gtS (the thread's 8 loci x PL bytes are one contiguous run).  KP = K rounded up to even. */
template <int PL>
__global__ __launch_bounds__(256) void k_partition_tile(rng_window base, const uint32_t *__restrict__ jump_i,
		const uint32_t *__restrict__ jump_r, int I, int L, int K, uint32_t magic, uint32_t shift, int lchunk, int tile,
		const int32_t *__restrict__ toff, const uint8_t *__restrict__ gtS, uint32_t *__restrict__ slabs, size_t slab_words,
		double *__restrict__ Spart)
{
	extern __shared__ uint32_t part_lds[];
	const int KP = (K + 1) & ~1;
	uint32_t *ring = part_lds, *scnt = part_lds + RNG_LAG * 256, *ncnt = scnt + (KP / 2) * 256;
	const int i = blockIdx.y * 256 + threadIdx.x;
	const bool live = i < I;
	const int l_begin = blockIdx.x * lchunk, l_end = L - l_begin < lchunk ? L : l_begin + lchunk;
	if (live) {
		uint32_t w[RNG_LAG];
		rng_window_at(base, jump_r + (size_t)blockIdx.x * RNG_LAG, jump_i, (size_t)I, (size_t)i, w);
#pragma unroll
		for (int e = 0; e < RNG_LAG; e++) ring[e * 256 + threadIdx.x] = w[e];
	}
	for (int x = 0; x < KP / 2; x++) scnt[x * 256 + threadIdx.x] = 0;
	int e = 0;
	auto next = [&]() -> uint32_t {
		const int e3 = e >= 3 ? e - 3 : e + RNG_LAG - 3;
		const uint32_t x = ring[e * 256 + threadIdx.x] + ring[e3 * 256 + threadIdx.x];	/* x_{j-31} + x_{j-3} */
		ring[e * 256 + threadIdx.x] = x;
		e = e + 1 == RNG_LAG ? 0 : e + 1;
		const uint32_t v = x >> 1;		/* rand() */
		return K == 1 ? 0u : v - (__umulhi(v, magic) >> shift) * (uint32_t)K;	/* rand() % K (k_draw_partition) */
	};
	uint32_t *slab = slabs + (size_t)blockIdx.y * slab_words;
	for (int t0 = l_begin; t0 < l_end; t0 += tile) {
		const int t1 = l_end - t0 < tile ? l_end : t0 + tile;
		const int col0 = toff[t0], nwords = (toff[t1] - col0) * (KP / 2);
		for (int x = threadIdx.x; x < nwords; x += 256) ncnt[x] = 0;
		__syncthreads();
		if (live)
			for (int g = t0 >> 3; g * 8 < t1; g++) {
				unsigned long long run[PL];
				{
					const unsigned long long *src = reinterpret_cast<const unsigned long long *>(gtS + (((size_t)g * I + i) * 8) * PL);
#pragma unroll
					for (int x = 0; x < PL; x++) run[x] = src[x];
				}
#pragma unroll
				for (int t = 0; t < 8; t++) {
					const int l = g * 8 + t;
					if (l < t1) {		/* uniform */
						const int cl = toff[l] - col0;
						uint32_t key[PL];
#pragma unroll
						for (int b = 0; b < PL; b++) {
							const int at = t * PL + b;
							const uint32_t mb = (uint32_t)(run[at >> 3] >> (8 * (at & 7))) & 0xFFu;
							const uint32_t kb = next();
							key[b] = mb | (kb << 8);
							bool count = mb != MCHIP_MISSING;
#pragma unroll
							for (int b2 = 0; b2 < b; b2++) count = count && key[b2] != key[b];
							if (count) {
								scnt[(kb >> 1) * 256 + threadIdx.x] += 1u << (16 * (kb & 1u));
								const uint32_t en = (uint32_t)(cl + (int)mb) * (uint32_t)KP + kb;
								atomicAdd(&ncnt[en >> 1], 1u << (16 * (en & 1u)));
							}
						}
					}
				}
			}
		__syncthreads();
		uint32_t *dst = slab + (size_t)col0 * (KP / 2);
		for (int x = threadIdx.x; x < nwords; x += 256) dst[x] = ncnt[x];
		__syncthreads();
	}
	if (live) {
		double *out = Spart + ((size_t)blockIdx.x * I + i) * K;
		for (int k = 0; k < K; k++) out[k] = (double)((scnt[(k >> 1) * 256 + threadIdx.x] >> (16 * (k & 1))) & 0xFFFFu);
	}
}

/* Apart[0][c][k] = sum over the workgroup slabs of k_partition_tile; thread = (c, k) */
__global__ void k_partition_finish(const uint32_t *__restrict__ slabs, size_t slab_words, int n_slabs, int T, int K, double *Apart)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (size_t)T * K) return;
	const int k = (int)(idx % K);
	const size_t c = idx / K;
	const int KP = (K + 1) & ~1;
	const size_t en = c * KP + k;
	unsigned sum = 0;
	for (int b = 0; b < n_slabs; b++) sum += (slabs[(size_t)b * slab_words + (en >> 1)] >> (16 * (en & 1))) & 0xFFFFu;
	Apart[idx] = (double)sum;
}

template <int PL>
static void launch_partition_tile(mchip_context *ctx, const rng_window &base, uint32_t magic, uint32_t shift, int tile, size_t lds,
				  const uint8_t *gtS, size_t slab_words)
{
	const unsigned by = (unsigned)((ctx->I + 255) / 256);
	hipLaunchKernelGGL(k_partition_tile<PL>, dim3((unsigned)ctx->n_lchunks, by), dim3(256), lds, ctx->stream, base, ctx->lat[1].d_i,
			   ctx->lat[1].d_r, ctx->I, ctx->L, ctx->K, magic, shift, ctx->lchunk, tile, ctx->d_toff, gtS, ctx->d_part_slabs, slab_words,
			   ctx->d_Spart);
}

/* random_allele_center's assignment (rnd_init.c:552-580) given the centers the host drew: copy (i, a) of locus l goes to the
 * first cluster k whose center allele it carries; a copy that matches no center (a missing copy never does) takes the next
 * value of rand() % K, in i, a order within the locus.  lane = locus: the locus's non-matching copies are numbered by the
 * lane's own running count, their draws start draw_offset[l] into the byte stream `draws` (rand() % K of the candidate's
 * whole stream span, from k_draw_partition).  Writes the partition in stream-independent raw order [I][L][pl]. */
__global__ __launch_bounds__(256) void k_assign_by_centers(const uint8_t *__restrict__ gtA, int I, int L, int pl, int K,
		const uint8_t *__restrict__ centers, const unsigned long long *__restrict__ draw_offset,
		const uint8_t *__restrict__ draws, unsigned long long n_draws, uint8_t *raw, int *bad)
{
	const int l = blockIdx.x * 256 + threadIdx.x;
	if (l >= L) return;
	uint8_t cen[MCHIP_MAX_K];
	for (int k = 0; k < K; k++) cen[k] = centers[(size_t)l * K + k];
	unsigned long long next = draw_offset[l];
	for (int ib = 0; ib < (I + 7) / 8; ib++) {
		const uint8_t *src = gtA + ((size_t)ib * L + l) * 8 * (size_t)pl;
		for (int j = 0; j < 8; j++) {
			const int i = ib * 8 + j;
			if (i >= I) break;
			for (int b = 0; b < pl; b++) {
				const uint8_t g = src[j * pl + b];
				int kk = -1;
				if (g != MCHIP_MISSING)
					for (int k = 0; k < K; k++) {
						if (cen[k] == 0xFF) break;	/* center[k] == -1 ends the list (rnd_init.c:562-563) */
						if (cen[k] == g) { kk = k; break; }
					}
				if (kk < 0) {
					if (K == 1) kk = 0;			/* every copy to cluster 0, nothing drawn */
					else if (next < n_draws) kk = draws[next++];
					else { kk = 0; atomicOr(bad, 1); }	/* the host counted other genotypes than the device holds */
				}
				raw[((size_t)i * L + l) * pl + b] = (uint8_t)kk;
			}
		}
	}
}

/* gtA -> raw [I][L][pl] (mchip_get_genotypes) */
__global__ void k_unlayout(const uint8_t *__restrict__ gtA, int I, int L, int pl, uint8_t *raw)
{
	const size_t n = (size_t)I * L * pl, stride = (size_t)gridDim.x * blockDim.x;
	for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < n; idx += stride) {
		size_t r = idx;
		const int a = (int)(r % pl); r /= pl;
		const int l = (int)(r % L);
		const size_t i = r / L;
		raw[idx] = gtA[(((i >> 3) * L + l) * 8 + (i & 7)) * (size_t)pl + a];
	}
}

/* gtA -> packed counts gtC[g][c]: thread = (group g of G individuals, column c) */
template <int BITS>
__global__ void k_build_counts(const uint8_t *__restrict__ gtA, int I, int L, int pl, int T,
			       const int32_t *__restrict__ col_locus, const uint8_t *__restrict__ col_allele, uint4 *gtC)
{
	constexpr int PERWORD = 32 / BITS, G = 4 * PERWORD;
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	const size_t ngroups = (size_t)(I + G - 1) / G;
	if (idx >= ngroups * T) return;
	const int c = (int)(idx % T);
	const size_t g = idx / T;
	const int l = col_locus[c];
	const unsigned m = col_allele[c];
	unsigned out[4] = {0, 0, 0, 0};
	for (int x = 0; x < G; x++) {
		const size_t i = g * G + x;
		if (i >= (size_t)I) break;
		const uint8_t *src = gtA + (((i >> 3) * L + l) * 8 + (i & 7)) * (size_t)pl;
		unsigned n = 0;
		for (int b = 0; b < pl; b++) n += (src[b] == m);
		out[x / PERWORD] |= n << (BITS * (x % PERWORD));
	}
	gtC[idx] = make_uint4(out[0], out[1], out[2], out[3]);
}

/* non-empty cells (i, c) with n_ic > 0 -- the unit SURVEY.md section 8d counts flops in -- and non-missing allele copies;
 * thread = (block of 8 individuals, column) over gtA, block partial sums, one atomic per block */
__global__ __launch_bounds__(256) void k_count_cells(const uint8_t *__restrict__ gtA, int I, int L, int pl, int T,
		const int32_t *__restrict__ col_locus, const uint8_t *__restrict__ col_allele, unsigned long long *out)
{
	__shared__ unsigned red[2][256];
	const size_t n = (size_t)((I + 7) / 8) * T, stride = (size_t)gridDim.x * 256;
	unsigned cells = 0, copies = 0;
	for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < n; idx += stride) {
		const int c = (int)(idx % T);
		const size_t ib = idx / T;
		const int l = col_locus[c];
		const unsigned m = col_allele[c];
		const uint8_t *src = gtA + (ib * L + l) * 8 * (size_t)pl;
		for (int j = 0; j < 8; j++) {
			unsigned cnt = 0;
			for (int b = 0; b < pl; b++) cnt += (src[j * pl + b] == m);	/* padded individuals carry 0xFF */
			cells += cnt != 0;
			copies += cnt;
		}
	}
	red[0][threadIdx.x] = cells;
	red[1][threadIdx.x] = copies;
	__syncthreads();
	for (int w = 128; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) {
			red[0][threadIdx.x] += red[0][threadIdx.x + w];
			red[1][threadIdx.x] += red[1][threadIdx.x + w];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		atomicAdd(&out[0], (unsigned long long)red[0][0]);
		atomicAdd(&out[1], (unsigned long long)red[1][0]);
	}
}

/* host order [K][T] <-> device order [T][K] */
__global__ void k_transpose_kt_to_tk(const double *__restrict__ src, double *__restrict__ dst, int K, int T)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (size_t)K * T) return;
	const int k = (int)(idx % K);
	const size_t c = idx / K;
	dst[idx] = src[(size_t)k * T + c];
}
__global__ void k_transpose_tk_to_kt(const double *__restrict__ src, double *__restrict__ dst, int K, int T)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= (size_t)K * T) return;
	const size_t c = idx % T;
	const int k = (int)(idx / T);
	dst[idx] = src[c * K + k];
}

/* acc + in[first] + in[first + stride] + ... (count terms, added in that order): the loads of eight terms are issued
 * together, the additions keep their order, so the result is the plain loop's to the last bit */
__device__ __forceinline__ double ordered_sum(double acc, const double *__restrict__ in, size_t first, size_t stride, int count)
{
	int x = 0;
	for (; x + 8 <= count; x += 8) {
		double v[8];
#pragma unroll
		for (int y = 0; y < 8; y++) v[y] = in[first + (size_t)(x + y) * stride];
#pragma unroll
		for (int y = 0; y < 8; y++) acc += v[y];
	}
	for (; x < count; x++) acc += in[first + (size_t)x * stride];
	return acc;
}

/* deterministic sum of n doubles by one block (fixed strided order, then a fixed tree); every thread must call it; the result is
 * valid in thread 0 (and red[] may be reused after the trailing barrier) */
__device__ __forceinline__ double block_ordered_sum(const double *__restrict__ in, int n, double *red)
{
	const double s = (int)threadIdx.x < n ? ordered_sum(0.0, in, threadIdx.x, MCHIP_BLOCK, (n - (int)threadIdx.x + MCHIP_BLOCK - 1) / MCHIP_BLOCK) : 0.0;
	red[threadIdx.x] = s;
	__syncthreads();
	for (int w = MCHIP_BLOCK / 2; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
		__syncthreads();
	}
	const double v = red[0];
	__syncthreads();
	return v;
}
__global__ __launch_bounds__(MCHIP_BLOCK) void k_reduce_sum(const double *__restrict__ in, int n, double *out, const int *stop = nullptr)
{
	__shared__ double red[MCHIP_BLOCK];
	if (stop && *stop) return;
	const double v = block_ordered_sum(in, n, red);
	if (threadIdx.x == 0) *out = v;
}
/* the end of a batched accelerated cycle in one single-block launch: emll (when in1 is given: the dual pass's second log
 * likelihood) -> sc[1], the extrapolated point's log likelihood -> sc[2] (k_reduce_sum's sums), accept iff ll > emll */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_reduce_accept(const double *__restrict__ in1, const double *__restrict__ in2, int n, double *sc,
		int *cyc, const int *stop)
{
	__shared__ double red[MCHIP_BLOCK];
	if (stop && *stop) return;
	if (in1) {
		const double v = block_ordered_sum(in1, n, red);
		if (threadIdx.x == 0) sc[1] = v;
	}
	const double v2 = block_ordered_sum(in2, n, red);
	if (threadIdx.x == 0) {
		sc[2] = v2;
		cyc[1] = (!cyc[0] && sc[2] > sc[1]) ? 1 : 0;
	}
}

/* the 2 * nout sums of a dot-product pass in one launch (block x < nout: eta part x -> sc[16 + x]; the others: p part -> sc[20 + .]):
 * each block is k_reduce_sum on its array, same order, same tree */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_reduce_dots(const double *__restrict__ part_q, int gq, const double *__restrict__ part_p, int gp,
		int nout, double *sc, const int *stop)
{
	__shared__ double red[MCHIP_BLOCK];
	if (stop && *stop) return;
	const int x = blockIdx.x;
	const double *in = x < nout ? part_q + (size_t)x * gq : part_p + (size_t)(x - nout) * gp;
	const int n = x < nout ? gq : gp;
	double *out = x < nout ? sc + 16 + x : sc + 20 + (x - nout);
	const double v = block_ordered_sum(in, n, red);
	if (threadIdx.x == 0) *out = v;
}

/* out[e] = sum over slabs of slabs[j][e], in a fixed order: a block handles 32 elements x 8 slab lanes (lane s adds
 * slabs s, s+8, ...; the 8 partial sums are then added in lane order), so the many per-chunk partial-sum slabs of the
 * individual pass are combined with element- AND slab-level parallelism instead of one serial loop per individual */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_sum_slabs(const double *__restrict__ slabs, int n_slabs, size_t n, double *out,
		const int *stop)
{
	__shared__ double part[8][32];
	if (stop && *stop) return;
	const int e_local = threadIdx.x & 31, s = threadIdx.x >> 5;
	const size_t e = (size_t)blockIdx.x * 32 + e_local;
	double acc = 0.0;
	if (e < n && s < n_slabs) acc = ordered_sum(0.0, slabs, (size_t)s * n + e, 8 * n, (n_slabs - s + 7) / 8);
	part[s][e_local] = acc;
	__syncthreads();
	if (s == 0 && e < n) {
		double t = part[0][e_local];
#pragma unroll
		for (int x = 1; x < 8; x++) t += part[x][e_local];
		out[e] = t;
	}
}


/* P[to][l,.][k] = normalise(P[from] * sum_chunks Apart) then project (em_alg.c:706-752); thread = (l,k).  Loci with at most 8
 * alleles (nearly always) are summed, normalised and projected in registers and written once; longer ones go through memory */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_finalize_p(int L, int K, int T, const int32_t *__restrict__ toff,
		int n_ichunks, const double *__restrict__ Apart, const double *Pfrom, double *Pto,
		int weighted, double add_lb, int do_projection, double lb, uint8_t *flags, const int *stop = nullptr)
{
	const size_t idx = (size_t)blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	if (idx >= (size_t)L * K || (stop && *stop)) return;
	const int k = (int)(idx % K);
	const int l = (int)(idx / K);
	const int c0 = toff[l], M = toff[l + 1] - c0;
	double temp = 0.0;
	if (M <= 8) {
		double v[8];
#pragma unroll
		for (int m = 0; m < 8; m++) {
			v[m] = 0.0;
			if (m < M) {
				const size_t e = (size_t)(c0 + m) * K + k;
				double s = ordered_sum(0.0, Apart, e, (size_t)T * K, n_ichunks);
				if (weighted) s *= Pfrom[e];
				s += add_lb;
				v[m] = s;
				temp += s;
			}
		}
#pragma unroll
		for (int m = 0; m < 8; m++)
			if (m < M) v[m] /= temp;
		if (do_projection) michelot_small(v, M, lb);
#pragma unroll
		for (int m = 0; m < 8; m++)
			if (m < M) Pto[(size_t)(c0 + m) * K + k] = v[m];
		return;
	}
	for (int m = 0; m < M; m++) {
		const size_t e = (size_t)(c0 + m) * K + k;
		double s = ordered_sum(0.0, Apart, e, (size_t)T * K, n_ichunks);
		if (weighted) s *= Pfrom[e];
		s += add_lb;		/* mixture M step starts every sum at p_lower_bound (em_alg.c:972); 0 otherwise */
		Pto[e] = s;
		temp += s;
	}
	for (int m = 0; m < M; m++) Pto[(size_t)(c0 + m) * K + k] /= temp;
	if (do_projection) michelot_strided(Pto + (size_t)c0 * K + k, K, M, lb, flags ? flags + (size_t)c0 * K + k : nullptr);
}

/* k_finalize_p with the memory side done by whole blocks (mchip_finalize.h: finalize_p_tile_body) */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_finalize_p_tile(int L, int K, int T, const int32_t *__restrict__ toff, int loci_per_block,
		int n_slabs, const double *__restrict__ Apart, const double *Pfrom, double *Pto,
		int weighted, double add_lb, int do_projection, double lb, const int *stop = nullptr)
{
	__shared__ double tile[FP_TILE];
	if (stop && *stop) return;
	finalize_p_tile_body(blockIdx.x, tile, L, K, T, toff, loci_per_block, n_slabs, Apart, Pfrom, Pto, weighted, add_lb, do_projection, lb);
}

/* loci per block of k_finalize_p_tile for this data set and K, or 0 where a locus can outgrow the tile (k_finalize_p then) */
static int finalize_p_loci(int K, int max_M)
{
	if (max_M > 64 || max_M < 1) return 0;
	const int n = FP_TILE / (max_M * K);
	return n >= 4 ? n : 0;
}

/* projection of every (l,k) block of a P slot (accelerated updates, accel_em.c:474-475) */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_project_p(int L, int K, const int32_t *__restrict__ toff, double *P,
		double lb, uint8_t *flags, const int *stop = nullptr)
{
	const size_t idx = (size_t)blockIdx.x * MCHIP_BLOCK + threadIdx.x;
	if (idx >= (size_t)L * K || (stop && *stop)) return;
	const int k = (int)(idx % K);
	const int l = (int)(idx / K);
	const int c0 = toff[l], M = toff[l + 1] - c0;
	if (M <= 8) {
		double v[8];
#pragma unroll
		for (int m = 0; m < 8; m++) v[m] = m < M ? P[(size_t)(c0 + m) * K + k] : 0.0;
		michelot_small(v, M, lb);
#pragma unroll
		for (int m = 0; m < 8; m++)
			if (m < M) P[(size_t)(c0 + m) * K + k] = v[m];
		return;
	}
	michelot_strided(P + (size_t)c0 * K + k, K, M, lb, flags ? flags + (size_t)c0 * K + k : nullptr);
}

/* shared eta (eta_constrained): eta_k = sum_i S_ik / sum, project (em_alg.c:604-648): one block per k */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_column_sums(const double *__restrict__ sik, int I, int K, double *out, const int *stop = nullptr)
{
	__shared__ double red[MCHIP_BLOCK];
	if (stop && *stop) return;
	const int k = blockIdx.x;
	double s = 0.0;
	for (int i = threadIdx.x; i < I; i += MCHIP_BLOCK) s += sik[(size_t)i * K + k];
	red[threadIdx.x] = s;
	__syncthreads();
	for (int w = MCHIP_BLOCK / 2; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
		__syncthreads();
	}
	if (threadIdx.x == 0) out[k] = red[0];
}
__global__ void k_normalize_row(const double *__restrict__ sums, int K, double *eta, const int *stop = nullptr, double add = 0.0)
{
	if (threadIdx.x || blockIdx.x || (stop && *stop)) return;
	double temp = 0.0;
	for (int k = 0; k < K; k++) temp += sums[k] + add;
	for (int k = 0; k < K; k++) eta[k] = (sums[k] + add) / temp;
}

/* mixture model: log P table; the E step skips p == 0 cells (em_alg.c:797-804), logL_mixture does not
 * (log_likelihood.c:197-200) */
__global__ void k_logp(const double *__restrict__ p, double *__restrict__ out, size_t n, int skip_zero, const int *stop = nullptr)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n || (stop && *stop)) return;
	const double v = p[idx];
	out[idx] = (skip_zero && v == 0.0) ? 0.0 : log(v);
}

/* stop() + stop_condition() + converged() of em_alg.c:101-182 on the device (time limit excepted).  One block: when `part` is
 * given the block first sums the E step's n partial log likelihoods exactly as k_reduce_sum does (same strided order, same tree:
 * the batched loops stay bit-identical to the call-by-call ones) and stores the sum in *ll -- one launch less per iteration,
 * which is what a small data set's step time is made of; then thread 0 applies the rule. */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_stop_check(mchip_run_state *s, double *ll, const double *__restrict__ part = nullptr, int n = 0)
{
	__shared__ double red[MCHIP_BLOCK];
	if (s->stopped) return;		/* uniform */
	if (part) {
		const double acc = (int)threadIdx.x < n ? ordered_sum(0.0, part, threadIdx.x, MCHIP_BLOCK, (n - (int)threadIdx.x + MCHIP_BLOCK - 1) / MCHIP_BLOCK) : 0.0;
		red[threadIdx.x] = acc;
		__syncthreads();
		for (int w = MCHIP_BLOCK / 2; w > 0; w >>= 1) {
			if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
			__syncthreads();
		}
		if (threadIdx.x == 0) *ll = red[0];
	}
	if (threadIdx.x) return;
	const double loglik = part ? red[0] : *ll;
	s->n_iter++;
	if (loglik != loglik) {
		s->fatal = 1;
		s->bad_loglik = loglik;
		s->stopped = 1;
		return;
	}
	int stop;
	if (s->max_iter && s->n_iter > s->max_iter) {
		s->iter_stop = 1;
		stop = 1;
	} else {
		double abs_diff = 0, rel_diff = 0;
		stop = 1;
		if (s->abs_error != 0) abs_diff = fabs(loglik - s->logL);
		if (s->rel_error != 0) rel_diff = abs_diff / fabs(s->logL);
		if (s->abs_error != 0 && abs_diff > s->abs_error) stop = 0;
		if (s->rel_error != 0 && rel_diff > s->rel_error) stop = 0;
		if (stop) s->converged = 1;
	}
	s->stopped = stop;
	if (loglik < s->logL && !stop) {
		s->fatal = 2;
		s->bad_loglik = loglik;
		s->stopped = 1;
		return;
	}
	s->logL = loglik;
}

/* secant: out = x_to - x_from (em_alg.c:1104-1161) */
__global__ void k_diff(const double *__restrict__ a, const double *__restrict__ b, double *out, size_t n, const int *stop = nullptr)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (stop && *stop) return;
	if (idx < n) out[idx] = a[idx] - b[idx];
}

/* three dot products of accel_em.c:143-184 (mode 0: utu, u(v-u), (v-u)^2) or the two of 291-310
 * (mode 1: u1.u2, u1.v2) over one array pair; block partials, combined by k_reduce_sum */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_dots(const double *__restrict__ u, const double *__restrict__ v,
		const double *__restrict__ u2, size_t n, int mode, double *part, const int *stop = nullptr)
{
	__shared__ double red[3][MCHIP_BLOCK];
	if (stop && *stop) return;
	double s0 = 0, s1 = 0, s2 = 0;
	for (size_t x = (size_t)blockIdx.x * MCHIP_BLOCK + threadIdx.x; x < n; x += (size_t)gridDim.x * MCHIP_BLOCK) {
		if (mode == 0) {
			const double uu = u[x], d = v[x] - uu;
			s0 += uu * uu; s1 += uu * d; s2 += d * d;
		} else {
			const double uu = u[x];
			s0 += uu * u2[x]; s1 += uu * v[x];
		}
	}
	red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
	__syncthreads();
	for (int w = MCHIP_BLOCK / 2; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) {
			red[0][threadIdx.x] += red[0][threadIdx.x + w];
			red[1][threadIdx.x] += red[1][threadIdx.x + w];
			red[2][threadIdx.x] += red[2][threadIdx.x + w];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		part[blockIdx.x] = red[0][0];
		part[gridDim.x + blockIdx.x] = red[1][0];
		part[2 * gridDim.x + blockIdx.x] = red[2][0];
	}
}

/* k_dots (mode 0) and k_accel_update of a batched cycle straight from the three iterates: u = x1 - x0 and v = x2 - x1 are formed
 * where they are used -- the values k_diff would have stored, so the sums and the update keep their bits -- and the cycle has
 * four launches and 100 MB of traffic less */
/* block `bid` of `gdim` of the dot-product pass over one array (u = x1 - x0, v = x2 - x1 formed here): grid-stride partial sums, a
 * fixed tree per block, three partials per block */
__device__ __forceinline__ void dots_slots_body(const double *__restrict__ x0, const double *__restrict__ x1, const double *__restrict__ x2,
		size_t n, double *part, unsigned bid, unsigned gdim, double (*red)[MCHIP_BLOCK])
{
	double s0 = 0, s1 = 0, s2 = 0;
	for (size_t x = (size_t)bid * MCHIP_BLOCK + threadIdx.x; x < n; x += (size_t)gdim * MCHIP_BLOCK) {
		const double uu = x1[x] - x0[x], vv = x2[x] - x1[x];
		const double d = vv - uu;
		s0 += uu * uu; s1 += uu * d; s2 += d * d;
	}
	red[0][threadIdx.x] = s0; red[1][threadIdx.x] = s1; red[2][threadIdx.x] = s2;
	__syncthreads();
	for (int w = MCHIP_BLOCK / 2; w > 0; w >>= 1) {
		if ((int)threadIdx.x < w) {
			red[0][threadIdx.x] += red[0][threadIdx.x + w];
			red[1][threadIdx.x] += red[1][threadIdx.x + w];
			red[2][threadIdx.x] += red[2][threadIdx.x + w];
		}
		__syncthreads();
	}
	if (threadIdx.x == 0) {
		part[bid] = red[0][0];
		part[gdim + bid] = red[1][0];
		part[2 * gdim + bid] = red[2][0];
	}
}
/* the eta part (blocks 0 .. gq-1) and the p part (the other gp blocks) of the step-size dot products in one launch; every block
 * does what it did when the two parts were two launches */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_dots_slots(const double *__restrict__ q0, const double *__restrict__ q1,
		const double *__restrict__ q2, size_t nq, double *part_q, unsigned gq, const double *__restrict__ p0,
		const double *__restrict__ p1, const double *__restrict__ p2, size_t np, double *part_p, const int *stop)
{
	__shared__ double red[3][MCHIP_BLOCK];
	if (stop && *stop) return;
	if (blockIdx.x < gq) dots_slots_body(q0, q1, q2, nq, part_q, blockIdx.x, gq, red);
	else dots_slots_body(p0, p1, p2, np, part_p, blockIdx.x - gq, gridDim.x - gq, red);
}
/* x_B = x_A - 2 s u + s^2 (v - u) (or x_A + u + s v) for the eta array (threads below nq) and the p array in one launch */
__global__ void k_accel_update_slots(const double *__restrict__ q0, const double *q1, const double *__restrict__ q2, double *qout, size_t nq,
				     const double *__restrict__ p0, const double *p1, const double *__restrict__ p2, double *pout, size_t np,
				     int qn_form, const double *s_dev, const int *stop)
{
#pragma clang fp contract(off)
	size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= nq + np || (stop && *stop)) return;
	const double *x0 = q0, *x1 = q1, *x2 = q2;
	double *out = qout;
	if (idx >= nq) { idx -= nq; x0 = p0; x1 = p1; x2 = p2; out = pout; }
	const double s = *s_dev;
	const double u = x1[idx] - x0[idx], v = x2[idx] - x1[idx];	/* (out may be x1: read before written, element by element) */
	if (qn_form) out[idx] = x0[idx] + u + s * v;
	else out[idx] = x0[idx] - 2 * s * u + s * s * (v - u);
}

/* accel_em.c:444-541 element updates (projection follows in k_project_*).  Evaluated exactly as the reference's expression
 * (left to right, every product and sum rounded: no fused multiply-add): where the extrapolation cancels to about zero, the
 * last bits decide whether the projection clamps the entry to the lower bound */
__global__ void k_accel_update(const double *__restrict__ base, const double *__restrict__ u, const double *__restrict__ v,
			       double *out, size_t n, double s, int qn_form, const double *s_dev = nullptr, const int *stop = nullptr)
{
#pragma clang fp contract(off)
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= n || (stop && *stop)) return;
	if (s_dev) s = *s_dev;		/* batched accelerated runs: the step size was computed on the device */
	if (qn_form) out[idx] = base[idx] + u[idx] + s * v[idx];
	else out[idx] = base[idx] - 2 * s * u[idx] + s * s * (v[idx] - u[idx]);
}
__global__ void k_axpy2(const double *__restrict__ v, double *out, size_t n, double ca, double cb)
{
#pragma clang fp contract(off)
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx < n) out[idx] += v[idx] * ca * cb;	/* accel_em.c:387-393: v * Ainv * cutu */
}
__global__ void k_add(const double *__restrict__ a, const double *__restrict__ b, double *out, size_t n)
{
	const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx < n) out[idx] = a[idx] + b[idx];
}

/* step_size() of accel_em.c:130-243 on the device (batched accelerated runs): sc[16..18] / sc[20..22] are the eta / p
 * parts of utu, u(v-u), (v-u)^2 (eta terms first, accel_em.c:143-184); the step goes to sc[24]; cyc[0] = 1 when the
 * reference would leave the cycle without an update (NaN / infinite step, S3's sqrt(utu) < 1e-8 guard) */
__device__ __forceinline__ void step_size_body(double *sc, int scheme, int *cyc)
{
	const double utu = sc[16] + sc[20], utvu = sc[17] + sc[21], vutvu = sc[18] + sc[22];
	double s;
	if (scheme == 1) s = utu / utvu;
	else if (scheme == 2) s = utvu / vutvu;
	else if (scheme == 3) s = (sqrt(utu) < 1e-8) ? (double)NAN : -sqrt(utu / vutvu);
	else s = -utu / utvu;				/* QN with one secant */
	if (scheme < 4 && s > -1) s = -1;		/* SQUAREM clamp (accel_em.c:236-237); false for NaN */
	sc[24] = s;
	cyc[0] = (s != s || s - s != 0.0) ? 1 : 0;	/* isnan || isinf */
}
/* the six sums of the step-size dot products (k_reduce_dots's blocks, one after the other: same order, same tree, same bits) and
 * step_size (accel_em.c:130-243) in one single-block launch */
__global__ __launch_bounds__(MCHIP_BLOCK) void k_reduce_dots_step(const double *__restrict__ part_q, int gq, const double *__restrict__ part_p,
		int gp, double *sc, int scheme, int *cyc, const int *stop)
{
	__shared__ double red[MCHIP_BLOCK];
	if (stop && *stop) return;
	for (int x = 0; x < 6; x++) {
		const double *in = x < 3 ? part_q + (size_t)x * gq : part_p + (size_t)(x - 3) * gp;
		const int n = x < 3 ? gq : gp;
		const double v = block_ordered_sum(in, n, red);
		if (threadIdx.x == 0) sc[(x < 3 ? 16 : 20) + (x % 3)] = v;
	}
	if (threadIdx.x == 0) step_size_body(sc, scheme, cyc);
}
__global__ void k_accept(const double *sc, int *cyc, const int *stop)
{
	if (threadIdx.x || blockIdx.x || (stop && *stop)) return;
	cyc[1] = (!cyc[0] && sc[2] > sc[1]) ? 1 : 0;
}

/* the cycle's outcome goes back to the slot every cycle starts from: the extrapolated point if accepted, else the second
 * EM iterate (pindex = tindex / findex, accel_em.c:89-101), so that the slot roles are the same in every cycle */
__global__ void k_select_copy(double *qdst, const double *__restrict__ q_if_accepted, const double *__restrict__ q_otherwise, size_t nq,
			      double *pdst, const double *__restrict__ p_if_accepted, const double *__restrict__ p_otherwise, size_t np,
			      const int *cyc, const int *stop)
{
	size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
	if (idx >= nq + np || (stop && *stop)) return;
	if (idx < nq) qdst[idx] = cyc[1] ? q_if_accepted[idx] : q_otherwise[idx];
	else { idx -= nq; pdst[idx] = cyc[1] ? p_if_accepted[idx] : p_otherwise[idx]; }
}

/* ------------------------------------------------------------------ helpers */
static inline unsigned nblk(size_t n, unsigned b = 256) { return (unsigned)((n + b - 1) / b); }
/* for the byte-per-thread layout kernels (grid-stride loops): at most 2^30 work-items per launch */
static inline unsigned nblk_capped(size_t n, unsigned b = 256)
{
	const size_t blocks = (n + b - 1) / b, cap = ((size_t)1 << 30) / b;
	return (unsigned)(blocks < cap ? blocks : cap);
}

static int check_slot(mchip_context *ctx, int slot)
{
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!ctx->K) return fail(ctx, MCHIP_ERR_STATE, "no model set%s", nullptr);
	if (slot < 0 || slot > 2) return fail(ctx, MCHIP_ERR_INVALID, "slot out of range%s", nullptr);
	return MCHIP_OK;
}

static void prof_mark(mchip_context *ctx, int kind, bool start)
{
	if (!ctx->profiling) return;
	if (ctx->ev_used >= ctx->ev_pool.size()) {
		hipEvent_t e;
		if (hipEventCreate(&e) != hipSuccess) return;
		ctx->ev_pool.push_back(e);
	}
	(void)hipEventRecord(ctx->ev_pool[ctx->ev_used++], ctx->stream);
	if (start) ctx->ev_kind.push_back(kind);
}

/* the captured steps carry their kernel arguments by value (buffer addresses, the data set's has_missing flag) */
static void drop_graphs(mchip_context *ctx)
{
	for (int s = 0; s < 3; s++)
		if (ctx->step_graph[s]) { (void)MCHIP_WAIT(hipGraphExecDestroy(ctx->step_graph[s])); ctx->step_graph[s] = nullptr; }
	for (int s = 0; s < 3; s++)
		for (int m = 0; m < 5; m++)
			if (ctx->cycle_graph[s][m]) { (void)MCHIP_WAIT(hipGraphExecDestroy(ctx->cycle_graph[s][m])); ctx->cycle_graph[s][m] = nullptr; }
}

static void free_model(mchip_context *ctx)
{
	drop_graphs(ctx);
	for (int s = 0; s < 3; s++) { dfree(ctx->d_p[s]); dfree(ctx->d_q[s]); }
	for (int s = 0; s < MCHIP_MAX_SECANTS; s++) { dfree(ctx->d_up[s]); dfree(ctx->d_vp[s]); dfree(ctx->d_uq[s]); dfree(ctx->d_vq[s]); }
	dfree(ctx->d_sik); dfree(ctx->d_stage); dfree(ctx->d_logp); dfree(ctx->d_Apart); dfree(ctx->d_Spart); dfree(ctx->d_llpart); dfree(ctx->d_llpart2);
	dfree(ctx->d_redpart); dfree(ctx->d_flags);
	ctx->K = 0;
	ctx->parked_K = 0;
	ctx->kt = nullptr;
	ctx->have_ll = 0;
	ctx->s_cache_slot = -1;
}

static void free_data(mchip_context *ctx)
{
	dfree(ctx->d_ua); dfree(ctx->d_toff); dfree(ctx->d_col_locus); dfree(ctx->d_col_allele);
	dfree(ctx->d_gtA); dfree(ctx->d_gtS); dfree(ctx->d_gtC); dfree(ctx->d_asA); dfree(ctx->d_asS);
	dfree(ctx->d_initA); dfree(ctx->d_initS);
	dfree(ctx->d_draw); dfree(ctx->d_jump_hi); dfree(ctx->d_jump_lo);
	dfree(ctx->d_cand_span); dfree(ctx->d_cand_centers); dfree(ctx->d_cand_off);
	ctx->cand_span_bytes = ctx->cand_loci = ctx->cand_center_bytes = 0;
	ctx->n_jump_hi = 0;
	for (int x = 0; x < 2; x++) { dfree(ctx->lat[x].d_i); dfree(ctx->lat[x].d_r); ctx->lat[x].nA = 0; }
	dfree(ctx->d_part_slabs);
	ctx->part_slab_bytes = 0;
	ctx->init_geno_set = 0;
	ctx->h_ua.clear();
	ctx->I = ctx->L = ctx->T = 0;
}

static mchip_pass_args pass_args(mchip_context *ctx, int slot)
{
	mchip_pass_args a;
	memset(&a, 0, sizeof a);
	a.I = ctx->I; a.L = ctx->L; a.T = ctx->T; a.ploidy = ctx->ploidy; a.K = ctx->K;
	a.gtA = ctx->d_gtA; a.gtS = ctx->d_gtS; a.gtC = ctx->d_gtC; a.count_bits = ctx->count_bits; a.has_missing = ctx->has_missing;
	a.ua = ctx->d_ua; a.toff = ctx->d_toff; a.col_locus = ctx->d_col_locus; a.col_allele = ctx->d_col_allele;
	a.P = ctx->d_p[slot]; a.Q = ctx->d_q[slot]; a.qstride = ctx->qstride;
	a.ichunk = ctx->ichunk; a.n_ichunks = ctx->n_ichunks; a.Apart = ctx->d_Apart; a.llpart = ctx->d_llpart;
	a.flush_blocks = ctx->flush_blocks;
	a.safe_rcp = ctx->safe_rcp;
	a.lchunk = ctx->lchunk; a.n_lchunks = ctx->n_lchunks; a.Spart = ctx->d_Spart;
	a.asA = ctx->d_asA; a.asS = ctx->d_asS;
	a.sparse = ctx->sparse; a.tile_cols = 8 * ctx->max_M;
	a.biallelic = (ctx->min_M == 2 && ctx->max_M == 2 && !ctx->knob.no_bial) ? 1 : 0;
	a.ind_waves = ctx->ind_waves;
	a.xcd_rows = ctx->xcd_rows;
	a.no_col_split = ctx->knob.no_col_split;
	return a;
}

/* P[to] from n_slabs slabs of N-side sums: the tiled form where the data set allows it */
static void launch_finalize_p(mchip_context *ctx, int n_slabs, const double *slabs, int from, int to, int weighted, double add_lb,
			      int do_projection, const int *stop)
{
	const int per = finalize_p_loci(ctx->K, ctx->max_M);
	if (per)
		hipLaunchKernelGGL(k_finalize_p_tile, dim3((ctx->L + per - 1) / per), dim3(MCHIP_BLOCK), 0, ctx->stream,
				   ctx->L, ctx->K, ctx->T, ctx->d_toff, per, n_slabs, slabs, ctx->d_p[from], ctx->d_p[to],
				   weighted, add_lb, do_projection, ctx->p_lb, stop);
	else
		hipLaunchKernelGGL(k_finalize_p, dim3(nblk((size_t)ctx->L * ctx->K)), dim3(MCHIP_BLOCK), 0, ctx->stream,
				   ctx->L, ctx->K, ctx->T, ctx->d_toff, n_slabs, slabs, ctx->d_p[from], ctx->d_p[to],
				   weighted, add_lb, do_projection, ctx->p_lb, ctx->d_flags, stop);
}

/* the "bad input" word of the layout kernels: the last double of d_scalars (slots 0..47 are in use), so that an
 * initialisation allocates and frees nothing (hipFree waits for the whole device, i.e. for every other stream's fits) */
static int *bad_flag(mchip_context *ctx) { return reinterpret_cast<int *>(ctx->d_scalars + 63); }

/* ------------------------------------------------------------------ C-ABI */
extern "C" {

int mchip_abi_version(void) { return MCHIP_ABI_VERSION; }

int mchip_device_count(int *count)
{
	MCHIP_ENTRY();
	if (!count) return MCHIP_ERR_INVALID;
	int n = 0;
	if (MCHIP_WAIT(hipGetDeviceCount(&n)) != hipSuccess) n = 0;
	*count = n;
	return MCHIP_OK;
}

int mchip_create(mchip_context **out, int device)
{
	MCHIP_ENTRY();
	if (!out) return MCHIP_ERR_INVALID;
	*out = nullptr;
	int n = 0;
	if (MCHIP_WAIT(hipGetDeviceCount(&n)) != hipSuccess || n <= 0) return MCHIP_ERR_NO_DEVICE;	/* a process's first runtime call: hipInit */
	if (device < 0 || device >= n) return MCHIP_ERR_INVALID;
	mchip_context *ctx = new mchip_context();
	ctx->first_empty = -1;
	ctx->device = device;
	ctx->err[0] = 0;
	ctx->ind_waves = 1;
	read_knobs(ctx);
	if (MCHIP_WAIT(hipSetDevice(device)) != hipSuccess || MCHIP_WAIT(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
		delete ctx;
		return MCHIP_ERR_HIP;
	}
	hipDeviceProp_t prop;
	ctx->n_cu = (MCHIP_WAIT(hipGetDeviceProperties(&prop, device)) == hipSuccess) ? prop.multiProcessorCount : 256;
	if (MCHIP_WAIT(hipHostMalloc((void **)&ctx->h_pinned, 64 * sizeof(double), hipHostMallocDefault)) != hipSuccess ||
	    MCHIP_WAIT(hipMalloc((void **)&ctx->d_scalars, 64 * sizeof(double))) != hipSuccess ||
	    MCHIP_WAIT(hipMalloc((void **)&ctx->d_run, sizeof(mchip_run_state))) != hipSuccess ||
	    MCHIP_WAIT(hipMalloc((void **)&ctx->d_cyc, 4 * sizeof(int))) != hipSuccess ||
	    MCHIP_WAIT(hipEventCreate(&ctx->ev_begin)) != hipSuccess || MCHIP_WAIT(hipEventCreate(&ctx->ev_end)) != hipSuccess) {
		delete ctx;
		return MCHIP_ERR_ALLOC;
	}
	*out = ctx;
	return MCHIP_OK;
}

int mchip_destroy(mchip_context *ctx)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_OK;
	(void)MCHIP_WAIT(hipSetDevice(ctx->device));
	(void)MCHIP_WAIT(hipStreamSynchronize(ctx->stream));
	free_model(ctx);
	free_data(ctx);
	dfree(ctx->d_scalars);
	dfree(ctx->d_cyc);
	dfree(ctx->d_run);
	if (ctx->h_pinned) (void)MCHIP_WAIT(hipHostFree(ctx->h_pinned));
	for (hipEvent_t e : ctx->ev_pool) (void)MCHIP_WAIT(hipEventDestroy(e));
	(void)MCHIP_WAIT(hipEventDestroy(ctx->ev_begin));
	(void)MCHIP_WAIT(hipEventDestroy(ctx->ev_end));
	(void)MCHIP_WAIT(hipStreamDestroy(ctx->stream));
	delete ctx;
	return MCHIP_OK;
}

const char *mchip_last_error(const mchip_context *ctx) { return ctx ? ctx->err : "null context"; }

int mchip_synchronize(mchip_context *ctx)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

int mchip_device_info(mchip_context *ctx, char *name, int name_len, int *compute_units, double *hbm_bytes)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	hipDeviceProp_t prop;
	HIPCHK(hipGetDeviceProperties(&prop, ctx->device));
	if (name && name_len > 0) snprintf(name, name_len, "%s (%s)", prop.name, prop.gcnArchName);
	if (compute_units) *compute_units = prop.multiProcessorCount;
	if (hbm_bytes) *hbm_bytes = (double)prop.totalGlobalMem;
	return MCHIP_OK;
}

/* shape of a data set: tables derived from uniquealleles, genotype buffers allocated but not filled */
static int set_shape_impl(mchip_context *ctx, int I, int L, int ploidy, const int32_t *ua, int keep_init);

static int set_shape(mchip_context *ctx, int I, int L, int ploidy, const int32_t *ua, int keep_init = 0)
{
	const int rc = set_shape_impl(ctx, I, L, ploidy, ua, keep_init);
	if (rc == MCHIP_ERR_HIP || rc == MCHIP_ERR_ALLOC) {	/* an allocation that failed half way: no data set, no model */
		free_model(ctx);
		free_data(ctx);
	}
	return rc;
}

static int set_shape_impl(mchip_context *ctx, int I, int L, int ploidy, const int32_t *ua, int keep_init)
{
	if (I <= 0 || L <= 0 || ploidy <= 0 || ploidy > 64 || !ua)
		return fail(ctx, MCHIP_ERR_INVALID, "set_genotypes: bad shape or null pointer%s", nullptr);
	read_knobs(ctx);
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	/* The same shape and allele lists as the data set held (the next bootstrap replicate, a re-upload): every buffer stays;
	 * the model is dropped as the contract says, but its buffers are parked for an mchip_set_model with the same arguments.
	 * Freeing and re-allocating ~10 GB per replicate costs little per call, but the runtime returns freed memory lazily and
	 * stalls for seconds once the card's memory has been through its hands (measured: every 11th config-5 replicate) */
	if (ctx->T && ctx->I == I && ctx->L == L && ctx->ploidy == ploidy && (int)ctx->h_ua.size() == L &&
	    !memcmp(ctx->h_ua.data(), ua, sizeof(int32_t) * (size_t)L)) {
		if (ctx->K) ctx->parked_K = ctx->K;
		ctx->K = 0;
		ctx->have_ll = 0;
		ctx->s_cache_slot = -1;
		if (!keep_init) ctx->init_geno_set = 0;
		return MCHIP_OK;
	}
	free_model(ctx);	/* workspaces depend on T */
	free_data(ctx);

	std::vector<int32_t> toff(L + 25);	/* padded: kernels read the offsets of a block of 8 loci and of the two blocks behind it */
	toff[0] = 0;
	int maxM = 0, minM = 1 << 30;
	for (int l = 0; l < L; l++) {
		if (ua[l] < minM) minM = ua[l];
		if (ua[l] < 0 || ua[l] > 255) return fail(ctx, MCHIP_ERR_INVALID, "uniquealleles[l] must be in [0,255]%s", nullptr);
		if ((long long)toff[l] + ua[l] > 2000000000LL) return fail(ctx, MCHIP_ERR_INVALID, "too many allele columns%s", nullptr);
		toff[l + 1] = toff[l] + ua[l];
		if (ua[l] > maxM) maxM = ua[l];
	}
	const int T = toff[L];
	for (int x = L + 1; x < L + 25; x++) toff[x] = T;
	if (T <= 0) return fail(ctx, MCHIP_ERR_INVALID, "no alleles%s", nullptr);
	std::vector<int32_t> col_locus(T);
	std::vector<uint8_t> col_allele(T);
	for (int l = 0; l < L; l++)
		for (int m = 0; m < ua[l]; m++) {
			col_locus[toff[l] + m] = l;
			col_allele[toff[l] + m] = (uint8_t)m;
		}
	ctx->I = I; ctx->L = L; ctx->ploidy = ploidy; ctx->T = T; ctx->max_M = maxM; ctx->min_M = minM;
	ctx->h_ua.assign(ua, ua + L);
	ctx->geno_bytes_A = (size_t)((I + 7) / 8) * L * 8 * ploidy;
	ctx->geno_bytes_S = (size_t)((L + 7) / 8) * I * 8 * ploidy;
	HIPCHK(hipMalloc((void **)&ctx->d_ua, sizeof(int32_t) * L));
	HIPCHK(hipMalloc((void **)&ctx->d_toff, sizeof(int32_t) * (L + 25)));
	HIPCHK(hipMalloc((void **)&ctx->d_col_locus, sizeof(int32_t) * T));
	HIPCHK(hipMalloc((void **)&ctx->d_col_allele, T));
	HIPCHK(hipMalloc((void **)&ctx->d_gtA, ctx->geno_bytes_A));
	HIPCHK(hipMalloc((void **)&ctx->d_gtS, ctx->geno_bytes_S));
	HIPCHK(hipMemcpyAsync(ctx->d_ua, ua, sizeof(int32_t) * L, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(ctx->d_toff, toff.data(), sizeof(int32_t) * (L + 25), hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(ctx->d_col_locus, col_locus.data(), sizeof(int32_t) * T, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(ctx->d_col_allele, col_allele.data(), T, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));	/* the host vectors go out of scope */
	return MCHIP_OK;
}

/* one byte per allele copy in stream order, padded to whole generator chunks: the device-drawn partition, a generated data
 * set before it goes into the kernels' layouts, an uploaded genotype on its way there.  Kept for the life of the data set. */
static int stream_buffer(mchip_context *ctx)
{
	if (ctx->d_draw) return MCHIP_OK;
	const size_t n = (size_t)ctx->I * ctx->L * ctx->ploidy;
	HIPCHK(hipMalloc((void **)&ctx->d_draw, ((n + RNG_CHUNK - 1) / RNG_CHUNK) * RNG_CHUNK));
	return MCHIP_OK;
}

/* gtA / gtS are in place: flags and the packed per-column allele counts of the column pass */
static int install_layouts(mchip_context *ctx, int has_missing)
{
	const int I = ctx->I, L = ctx->L, ploidy = ctx->ploidy, T = ctx->T;
	if (ctx->has_missing != has_missing) drop_graphs(ctx);	/* parked model buffers: the kernel variant changes */
	ctx->has_missing = has_missing;
	ctx->counts_valid = 0;	/* counted when asked for (mchip_data_counts): a bootstrap replicate never asks */
	ctx->count_bits = ploidy <= 3 ? 2 : (ploidy <= 15 ? 4 : 0);
	if (ctx->knob.no_counts) ctx->count_bits = 0;
	if (ctx->count_bits) {
		const int G = 128 / ctx->count_bits;
		const size_t nwords = (size_t)((I + G - 1) / G) * T;
		if (!ctx->d_gtC) HIPCHK(hipMalloc((void **)&ctx->d_gtC, nwords * 16));
		if (ctx->count_bits == 2)
			hipLaunchKernelGGL(k_build_counts<2>, dim3(nblk(nwords)), dim3(256), 0, ctx->stream, ctx->d_gtA, I, L, ploidy, T,
					   ctx->d_col_locus, ctx->d_col_allele, (uint4 *)ctx->d_gtC);
		else
			hipLaunchKernelGGL(k_build_counts<4>, dim3(nblk(nwords)), dim3(256), 0, ctx->stream, ctx->d_gtA, I, L, ploidy, T,
					   ctx->d_col_locus, ctx->d_col_allele, (uint4 *)ctx->d_gtC);
		HIPCHK(hipGetLastError());
		HIPCHK(hipStreamSynchronize(ctx->stream));
	}
	return MCHIP_OK;
}

/* genotype held on the device as [I][L][ploidy] bytes -> the kernels' layouts (validated), packed counts */
static int install_raw(mchip_context *ctx, const uint8_t *d_raw)
{
	const int I = ctx->I, L = ctx->L, ploidy = ctx->ploidy;
	int *d_bad = bad_flag(ctx);
	HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
	if (launch_relayout(ctx->stream, d_raw, I, L, ploidy, ctx->d_ua, 0, ctx->d_gtA, ctx->d_gtS, d_bad))
		return fail(ctx, MCHIP_ERR_UNSUPPORTED, "data set too large for the layout kernel%s", nullptr);
	HIPCHK(hipGetLastError());
	int bad = 0;
	HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	if (bad & 1) {
		free_data(ctx);
		return fail(ctx, MCHIP_ERR_INVALID, "genotype allele index >= uniquealleles[l]%s", nullptr);
	}
	return install_layouts(ctx, (bad & 2) ? 1 : 0);
}

int mchip_set_genotypes(mchip_context *ctx, int I, int L, int ploidy, const int32_t *ua, const uint8_t *geno)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!geno) return fail(ctx, MCHIP_ERR_INVALID, "set_genotypes: bad shape or null pointer%s", nullptr);
	int rc = set_shape(ctx, I, L, ploidy, ua);
	if (rc) return rc;
	const size_t raw_bytes = (size_t)I * L * ploidy;
	/* an individual whose every copy is missing (one pass that stops at each individual's first observed copy: O(I) on real data) */
	ctx->first_empty = -1;
	ctx->empty_rows.clear();
	for (int sl = 0; sl < 3; sl++) ctx->empty_rows_nan[sl] = 0;
	for (int i = 0; i < I; i++) {
		const uint8_t *row = geno + (size_t)i * L * ploidy;
		size_t x = 0;
		while (x < (size_t)L * ploidy && row[x] == MCHIP_MISSING) x++;
		if (x == (size_t)L * ploidy) ctx->empty_rows.push_back(i);
	}
	if (!ctx->empty_rows.empty()) ctx->first_empty = ctx->empty_rows[0];
	if ((rc = stream_buffer(ctx))) return rc;
	HIPCHK(hipMemcpyAsync(ctx->d_draw, geno, raw_bytes, hipMemcpyHostToDevice, ctx->stream));
	return install_raw(ctx, ctx->d_draw);
}

int mchip_copy_genotypes(mchip_context *ctx, const mchip_context *src)
{
	MCHIP_ENTRY();
	if (!ctx || !src || ctx == src) return MCHIP_ERR_INVALID;
	if (!src->T) return fail(ctx, MCHIP_ERR_STATE, "copy_genotypes: the source holds no data set%s", nullptr);
	if (src->device != ctx->device) return fail(ctx, MCHIP_ERR_UNSUPPORTED, "copy_genotypes: contexts on different devices%s", nullptr);
	int rc = set_shape(ctx, src->I, src->L, src->ploidy, src->h_ua.data(), 1);
	if (rc) return rc;
	HIPCHK(hipStreamSynchronize(src->stream));
	HIPCHK(hipMemcpyAsync(ctx->d_gtA, src->d_gtA, ctx->geno_bytes_A, hipMemcpyDeviceToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(ctx->d_gtS, src->d_gtS, ctx->geno_bytes_S, hipMemcpyDeviceToDevice, ctx->stream));
	if (src->count_bits) {
		const int G = 128 / src->count_bits;
		const size_t bytes = (size_t)((ctx->I + G - 1) / G) * ctx->T * 16;
		if (!ctx->d_gtC) HIPCHK(hipMalloc((void **)&ctx->d_gtC, bytes));
		HIPCHK(hipMemcpyAsync(ctx->d_gtC, src->d_gtC, bytes, hipMemcpyDeviceToDevice, ctx->stream));
	}
	if (ctx->has_missing != src->has_missing) drop_graphs(ctx);
	ctx->has_missing = src->has_missing;
	ctx->first_empty = src->first_empty;
	ctx->empty_rows = src->empty_rows;
	for (int sl = 0; sl < 3; sl++) ctx->empty_rows_nan[sl] = 0;
	ctx->count_bits = src->count_bits;
	ctx->counts_valid = src->counts_valid;
	ctx->nnz_cells = src->nnz_cells;
	ctx->n_copies = src->n_copies;
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

int mchip_set_init_genotypes(mchip_context *ctx, const uint8_t *geno)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!ctx->T) return fail(ctx, MCHIP_ERR_STATE, "no genotypes set%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	ctx->init_geno_set = 0;
	if (!geno) {
		dfree(ctx->d_initA);
		dfree(ctx->d_initS);
		return MCHIP_OK;
	}
	const size_t n = (size_t)ctx->I * ctx->L * ctx->ploidy;
	int *d_bad = bad_flag(ctx);
	int rc = stream_buffer(ctx);
	if (rc) return rc;
	uint8_t *d_obs = ctx->d_draw;
	if (!ctx->d_initA) HIPCHK(hipMalloc((void **)&ctx->d_initA, ctx->geno_bytes_A));
	if (!ctx->d_initS) HIPCHK(hipMalloc((void **)&ctx->d_initS, ctx->geno_bytes_S));
	HIPCHK(hipMemcpyAsync(d_obs, geno, n, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
	if (launch_relayout(ctx->stream, d_obs, ctx->I, ctx->L, ctx->ploidy, ctx->d_ua, 0, ctx->d_initA, ctx->d_initS, d_bad))
		return fail(ctx, MCHIP_ERR_UNSUPPORTED, "data set too large for the layout kernel%s", nullptr);
	HIPCHK(hipGetLastError());
	int bad = 0;
	HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	if (bad & 1) {
		dfree(ctx->d_initA);
		dfree(ctx->d_initS);
		return fail(ctx, MCHIP_ERR_INVALID, "init genotype allele index >= uniquealleles[l]%s", nullptr);
	}
	ctx->init_geno_set = 1;
	return MCHIP_OK;
}

int mchip_data_counts(mchip_context *ctx, uint64_t *nonempty_cells, uint64_t *allele_copies)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!ctx->T) return fail(ctx, MCHIP_ERR_STATE, "no genotypes set%s", nullptr);
	if (!ctx->counts_valid) {	/* integer sums: order does not matter */
		HIPCHK(hipSetDevice(ctx->device));
		unsigned long long *d_cnt = reinterpret_cast<unsigned long long *>(ctx->d_scalars + 60), h_cnt[2] = {0, 0};
		HIPCHK(hipMemsetAsync(d_cnt, 0, 2 * sizeof(unsigned long long), ctx->stream));
		const size_t n = (size_t)((ctx->I + 7) / 8) * ctx->T;
		hipLaunchKernelGGL(k_count_cells, dim3(nblk_capped(n) > 65536u ? 65536u : nblk_capped(n)), dim3(256), 0, ctx->stream, ctx->d_gtA,
				   ctx->I, ctx->L, ctx->ploidy, ctx->T, ctx->d_col_locus, ctx->d_col_allele, d_cnt);
		HIPCHK(hipGetLastError());
		HIPCHK(hipMemcpyAsync(h_cnt, d_cnt, sizeof h_cnt, hipMemcpyDeviceToHost, ctx->stream));
		HIPCHK(hipStreamSynchronize(ctx->stream));
		ctx->nnz_cells = h_cnt[0];
		ctx->n_copies = h_cnt[1];
		ctx->counts_valid = 1;
	}
	if (nonempty_cells) *nonempty_cells = ctx->nnz_cells;
	if (allele_copies) *allele_copies = ctx->n_copies;
	return MCHIP_OK;
}

int mchip_get_genotypes(mchip_context *ctx, uint8_t *geno)
{
	MCHIP_ENTRY();
	if (!ctx || !geno) return MCHIP_ERR_INVALID;
	if (!ctx->T) return fail(ctx, MCHIP_ERR_STATE, "no genotypes set%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	const size_t n = (size_t)ctx->I * ctx->L * ctx->ploidy;
	scoped_dev<uint8_t> d_raw;
	HIPCHK(d_raw.alloc(n));
	hipLaunchKernelGGL(k_unlayout, dim3(nblk_capped(n)), dim3(256), 0, ctx->stream, ctx->d_gtA, ctx->I, ctx->L, ctx->ploidy, d_raw.p);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(geno, d_raw, n, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

static int set_model_impl(mchip_context *ctx, int K, int admixture, int eta_constrained, int do_projection,
			  double eta_lb, double p_lb, int n_secants);

int mchip_set_model(mchip_context *ctx, int K, int admixture, int eta_constrained, int do_projection,
		    double eta_lb, double p_lb, int n_secants)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	const int rc = set_model_impl(ctx, K, admixture, eta_constrained, do_projection, eta_lb, p_lb, n_secants);
	if (rc == MCHIP_ERR_HIP || rc == MCHIP_ERR_ALLOC) free_model(ctx);	/* an allocation that failed half way leaves no model, not a partial one */
	return rc;
}

static int set_model_impl(mchip_context *ctx, int K, int admixture, int eta_constrained, int do_projection,
			  double eta_lb, double p_lb, int n_secants)
{
	if (!ctx->T) return fail(ctx, MCHIP_ERR_STATE, "set_model before set_genotypes%s", nullptr);
	if (K < 1) return fail(ctx, MCHIP_ERR_INVALID, "K must be >= 1%s", nullptr);
	if (K > MCHIP_MAX_K) return fail(ctx, MCHIP_ERR_UNSUPPORTED, "K > MCHIP_MAX_K is not built%s", nullptr);
	if (n_secants < 0 || n_secants > MCHIP_MAX_SECANTS) return fail(ctx, MCHIP_ERR_INVALID, "n_secants out of range%s", nullptr);
	/* the element-per-thread kernels over parameters take one work-item per entry of P or Q */
	if ((size_t)K * ctx->T >= ((size_t)1 << 31) || (size_t)K * ctx->I >= ((size_t)1 << 31))
		return fail(ctx, MCHIP_ERR_UNSUPPORTED, "K*T or K*I of 2^31 or more is not supported%s", nullptr);
	read_knobs(ctx);
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	{	/* the buffers of this very model are still allocated (same data shape, same arguments): zero the parameters and go */
		const int have = ctx->K ? ctx->K : ctx->parked_K;
		if (have == K && ctx->d_p[0] && ctx->sig_admixture == admixture && ctx->sig_constrained == (eta_constrained ? 1 : 0) &&
		    ctx->sig_projection == do_projection && ctx->sig_nsec == n_secants && ctx->sig_eta_lb == eta_lb && ctx->sig_p_lb == p_lb) {
			const size_t KTr = (size_t)K * ctx->T;
			for (int s = 0; s < 3; s++) {
				HIPCHK(hipMemsetAsync(ctx->d_p[s], 0, KTr * sizeof(double), ctx->stream));
				HIPCHK(hipMemsetAsync(ctx->d_q[s], 0, (size_t)ctx->nq * sizeof(double), ctx->stream));
			}
			HIPCHK(hipMemsetAsync(ctx->d_sik, 0, (size_t)ctx->I * K * sizeof(double), ctx->stream));
			ctx->K = K;
			ctx->parked_K = 0;
			ctx->have_ll = 0;
			ctx->s_cache_slot = -1;
			HIPCHK(hipStreamSynchronize(ctx->stream));
			return MCHIP_OK;
		}
	}
	free_model(ctx);
	ctx->sig_admixture = admixture; ctx->sig_constrained = eta_constrained ? 1 : 0; ctx->sig_projection = do_projection;
	ctx->sig_nsec = n_secants; ctx->sig_eta_lb = eta_lb; ctx->sig_p_lb = p_lb;
	ctx->K = K; ctx->admixture = admixture; ctx->constrained = eta_constrained ? 1 : 0;
	ctx->do_projection = do_projection; ctx->eta_lb = eta_lb; ctx->p_lb = p_lb; ctx->nsec = n_secants;
	ctx->kt = mchip_get_ktable(K);
	const int shared_eta = (!admixture || eta_constrained);
	ctx->nq = shared_eta ? K : ctx->I * K;
	ctx->qstride = shared_eta ? 0 : K;
	const size_t KT = (size_t)K * ctx->T;
	for (int s = 0; s < 3; s++) {
		HIPCHK(hipMalloc((void **)&ctx->d_p[s], KT * sizeof(double)));
		HIPCHK(hipMalloc((void **)&ctx->d_q[s], (size_t)ctx->nq * sizeof(double)));
		HIPCHK(hipMemsetAsync(ctx->d_p[s], 0, KT * sizeof(double), ctx->stream));
		HIPCHK(hipMemsetAsync(ctx->d_q[s], 0, (size_t)ctx->nq * sizeof(double), ctx->stream));
	}
	for (int s = 0; s < n_secants; s++) {
		HIPCHK(hipMalloc((void **)&ctx->d_up[s], KT * sizeof(double)));
		HIPCHK(hipMalloc((void **)&ctx->d_vp[s], KT * sizeof(double)));
		HIPCHK(hipMalloc((void **)&ctx->d_uq[s], (size_t)ctx->nq * sizeof(double)));
		HIPCHK(hipMalloc((void **)&ctx->d_vq[s], (size_t)ctx->nq * sizeof(double)));
	}
	HIPCHK(hipMalloc((void **)&ctx->d_sik, (size_t)ctx->I * K * sizeof(double)));
	HIPCHK(hipMemsetAsync(ctx->d_sik, 0, (size_t)ctx->I * K * sizeof(double), ctx->stream));
	HIPCHK(hipMalloc((void **)&ctx->d_stage, KT * sizeof(double)));
	if (!admixture) HIPCHK(hipMalloc((void **)&ctx->d_logp, KT * sizeof(double)));

	/* launch geometry.  Both passes are FP64-issue-bound and every workgroup does the same amount of work, so
	 * the grid's last partial "round" of workgroups runs at low occupancy: measured at config 3, 8 workgroups
	 * per CU cost 6.9 ms per EM step, 64 per CU 5.4 ms (profiles/r01_geometry_sweep.txt).  More chunks mean
	 * more partial-sum slabs (Apart/Spart are written and re-read once per step), so the chunk count is capped
	 * where the slab bytes reach ~30 % of the genotype bytes the pass streams (config 2: 0.188 -> 0.179 ms per step against 15 %;
	 * config 3 reaches its 64 workgroups per CU before either cap).  The column pass stops at 15 % once the grid fills the
	 * device twice over: its slabs are K*T doubles each and k_finalize_p reads them all (config 5: 28 -> 15 slabs, 1.61 ->
	 * 1.56 ms per step; scripts/diag/geom.sh).  Chunk sizes are multiples of 8. */
	const int per_cu_col = ctx->knob.per_cu_col, per_cu_ind = ctx->knob.per_cu_ind;	/* tuning knobs (default 64) */
	int target = per_cu_col * ctx->n_cu;
	const int col_tiles = (ctx->T + MCHIP_BLOCK - 1) / MCHIP_BLOCK;
	const int iblocks = (ctx->I + 7) / 8, lblocks = (ctx->L + 7) / 8;
	const double slab_frac = ctx->knob.slab_frac;	/* tuning knob (default 0.3) */
	/* column pass: slab bytes per chunk 8*K*T, genotype bytes per chunk ichunk*L*ploidy */
	int min_ichunk = (int)ceil(8.0 * K * ctx->T / (slab_frac * ctx->L * ctx->ploidy));
	int want = (target + col_tiles - 1) / col_tiles;
	int cap = ctx->I / (min_ichunk > 0 ? min_ichunk : 1);
	{
		const int cap_lo = cap / 2;						/* half the slab budget */
		const int fill = (2 * 8 * ctx->n_cu + col_tiles - 1) / col_tiles;	/* two rounds of 8 resident workgroups per CU */
		const int soft = cap_lo > (fill < cap ? fill : cap) ? cap_lo : (fill < cap ? fill : cap);
		if (!ctx->knob.geometry_given && want > soft) want = soft;
	}
	if (want > cap) want = cap;
	if (want < 1) want = 1;
	if (want > iblocks) want = iblocks;
	if (ctx->count_bits && !ctx->knob.no_roundup) {	/* the packed-count column pass puts MCHIP_COL_WAVES chunks into a workgroup: whole workgroups where the caps allow */
		const int up = ((want + MCHIP_COL_WAVES - 1) / MCHIP_COL_WAVES) * MCHIP_COL_WAVES;
		if (up <= cap && up <= iblocks) want = up;
	}
	{
		const int gran = ctx->count_bits ? 128 / ctx->count_bits : 8;	/* individuals per packed word */
		const int per = (ctx->I + want - 1) / want;
		ctx->ichunk = ((per + gran - 1) / gran) * gran;
	}
	ctx->n_ichunks = (ctx->I + ctx->ichunk - 1) / ctx->ichunk;
	/* individual pass: slab bytes per chunk 8*K*I, genotype bytes per chunk lchunk*I*ploidy */
	/* (the sparse pass gives an individual mchip_ind_split(K) lanes; the dense and mixture kernels one, and do not use n_ll_ind) */
	const int ind_per_block = mchip_qblock(K) / (admixture ? mchip_ind_split(K) : 1);
	const int ind_tiles = (ctx->I + ind_per_block - 1) / ind_per_block;
	int min_lchunk = (int)ceil(8.0 * K / (slab_frac * ctx->ploidy));
	target = per_cu_ind * ctx->n_cu;
	want = (target + ind_tiles - 1) / ind_tiles;
	cap = ctx->L / (min_lchunk > 0 ? min_lchunk : 1);
	if (want > cap) want = cap;
	if (want < 1) want = 1;
	if (want > lblocks) want = lblocks;
	{	/* the sparse individual pass stages two tiles of 8 loci of P rows in LDS: use it while they fit 64 KiB */
		const size_t kp = (size_t)mchip_kp(K);
		const size_t lds = (2 * 8 * (size_t)ctx->max_M * kp + mchip_qblock(K)) * sizeof(double);
		ctx->sparse = (ctx->max_M <= MCHIP_SPARSE_MAX_M) && lds <= 65536 && !ctx->knob.force_dense;
	}
	/* cooperating waves (mchip_internal.h): chunks are still what ONE WAVE takes; a workgroup of the sparse individual-side kernels
	 * is ind_waves of them, so a whole number of workgroups wants a multiple of that many chunks where the caps allow it */
	ctx->ind_waves = (admixture && ctx->sparse && mchip_ind_split(K) == 1) ? mchip_ind_waves(K, 8 * ctx->max_M) : 1;
	const bool ind_coop = admixture && ctx->sparse && mchip_ind_split(K) == 1;
	if (ind_coop) {
		/* ... and a multiple of eight slab rows where possible: the sparse kernel then keeps a row on one XCD (coop_rows() in
		 * mchip_kernels_k.hip), and eight XCDs with 52 rows between them would have 7 and 6 each */
		const int w8 = 8 * ctx->ind_waves, w1 = ctx->ind_waves;
		const int up8 = ((want + w8 - 1) / w8) * w8, up1 = ((want + w1 - 1) / w1) * w1;
		if (up8 <= cap && up8 <= lblocks) want = up8;
		else if (up1 <= cap && up1 <= lblocks) want = up1;
	}
	ctx->lchunk = ((lblocks + want - 1) / want) * 8;
	ctx->n_lchunks = (ctx->L + ctx->lchunk - 1) / ctx->lchunk;
	ctx->xcd_rows = (ind_coop && ((ctx->n_lchunks + ctx->ind_waves - 1) / ctx->ind_waves) % 8 == 0) ? 1 : 0;
	{	/* partial log likelihoods any pass can leave: one per workgroup; no individual-side kernel has fewer than 64 individuals
		 * per workgroup, no column-side one fewer than 64 columns */
		const int by_col = ((ctx->T + 63) / 64) * ctx->n_ichunks, by_ind = ((ctx->I + 63) / 64) * ctx->n_lchunks;
		(void)col_tiles; (void)ind_tiles;
		ctx->n_llpart = by_col > by_ind ? by_col : by_ind;
	}
	HIPCHK(hipMalloc((void **)&ctx->d_Apart, (size_t)ctx->n_ichunks * KT * sizeof(double)));
	HIPCHK(hipMalloc((void **)&ctx->d_Spart, (size_t)ctx->n_lchunks * ctx->I * K * sizeof(double)));
	HIPCHK(hipMalloc((void **)&ctx->d_llpart, (size_t)ctx->n_llpart * sizeof(double)));
	HIPCHK(hipMalloc((void **)&ctx->d_llpart2, (size_t)ctx->n_llpart * sizeof(double)));
	HIPCHK(hipMalloc((void **)&ctx->d_redpart, (size_t)3 * 1024 * sizeof(double)));
	if (ctx->max_M > 64) {
		HIPCHK(hipMalloc((void **)&ctx->d_flags, KT));
	}
	/* log-product check interval: with every parameter >= its lower bound, t = sum_k q_k p_k >= p_lb / K, so
	 * floor(200 / -log10(t_min)) multiplications keep a product that starts above 1e-100 above 1e-300
	 * (DESIGN.md section 4).  A block of 8 loci (or individuals) multiplies 8*ploidy times. */
	{
		const double tmin = p_lb / K;
		int blocks = 0;
		if (do_projection && tmin > 0 && tmin < 1) {
			const double mults = floor(200.0 / -log10(tmin));
			/* the sparse tetraploid pass checks twice per block (every 4 loci = 16 multiplications) */
			blocks = (int)(mults / (ctx->ploidy == 4 ? 16.0 : 8.0 * ctx->ploidy));
			if (blocks > 1 << 16) blocks = 1 << 16;
		}
		ctx->flush_blocks = blocks;
		/* Supported lower bounds (--bound): any value >= 0.  The fast kernels share one reciprocal among the four cells of a
		 * column-pass step (1 / (t0 t1 t2 t3)) and between the two copies of a locus, which needs every t > 0 and
		 * t^4 >= DBL_MIN: guaranteed when projection is on and p_lb >= 1e-75 (t >= p_lb).  Otherwise -- projection off
		 * (--projection: an unobserved allele column, e.g. the phantom slot of a locus with missing data or an allele a
		 * bootstrap replicate lacks, has P = 0 for every k, so t = 0 in its zero-count cells) or a smaller bound -- the
		 * kernels take one reciprocal per non-empty cell and zero-count cells contribute exactly 0, as in em_alg.c:338-342.
		 * A non-empty cell with t = 0 is NaN here as it is in the reference (n * 0 / 0). */
		ctx->safe_rcp = (!do_projection || !(p_lb >= 1e-75)) ? 1 : 0;
		if (ctx->knob.force_safe) ctx->safe_rcp = 1;
		if (ctx->safe_rcp) ctx->flush_blocks = 0;
	}
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

int mchip_q_length(const mchip_context *ctx, int *n)
{
	MCHIP_ENTRY();
	if (!ctx || !n || !ctx->K) return MCHIP_ERR_STATE;
	*n = ctx->nq;
	return MCHIP_OK;
}
int mchip_p_length(const mchip_context *ctx, int *n)
{
	MCHIP_ENTRY();
	if (!ctx || !n || !ctx->K) return MCHIP_ERR_STATE;
	*n = ctx->K * ctx->T;
	return MCHIP_OK;
}

int mchip_set_p(mchip_context *ctx, int slot, const double *p)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	ctx->s_cache_slot = -1;
	if (!p) return fail(ctx, MCHIP_ERR_INVALID, "null pointer%s", nullptr);
	const size_t KT = (size_t)ctx->K * ctx->T;
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(ctx->d_stage, p, KT * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(k_transpose_kt_to_tk, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_stage, ctx->d_p[slot], ctx->K, ctx->T);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(ctx->stream));	/* the host buffer may be reused on return */
	return MCHIP_OK;
}

int mchip_get_p(mchip_context *ctx, int slot, double *p)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	if (!p) return fail(ctx, MCHIP_ERR_INVALID, "null pointer%s", nullptr);
	const size_t KT = (size_t)ctx->K * ctx->T;
	HIPCHK(hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_transpose_tk_to_kt, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_p[slot], ctx->d_stage, ctx->K, ctx->T);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(p, ctx->d_stage, KT * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

int mchip_set_q(mchip_context *ctx, int slot, const double *q)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	ctx->s_cache_slot = -1;
	if (!q) return fail(ctx, MCHIP_ERR_INVALID, "null pointer%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	/* A row of an individual without a single observed copy that comes in as NaN (what mchip_get_q reports for it after an M
	 * step: a warm start, a checkpoint) is kept on the device as the finite 1 / K k_finalize_q keeps -- a NaN there would reach
	 * every column's N-side sum through 0 * (1 / NaN) -- and the slot goes on reporting it as NaN; a finite row is taken as it is */
	int nan_rows = 0;
	if (ctx->qstride)
		for (int i : ctx->empty_rows)
			for (int k = 0; k < ctx->K; k++) nan_rows |= (q[(size_t)i * ctx->K + k] != q[(size_t)i * ctx->K + k]);
	if (nan_rows) {
		std::vector<double> clean(q, q + ctx->nq);
		for (int i : ctx->empty_rows) {
			bool bad = false;
			for (int k = 0; k < ctx->K; k++) bad |= (clean[(size_t)i * ctx->K + k] != clean[(size_t)i * ctx->K + k]);
			if (bad)
				for (int k = 0; k < ctx->K; k++) clean[(size_t)i * ctx->K + k] = 1.0 / ctx->K;
		}
		HIPCHK(hipMemcpyAsync(ctx->d_q[slot], clean.data(), (size_t)ctx->nq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
		HIPCHK(hipStreamSynchronize(ctx->stream));	/* `clean` goes out of scope */
	} else {
		HIPCHK(hipMemcpyAsync(ctx->d_q[slot], q, (size_t)ctx->nq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
		HIPCHK(hipStreamSynchronize(ctx->stream));
	}
	ctx->empty_rows_nan[slot] = nan_rows;
	return MCHIP_OK;
}

int mchip_get_q(mchip_context *ctx, int slot, double *q)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	if (!q) return fail(ctx, MCHIP_ERR_INVALID, "null pointer%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(q, ctx->d_q[slot], (size_t)ctx->nq * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	/* An individual without a single observed copy: the reference's M step gives it 0 / 0 (em_alg.c:685-690; printed "-nan") and the
	 * row then sits in its arrays untouched by anything else, every zero-count cell being skipped.  The device keeps a finite row
	 * for it (k_finalize_q: 1 / K) so that nothing it is multiplied into turns NaN, and it is reported here as the reference has it */
	if (ctx->qstride && ctx->empty_rows_nan[slot])
		for (int i : ctx->empty_rows)
			for (int k = 0; k < ctx->K; k++) q[(size_t)i * ctx->K + k] = -__builtin_nan("");
	return MCHIP_OK;
}

int mchip_empty_individuals(const mchip_context *ctx, int *first)
{
	MCHIP_ENTRY();
	if (!ctx) return 0;
	if (first) *first = ctx->first_empty;
	return (int)ctx->empty_rows.size();
}

static int fetch_scalars(mchip_context *ctx, int first, int count, double *out)
{
	HIPCHK(hipMemcpyAsync(ctx->h_pinned, ctx->d_scalars + first, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	for (int x = 0; x < count; x++) out[x] = ctx->h_pinned[x];
	return MCHIP_OK;
}

/* shared eta: eta[to] = normalise(sum_i S_ik), project (em_alg.c:604-648) */
static int finalize_shared_eta(mchip_context *ctx, int to, const int *stop = nullptr, double add = 0.0, int project = 1)
{
	hipLaunchKernelGGL(k_column_sums, dim3(ctx->K), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_sik, ctx->I, ctx->K, ctx->d_scalars + 8, stop);
	hipLaunchKernelGGL(k_normalize_row, dim3(1), dim3(64), 0, ctx->stream, ctx->d_scalars + 8, ctx->K, ctx->d_q[to], stop, add);
	if (ctx->do_projection && project) ctx->kt->project_q(1, ctx->K, ctx->d_q[to], ctx->eta_lb, stop, ctx->stream);
	HIPCHK(hipGetLastError());
	return MCHIP_OK;
}

/* mixture model: E step (mode 0, optionally followed by the M step) or logL_mixture (mode 1) */
static int run_mixture(mchip_context *ctx, int from, int to, int do_mstep, int mode, const int *stop = nullptr, int ll_slot = -1,
		       bool defer_ll = false)
{
	if (ll_slot < 0) ll_slot = mode ? 1 : 0;	/* d_scalars entry that receives the log likelihood */
	const size_t KT = (size_t)ctx->K * ctx->T;
	hipLaunchKernelGGL(k_logp, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_p[from], ctx->d_logp, KT, mode == 0, stop);
	mchip_pass_args a = pass_args(ctx, from);
	a.P = ctx->d_logp;
	a.stop = stop;
	prof_mark(ctx, mode == 0 ? MCHIP_KERN_ACCUM_Q : MCHIP_KERN_LOGLIK, true);
	ctx->kt->mix_gather(a, ctx->stream);
	prof_mark(ctx, mode == 0 ? MCHIP_KERN_ACCUM_Q : MCHIP_KERN_LOGLIK, false);
	const int nb = (ctx->I + MCHIP_BLOCK - 1) / MCHIP_BLOCK;
	ctx->kt->mix_finalize(ctx->I, ctx->n_lchunks, ctx->d_Spart, ctx->d_q[from], ctx->d_sik, ctx->d_llpart, mode, stop, ctx->stream);
	ctx->ll_parts = nb;
	if (!defer_ll)
		hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_llpart, nb, ctx->d_scalars + ll_slot, stop);
	if (do_mstep) {
		int rc = finalize_shared_eta(ctx, to, stop);		/* em_alg.c:916-962 */
		if (rc) return rc;
		mchip_pass_args b = pass_args(ctx, from);
		b.Q = ctx->d_sik;				/* vik rows */
		b.qstride = ctx->K;
		b.stop = stop;
		prof_mark(ctx, MCHIP_KERN_ACCUM_P, true);
		ctx->kt->mix_column(b, ctx->stream);
		prof_mark(ctx, MCHIP_KERN_ACCUM_P, false);
		launch_finalize_p(ctx, ctx->kt->col_slabs(b, 1), ctx->d_Apart, from, to, 0, ctx->p_lb, ctx->do_projection, stop);
	}
	HIPCHK(hipGetLastError());
	if (mode == 0) ctx->have_ll = 1;
	return MCHIP_OK;
}

/* defer_ll: the caller's k_stop_check sums the partial log likelihoods (ctx->ll_parts of them in d_llpart) itself */
static int run_estep(mchip_context *ctx, int from, int to, int do_mstep, const int *stop = nullptr, const int *skip_ind = nullptr,
		     bool defer_ll = false)
{
	if (!ctx->admixture) return run_mixture(ctx, from, to, do_mstep, 0, stop, -1, defer_ll);
	mchip_pass_args a = pass_args(ctx, from);
	a.stop = stop;
	a.skip_ind = skip_ind;
	if (do_mstep || !ctx->sparse) {
		prof_mark(ctx, MCHIP_KERN_ACCUM_P, true);
		if (do_mstep) ctx->kt->accum_p(a, ctx->stream); else ctx->kt->loglik(a, ctx->stream);
		prof_mark(ctx, MCHIP_KERN_ACCUM_P, false);
	}
	if (ctx->sparse && do_mstep && !stop && ctx->s_cache_slot == from) {
		/* the individual pass over these very parameters already ran (mchip_loglik_prefetch): its sums are in Spart,
		 * its log likelihood in d_scalars[2] */
		HIPCHK(hipMemcpyAsync(ctx->d_scalars, ctx->d_scalars + 2, sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
	} else {
		prof_mark(ctx, MCHIP_KERN_ACCUM_Q, true);
		ctx->kt->accum_q(a, ctx->stream);
		prof_mark(ctx, MCHIP_KERN_ACCUM_Q, false);
		ctx->ll_parts = ctx->kt->ind_ll_parts(a);
		if (!defer_ll)
			hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_llpart, ctx->ll_parts, ctx->d_scalars, stop);
	}
	ctx->s_cache_slot = -1;		/* Spart is consumed below; slot `to` is about to change */
	const int indiv = ctx->qstride != 0;
	const int s_slabs = ctx->kt->ind_slabs(a), n_slabs = ctx->kt->col_slabs(a, 0);	/* what the passes above left (cooperating waves: a quarter) */
	/* an M step with individual mixing proportions whose P side takes the tiled form: both finalisers in one launch */
	const int p_loci = finalize_p_loci(ctx->K, ctx->max_M);
	if (do_mstep && indiv && p_loci && !(n_slabs > 8 && !ctx->knob.no_slab_sum) && !ctx->knob.no_fused_finalize) {
		mchip_finalize_p_args p;
		p.L = ctx->L; p.K = ctx->K; p.T = ctx->T; p.loci_per_block = p_loci; p.n_slabs = n_slabs; p.weighted = 1;
		p.do_projection = ctx->do_projection; p.toff = ctx->d_toff; p.Apart = ctx->d_Apart; p.Pfrom = ctx->d_p[from];
		p.Pto = ctx->d_p[to]; p.add_lb = 0.0; p.lb = ctx->p_lb;
		ctx->kt->finalize_qp(ctx->I, s_slabs, ctx->d_Spart, ctx->d_q[from], ctx->qstride, ctx->d_q[to], ctx->d_sik,
				     ctx->do_projection, ctx->eta_lb, p, stop, ctx->stream);
		ctx->empty_rows_nan[to] = 1;
		HIPCHK(hipGetLastError());
		ctx->have_ll = 1;
		return MCHIP_OK;
	}
	/* k_finalize_q adds the slabs itself, eight threads per individual (no slab-sum launch in front of it: at config 2 that launch
	 * was 4 of the 137 us of a step, at config 3 two of the cycle's small kernels) */
	ctx->kt->finalize_q(ctx->I, ctx->K, s_slabs, ctx->d_Spart, ctx->d_q[from], ctx->qstride,
			    ctx->d_q[to], ctx->d_sik, do_mstep && indiv, 1, ctx->do_projection, ctx->eta_lb, stop, ctx->stream, 0.0);
	if (do_mstep && indiv) ctx->empty_rows_nan[to] = 1;
	if (do_mstep) {
		if (!indiv) {
			int rc = finalize_shared_eta(ctx, to, stop);
			if (rc) return rc;
		}
		if (n_slabs > 8 && !ctx->knob.no_slab_sum) {
			/* many N-side slabs: k_sum_slabs adds them at 5 TB/s (element- and slab-level parallelism, fully coalesced; the
			 * (l, k) threads of k_finalize_p reach 2.7 TB/s on the same bytes), k_finalize_p then reads one slab */
			const size_t n = (size_t)ctx->K * ctx->T;
			hipLaunchKernelGGL(k_sum_slabs, dim3(nblk(n, 32)), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_Apart, n_slabs, n, ctx->d_stage, stop);
			launch_finalize_p(ctx, 1, ctx->d_stage, from, to, 1, 0.0, ctx->do_projection, stop);
		} else {
			launch_finalize_p(ctx, n_slabs, ctx->d_Apart, from, to, 1, 0.0, ctx->do_projection, stop);
		}
	}
	HIPCHK(hipGetLastError());
	ctx->have_ll = 1;
	return MCHIP_OK;
}

int mchip_em_run(mchip_context *ctx, int slot, int n_steps, mchip_run_state *state)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	if (!state || n_steps < 1) return fail(ctx, MCHIP_ERR_INVALID, "em_run: bad arguments%s", nullptr);
	if (ctx->qstride) ctx->empty_rows_nan[slot] = 1;	/* (set here too: a replayed graph does not pass through the enqueueing code) */
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(ctx->d_run, state, sizeof *state, hipMemcpyHostToDevice, ctx->stream));
	const int *stop = &ctx->d_run->stopped;
	/* The step is a launch-bound chain of 6-7 kernels on small data sets: capture it once into a hipGraph and replay it
	 * (eager launches cost ~8 us of host time each here, graph nodes ~1.5 us of device time).  Event marks cannot be
	 * captured, so profiled runs stay eager. */
	bool use_graph = !ctx->profiling && n_steps >= 4 && !ctx->knob.no_graph;
	if (use_graph && !ctx->step_graph[slot]) {
		hipGraph_t graph = nullptr;
		if (MCHIP_WAIT(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal)) == hipSuccess) {
			rc = run_estep(ctx, slot, slot, 1, stop, nullptr, true);
			hipLaunchKernelGGL(k_stop_check, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_run, ctx->d_scalars, ctx->d_llpart, ctx->ll_parts);
			const hipError_t e = MCHIP_WAIT(hipStreamEndCapture(ctx->stream, &graph));
			if (rc || e != hipSuccess || !graph || MCHIP_WAIT(hipGraphInstantiate(&ctx->step_graph[slot], graph, nullptr, nullptr, 0)) != hipSuccess)
				ctx->step_graph[slot] = nullptr;
			if (graph) (void)MCHIP_WAIT(hipGraphDestroy(graph));
			(void)hipGetLastError();
		}
		if (!ctx->step_graph[slot]) use_graph = false;	/* capture unavailable: eager path below */
	}
	for (int s = 0; s < n_steps; s++) {
		if (use_graph) {
			HIPCHK(hipGraphLaunch(ctx->step_graph[slot], ctx->stream));
			continue;
		}
		if ((rc = run_estep(ctx, slot, slot, 1, stop, nullptr, true))) return rc;
		hipLaunchKernelGGL(k_stop_check, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_run, ctx->d_scalars, ctx->d_llpart, ctx->ll_parts);
	}
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(state, ctx->d_run, sizeof *state, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

int mchip_em_step(mchip_context *ctx, int from, int to, double *loglik)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, from);
	if (rc) return rc;
	if ((rc = check_slot(ctx, to))) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	if ((rc = run_estep(ctx, from, to, 1))) return rc;
	if (ctx->qstride) ctx->empty_rows_nan[to] = 1;
	if (loglik) return fetch_scalars(ctx, 0, 1, loglik);
	return MCHIP_OK;
}

int mchip_last_loglik(mchip_context *ctx, double *loglik)
{
	MCHIP_ENTRY();
	if (!ctx || !loglik) return MCHIP_ERR_INVALID;
	if (!ctx->have_ll) return fail(ctx, MCHIP_ERR_STATE, "no log likelihood computed yet%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	return fetch_scalars(ctx, 0, 1, loglik);
}

int mchip_e_step(mchip_context *ctx, int slot, double *loglik)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	if ((rc = run_estep(ctx, slot, slot, 0))) return rc;
	if (loglik) return fetch_scalars(ctx, 0, 1, loglik);
	return MCHIP_OK;
}

int mchip_loglik(mchip_context *ctx, int slot, double *loglik)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	if (!ctx->admixture) {
		if ((rc = run_mixture(ctx, slot, slot, 0, 1))) return rc;
		if (loglik) return fetch_scalars(ctx, 1, 1, loglik);
		return MCHIP_OK;
	}
	mchip_pass_args a = pass_args(ctx, slot);
	prof_mark(ctx, MCHIP_KERN_LOGLIK, true);
	ctx->kt->loglik(a, ctx->stream);
	prof_mark(ctx, MCHIP_KERN_LOGLIK, false);
	hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_llpart, ctx->kt->ind_ll_parts(a), ctx->d_scalars + 1);
	HIPCHK(hipGetLastError());
	if (loglik) return fetch_scalars(ctx, 1, 1, loglik);
	return MCHIP_OK;
}

int mchip_loglik_prefetch(mchip_context *ctx, int slot, double *loglik)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	if (!ctx->admixture || !ctx->sparse) return mchip_loglik(ctx, slot, loglik);	/* nothing to share on those paths */
	HIPCHK(hipSetDevice(ctx->device));
	mchip_pass_args a = pass_args(ctx, slot);
	prof_mark(ctx, MCHIP_KERN_ACCUM_Q, true);
	ctx->kt->accum_q(a, ctx->stream);		/* S-side sums -> Spart, logL partials -> llpart */
	prof_mark(ctx, MCHIP_KERN_ACCUM_Q, false);
	hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_llpart, ctx->kt->ind_ll_parts(a), ctx->d_scalars + 2);
	HIPCHK(hipGetLastError());
	ctx->s_cache_slot = slot;
	if (loglik) return fetch_scalars(ctx, 2, 1, loglik);
	return MCHIP_OK;
}

/* the hard-partition counts are in Spart ([n_lchunks][I][K]) and in n_aslabs slabs of Apart: normalise (and project) into slot `to` */
static int partition_finalize(mchip_context *ctx, int to, int counts, int n_aslabs)
{
	int rc;
	const int indiv = ctx->qstride != 0;
	const int project = counts ? 0 : ctx->do_projection;
	const double add = counts ? 1.0 : 0.0;
	ctx->empty_rows_nan[to] = counts ? 0 : (ctx->qstride ? 1 : 0);	/* an M step of the hard partition: 0 / 0 for such a row (rnd_init.c:356); with counts
									 * every count starts at 1 (rnd_init.c:624): 1 / K, as the device holds it */
	ctx->kt->finalize_q(ctx->I, ctx->K, ctx->n_lchunks, ctx->d_Spart, ctx->d_q[to], ctx->qstride,
			    ctx->d_q[to], ctx->d_sik, indiv, 0, project, ctx->eta_lb, nullptr, ctx->stream, add);
	if (!indiv && (rc = finalize_shared_eta(ctx, to, nullptr, add, !counts))) return rc;
	hipLaunchKernelGGL(k_finalize_p, dim3(nblk((size_t)ctx->L * ctx->K)), dim3(MCHIP_BLOCK), 0, ctx->stream,
			   ctx->L, ctx->K, ctx->T, ctx->d_toff, n_aslabs, ctx->d_Apart, ctx->d_p[to], ctx->d_p[to],
			   0, add, project, ctx->p_lb, ctx->d_flags);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

/* relayout of a partition held on the device in stream order, then the hard-partition M step into slot `to`
 * (counts = 0: random_initialize_admixture, rnd_init.c:349-357), or initialize_parameters_admixture (rnd_init.c:603-705)
 * from that partition (counts = 1: one count per copy on top of 1, normalised, never projected) */
static int partition_mstep(mchip_context *ctx, const uint8_t *d_raw, int to, int counts = 0)
{
	if (!ctx->d_asA) HIPCHK(hipMalloc((void **)&ctx->d_asA, ctx->geno_bytes_A));
	if (!ctx->d_asS) HIPCHK(hipMalloc((void **)&ctx->d_asS, ctx->geno_bytes_S));
	int *d_bad = bad_flag(ctx);
	HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
	if (launch_relayout(ctx->stream, d_raw, ctx->I, ctx->L, ctx->ploidy, nullptr, ctx->K, ctx->d_asA, ctx->d_asS, d_bad))
		return fail(ctx, MCHIP_ERR_UNSUPPORTED, "data set too large for the layout kernel%s", nullptr);
	HIPCHK(hipGetLastError());
	int bad = 0;
	HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	if (bad) return fail(ctx, MCHIP_ERR_INVALID, "partition assignment >= K%s", nullptr);

	mchip_pass_args a = pass_args(ctx, to);
	if (ctx->init_geno_set) {	/* bootstrap fits: the partition indicators come from the observed haplotypes (rnd_init.c:471) */
		a.gtA = ctx->d_initA;
		a.gtS = ctx->d_initS;
	}
	a.part_counts = counts;
	ctx->kt->part_p(a, ctx->stream);
	ctx->kt->part_q(a, ctx->stream);
	return partition_finalize(ctx, to, counts, ctx->n_ichunks);
}

static int check_partition_call(mchip_context *ctx, const void *arg, int to)
{
	int rc = check_slot(ctx, to);
	if (rc) return rc;
	ctx->s_cache_slot = -1;
	if (!arg) return fail(ctx, MCHIP_ERR_INVALID, "null pointer%s", nullptr);
	if (!ctx->admixture) return fail(ctx, MCHIP_ERR_STATE, "allele partitions initialise the admixture model only%s", nullptr);
	return MCHIP_OK;
}

int mchip_mstep_from_partition(mchip_context *ctx, const uint8_t *assign, int to)
{
	MCHIP_ENTRY();
	int rc = check_partition_call(ctx, assign, to);
	if (rc) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	const size_t raw_bytes = (size_t)ctx->I * ctx->L * ctx->ploidy;
	scoped_dev<uint8_t> d_raw;
	HIPCHK(d_raw.alloc(raw_bytes));
	HIPCHK(hipMemcpyAsync(d_raw, assign, raw_bytes, hipMemcpyHostToDevice, ctx->stream));
	return partition_mstep(ctx, d_raw, to);
}

/* (a * b) mod (x^31 - x^28 - 1) over Z/2^32 */
static void rng_polymul(const uint32_t *a, const uint32_t *b, uint32_t *out)
{
	uint32_t t[2 * RNG_LAG - 1] = {0};
	for (int i = 0; i < RNG_LAG; i++)
		for (int j = 0; j < RNG_LAG; j++) t[i + j] += a[i] * b[j];
	for (int d = 2 * RNG_LAG - 2; d >= RNG_LAG; d--) {
		t[d - 3] += t[d];
		t[d - RNG_LAG] += t[d];
	}
	memcpy(out, t, RNG_LAG * sizeof(uint32_t));
}

/* tables of x^(lo*CHUNK), lo < 256 (stored [31][256]) and x^(hi*256*CHUNK), hi < n_hi (stored [n_hi][31]) */
static int rng_jump_tables(mchip_context *ctx, size_t n_hi)
{
	if (ctx->d_jump_lo && ctx->n_jump_hi >= n_hi) return MCHIP_OK;
	dfree(ctx->d_jump_hi);
	dfree(ctx->d_jump_lo);
	ctx->n_jump_hi = 0;
	uint32_t step[RNG_LAG] = {0}, sq[RNG_LAG] = {0}, cur[RNG_LAG];
	step[0] = 1;	/* x^0 */
	sq[1] = 1;	/* x^1 */
	for (unsigned e = RNG_CHUNK; e; e >>= 1) {	/* step = x^CHUNK */
		if (e & 1) rng_polymul(step, sq, step);
		rng_polymul(sq, sq, sq);
	}
	std::vector<uint32_t> lo((size_t)RNG_LAG * 256), hi(n_hi * RNG_LAG);
	memset(cur, 0, sizeof cur);
	cur[0] = 1;
	for (int x = 0; x < 256; x++) {
		for (int j = 0; j < RNG_LAG; j++) lo[(size_t)j * 256 + x] = cur[j];
		rng_polymul(cur, step, cur);
	}
	memcpy(step, cur, sizeof step);	/* x^(256*CHUNK) */
	memset(cur, 0, sizeof cur);
	cur[0] = 1;
	for (size_t x = 0; x < n_hi; x++) {
		memcpy(&hi[x * RNG_LAG], cur, sizeof cur);
		rng_polymul(cur, step, cur);
	}
	HIPCHK(hipMalloc((void **)&ctx->d_jump_lo, lo.size() * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&ctx->d_jump_hi, hi.size() * sizeof(uint32_t)));
	HIPCHK(hipMemcpy(ctx->d_jump_lo, lo.data(), lo.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	HIPCHK(hipMemcpy(ctx->d_jump_hi, hi.data(), hi.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
	ctx->n_jump_hi = n_hi;
	return MCHIP_OK;
}

static int rng_stream_setup(mchip_context *ctx, const uint32_t *window, size_t n_draws, rng_window *base, size_t *n_chunks, size_t *n_blocks)
{
	*n_chunks = (n_draws + RNG_CHUNK - 1) / RNG_CHUNK;
	*n_blocks = (*n_chunks + 255) / 256;
	for (int t = 0; t < RNG_LAG; t++) base->s[t] = window[t];
	for (int t = 0; t < RNG_LAG - 1; t++) base->s[RNG_LAG + t] = base->s[t] + base->s[RNG_LAG - 3 + t];
	return rng_jump_tables(ctx, *n_blocks);
}

/* x^e mod (x^31 - x^28 - 1) */
static void rng_xpow(uint64_t e, uint32_t *out)
{
	uint32_t acc[RNG_LAG] = {0}, sq[RNG_LAG] = {0};
	acc[0] = 1;
	sq[1] = 1;
	for (; e; e >>= 1) {
		if (e & 1) rng_polymul(acc, sq, acc);
		rng_polymul(sq, sq, sq);
	}
	memcpy(out, acc, sizeof acc);
}

/* d_out[j * sj + n * sn] = coefficient j of x^(A*n), n < count */
static int rng_lattice_table(mchip_context *ctx, uint64_t A, unsigned count, size_t sj, size_t sn, uint32_t *d_out)
{
	uint32_t pw[32 * RNG_LAG];
	rng_xpow(A, pw);
	for (int b = 1; b < 32; b++) rng_polymul(pw + (b - 1) * RNG_LAG, pw + (b - 1) * RNG_LAG, pw + b * RNG_LAG);
	scoped_dev<uint32_t> d_pw;
	HIPCHK(d_pw.alloc(32 * RNG_LAG));
	HIPCHK(hipMemcpyAsync(d_pw, pw, sizeof pw, hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(k_jump_powers, dim3(nblk(count)), dim3(256), 0, ctx->stream, d_pw.p, count, sj, sn, d_out);
	HIPCHK(hipGetLastError());
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

/* jump polynomials of a tiled generator (see mchip_context::lat): thread (i, r) starts A*i + B*r draws into the stream */
static int lattice_tables(mchip_context *ctx, int which, uint64_t A, unsigned nA, uint64_t B, unsigned nB)
{
	mchip_context::lattice &t = ctx->lat[which];
	if (t.d_i && t.A == A && t.nA == nA && t.B == B && t.nB == nB) return MCHIP_OK;
	dfree(t.d_i);
	dfree(t.d_r);
	t.nA = 0;
	HIPCHK(hipMalloc((void **)&t.d_i, (size_t)RNG_LAG * nA * sizeof(uint32_t)));
	HIPCHK(hipMalloc((void **)&t.d_r, (size_t)RNG_LAG * nB * sizeof(uint32_t)));
	int rc = rng_lattice_table(ctx, A, nA, (size_t)nA, 1, t.d_i);
	if (!rc) rc = rng_lattice_table(ctx, B, nB, 1, RNG_LAG, t.d_r);
	if (rc) return rc;
	t.A = A; t.nA = nA; t.B = B; t.nB = nB;
	return MCHIP_OK;
}

/* v % K through floor(v / K) = (v * magic) >> (31 + l), l = ceil(log2 K), magic = ceil(2^(31+l) / K) < 2^32: exact for every
 * v < 2^31 (division by an invariant integer); the kernels take the high word, so shift = l - 1.  K >= 2. */
static void mod_k_magic(int K, uint32_t *magic, uint32_t *shift)
{
	uint32_t l = 0;
	while ((1u << l) < (uint32_t)K) l++;
	const uint64_t pw = (uint64_t)1 << (31 + l);
	*magic = (uint32_t)((pw + (uint64_t)K - 1) / (uint64_t)K);
	*shift = l - 1;
}

/* the tiled form of the random partition + first M step (k_partition_tile); returns -1 when the shape does not fit it */
static int rand_partition_tiled(mchip_context *ctx, const uint32_t *window, int to)
{
	const int K = ctx->K, KP = (K + 1) & ~1, pl = ctx->ploidy;
	if (pl > 8 || (size_t)ctx->lchunk * pl > 65535 || ctx->knob.part_no_tile) return -1;
	/* N-side counters of a tile: tile * max_M * KP / 2 words, about 16 KiB, a multiple of 8 loci */
	int tile = (int)((16384 / ((size_t)ctx->max_M * KP * 2)) & ~(size_t)7);
	if (tile < 8) tile = 8;
	if (tile > ctx->lchunk) tile = ctx->lchunk;
	const size_t lds = ((size_t)RNG_LAG * 256 + (size_t)(KP / 2) * 256 + (size_t)tile * ctx->max_M * (KP / 2)) * sizeof(uint32_t);
	if (lds > 64 * 1024) return -1;
	int rc;
	rng_window base;
	for (int t = 0; t < RNG_LAG; t++) base.s[t] = window[t];
	for (int t = 0; t < RNG_LAG - 1; t++) base.s[RNG_LAG + t] = base.s[t] + base.s[RNG_LAG - 3 + t];
	if ((rc = lattice_tables(ctx, 1, (uint64_t)ctx->L * pl, (unsigned)ctx->I, (uint64_t)ctx->lchunk * pl, (unsigned)ctx->n_lchunks))) return rc;
	const size_t slab_words = (size_t)ctx->T * (KP / 2), n_slabs = (size_t)(ctx->I + 255) / 256;
	if (ctx->part_slab_bytes < n_slabs * slab_words * sizeof(uint32_t)) {
		dfree(ctx->d_part_slabs);
		ctx->part_slab_bytes = 0;
		HIPCHK(hipMalloc((void **)&ctx->d_part_slabs, n_slabs * slab_words * sizeof(uint32_t)));
		ctx->part_slab_bytes = n_slabs * slab_words * sizeof(uint32_t);
	}
	uint32_t magic = 0, shift = 0;
	if (K > 1) mod_k_magic(K, &magic, &shift);
	const uint8_t *gtS = ctx->init_geno_set ? ctx->d_initS : ctx->d_gtS;	/* bootstrap fits: the observed haplotypes (rnd_init.c:471) */
	switch (pl) {
	case 1: launch_partition_tile<1>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	case 2: launch_partition_tile<2>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	case 3: launch_partition_tile<3>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	case 4: launch_partition_tile<4>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	case 5: launch_partition_tile<5>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	case 6: launch_partition_tile<6>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	case 7: launch_partition_tile<7>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	default: launch_partition_tile<8>(ctx, base, magic, shift, tile, lds, gtS, slab_words); break;
	}
	HIPCHK(hipGetLastError());
	hipLaunchKernelGGL(k_partition_finish, dim3(nblk((size_t)ctx->T * K)), dim3(256), 0, ctx->stream, ctx->d_part_slabs, slab_words,
			   (int)n_slabs, ctx->T, K, ctx->d_Apart);
	HIPCHK(hipGetLastError());
	return partition_finalize(ctx, to, 0, 1);
}

int mchip_mstep_from_rand_partition(mchip_context *ctx, const uint32_t *window, int to)
{
	MCHIP_ENTRY();
	int rc = check_partition_call(ctx, window, to);
	if (rc) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	if ((rc = rand_partition_tiled(ctx, window, to)) >= 0) return rc;
	const size_t n = (size_t)ctx->I * ctx->L * ctx->ploidy;
	rng_window base;
	size_t n_chunks, n_blocks;
	if ((rc = rng_stream_setup(ctx, window, n, &base, &n_chunks, &n_blocks))) return rc;
	if ((rc = stream_buffer(ctx))) return rc;
	if (ctx->K == 1) {
		HIPCHK(hipMemsetAsync(ctx->d_draw, 0, n, ctx->stream));	/* rand() % 1 */
	} else {
		uint32_t magic, shift;
		mod_k_magic(ctx->K, &magic, &shift);
		hipLaunchKernelGGL(k_draw_partition, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, base, ctx->d_jump_hi,
				   ctx->d_jump_lo, n_chunks, (uint32_t)ctx->K, magic, shift, (uint32_t *)ctx->d_draw);
		HIPCHK(hipGetLastError());
	}
	return partition_mstep(ctx, ctx->d_draw, to);
}

int mchip_init_from_allele_centers(mchip_context *ctx, const uint8_t *centers, const uint64_t *draw_offset, const uint32_t *window,
				   uint64_t n_draws, int to)
{
	MCHIP_ENTRY();
	int rc = check_partition_call(ctx, centers, to);
	if (rc) return rc;
	if (!draw_offset || !window) return fail(ctx, MCHIP_ERR_INVALID, "null pointer%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	/* rand() % K for the candidate's whole span of the stream (center draws included: the host knows which draws are whose).
	 * No allocation per candidate: the span, centers and offsets live in the context (grown when a candidate needs more), the
	 * assignment goes to the stream-order scratch buffer -- a hipFree would synchronise the whole device and stall the other
	 * --streams workers, n_rand_em_init times per initialisation */
	if ((rc = stream_buffer(ctx))) return rc;
	uint8_t *d_raw = ctx->d_draw;
	rng_window base;
	size_t n_chunks = 0, n_blocks = 0;
	auto grow = [&](void **p, size_t *have, size_t want) -> hipError_t {
		if (*have >= want) return hipSuccess;
		if (*p) { (void)MCHIP_WAIT(hipStreamSynchronize(ctx->stream)); (void)MCHIP_WAIT(hipFree(*p)); *p = nullptr; *have = 0; }
		const hipError_t e = hipMalloc(p, want);
		if (e == hipSuccess) *have = want;
		return e;
	};
	if (n_draws) {
		if ((rc = rng_stream_setup(ctx, window, (size_t)n_draws, &base, &n_chunks, &n_blocks))) return rc;
		HIPCHK(grow((void **)&ctx->d_cand_span, &ctx->cand_span_bytes, n_chunks * RNG_CHUNK + n_chunks * RNG_CHUNK / 8));
		if (ctx->K == 1) {
			HIPCHK(hipMemsetAsync(ctx->d_cand_span, 0, n_chunks * RNG_CHUNK, ctx->stream));
		} else {
			uint32_t magic, shift;
			mod_k_magic(ctx->K, &magic, &shift);
			hipLaunchKernelGGL(k_draw_partition, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, base, ctx->d_jump_hi,
					   ctx->d_jump_lo, n_chunks, (uint32_t)ctx->K, magic, shift, (uint32_t *)ctx->d_cand_span);
		}
	} else {
		HIPCHK(grow((void **)&ctx->d_cand_span, &ctx->cand_span_bytes, 16));	/* no copy draws: never read */
	}
	/* every offset must leave room for the locus's copies inside the span: a locus reads at most I*ploidy draws */
	for (int l = 0; l < ctx->L; l++)
		if (draw_offset[l] > n_draws) return fail(ctx, MCHIP_ERR_INVALID, "draw offset beyond the stream span%s", nullptr);
	HIPCHK(grow((void **)&ctx->d_cand_centers, &ctx->cand_center_bytes, (size_t)ctx->L * ctx->K));
	{
		size_t have = ctx->cand_loci * sizeof(unsigned long long);
		HIPCHK(grow((void **)&ctx->d_cand_off, &have, (size_t)ctx->L * sizeof(unsigned long long)));
		ctx->cand_loci = have / sizeof(unsigned long long);
	}
	uint8_t *d_span = ctx->d_cand_span, *d_cen = ctx->d_cand_centers;
	unsigned long long *d_off = ctx->d_cand_off;
	HIPCHK(hipMemcpyAsync(d_cen, centers, (size_t)ctx->L * ctx->K, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(d_off, draw_offset, sizeof(uint64_t) * (size_t)ctx->L, hipMemcpyHostToDevice, ctx->stream));
	int *d_bad = bad_flag(ctx);
	HIPCHK(hipMemsetAsync(d_bad, 0, sizeof(int), ctx->stream));
	hipLaunchKernelGGL(k_assign_by_centers, dim3(nblk((size_t)ctx->L)), dim3(256), 0, ctx->stream,
			   ctx->init_geno_set ? ctx->d_initA : ctx->d_gtA, ctx->I, ctx->L, ctx->ploidy, ctx->K, d_cen, d_off, d_span,
			   (unsigned long long)n_draws, d_raw, d_bad);
	HIPCHK(hipGetLastError());
	int bad = 0;
	HIPCHK(hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	if (bad) return fail(ctx, MCHIP_ERR_INVALID, "allele-center draws run past the stream span: offsets do not match the genotype held%s", nullptr);
	rc = partition_mstep(ctx, d_raw, to, 1);
	(void)MCHIP_WAIT(hipStreamSynchronize(ctx->stream));	/* centers / draw_offset are the caller's: uploaded by now */
	return rc;
}

int mchip_copy_slot(mchip_context *ctx, int to, int from)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, to);
	if (rc || (rc = check_slot(ctx, from))) return rc;
	if (to == from) return MCHIP_OK;
	ctx->s_cache_slot = -1;
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(ctx->d_p[to], ctx->d_p[from], (size_t)ctx->K * ctx->T * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(ctx->d_q[to], ctx->d_q[from], (size_t)ctx->nq * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
	ctx->empty_rows_nan[to] = ctx->empty_rows_nan[from];
	return MCHIP_OK;
}

int mchip_simulate_genotypes(mchip_context *ctx, int I, int L, int ploidy, const int32_t *ua, const uint32_t *window,
			     int K, int eta_constrained, const double *q, const double *p)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!window || !q || !p || K < 1) return fail(ctx, MCHIP_ERR_INVALID, "simulate_genotypes: null pointer or K < 1%s", nullptr);
	/* a replicate of the same observed data: the observed haplotypes installed by mchip_set_init_genotypes stay in force
	 * (the reference's dat->IL stays in place across parametric_bootstrap calls, bootstrap.c:35-41) */
	int rc = set_shape(ctx, I, L, ploidy, ua, 1);
	if (rc) return rc;
	ctx->first_empty = -1;		/* every copy of a generated data set is drawn */
	ctx->empty_rows.clear();
	const size_t n_copies = (size_t)I * L * ploidy;
	rng_window base;
	size_t n_chunks, n_blocks;
	if ((rc = rng_stream_setup(ctx, window, 2 * n_copies, &base, &n_chunks, &n_blocks))) return rc;
	const size_t n_qrows = eta_constrained ? 1 : (size_t)I, nq = n_qrows * K, np = (size_t)K * ctx->T;
	const int fast = K <= SIM_QW && ctx->max_M <= SIM_PW;
	const size_t ncq = n_qrows * (fast ? SIM_QW : K), ncp = fast ? (size_t)K * L * SIM_PW : np;
	scoped_dev<double> d_q, d_p;
	scoped_dev<uint32_t> d_cq, d_cp;
	HIPCHK(d_q.alloc(nq));
	HIPCHK(d_p.alloc(np));
	HIPCHK(d_cq.alloc(ncq));
	HIPCHK(d_cp.alloc(ncp));
	HIPCHK(hipMemcpyAsync(d_q, q, nq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipMemcpyAsync(d_p, p, np * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	{
		const size_t n = n_qrows > (size_t)K * L ? n_qrows : (size_t)K * L;
		hipLaunchKernelGGL(k_walk_tables, dim3(nblk(n)), dim3(256), 0, ctx->stream, (int)n_qrows, K, L, ctx->T, ctx->d_toff, d_q.p, d_p.p,
				   fast, d_cq.p, d_cp.p);
	}
	if (fast && ploidy <= 8 && !ctx->knob.sim_no_tile) {
		/* tile: as many loci as keep the staged thresholds near 16 KiB (three workgroups per compute unit), a multiple of 8 */
		int tile = (16384 / (K * 12)) & ~7;
		if (tile > 512) tile = 512;
		if (tile > ((L + 7) & ~7)) tile = (L + 7) & ~7;
		if ((rc = lattice_tables(ctx, 0, 2ull * (uint64_t)L * ploidy, (unsigned)I, 2ull * (uint64_t)tile * ploidy, (unsigned)((L + tile - 1) / tile))))
			return rc;
		switch (ploidy) {
		case 1: launch_simulate_tile<1>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		case 2: launch_simulate_tile<2>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		case 3: launch_simulate_tile<3>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		case 4: launch_simulate_tile<4>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		case 5: launch_simulate_tile<5>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		case 6: launch_simulate_tile<6>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		case 7: launch_simulate_tile<7>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		default: launch_simulate_tile<8>(ctx, base, K, tile, d_cq.p, eta_constrained ? 0 : 1, d_cp.p); break;
		}
		HIPCHK(hipGetLastError());
		rc = install_layouts(ctx, 0);
		(void)MCHIP_WAIT(hipStreamSynchronize(ctx->stream));	/* the temporaries go out of scope */
		return rc;
	}
	if ((rc = stream_buffer(ctx))) return rc;	/* n_chunks * RNG_CHUNK / 2 <= its size */
	uint8_t *d_raw = ctx->d_draw;
	if (fast)
		hipLaunchKernelGGL(k_simulate_admixture<true>, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, base, ctx->d_jump_hi,
				   ctx->d_jump_lo, n_chunks, I, L, ploidy, K, ctx->T, ctx->d_toff, d_cq.p, eta_constrained ? 0 : 1, d_cp.p,
				   (uint32_t *)d_raw);
	else
		hipLaunchKernelGGL(k_simulate_admixture<false>, dim3((unsigned)n_blocks), dim3(256), 0, ctx->stream, base, ctx->d_jump_hi,
				   ctx->d_jump_lo, n_chunks, I, L, ploidy, K, ctx->T, ctx->d_toff, d_cq.p, eta_constrained ? 0 : 1, d_cp.p,
				   (uint32_t *)d_raw);
	HIPCHK(hipGetLastError());
	rc = install_raw(ctx, d_raw);
	(void)MCHIP_WAIT(hipStreamSynchronize(ctx->stream));	/* the temporaries go out of scope */
	return rc;
}

int mchip_get_expected_counts(mchip_context *ctx, double *sik)
{
	MCHIP_ENTRY();
	if (!ctx || !sik) return MCHIP_ERR_INVALID;
	if (!ctx->K) return fail(ctx, MCHIP_ERR_STATE, "no model set%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(sik, ctx->d_sik, (size_t)ctx->I * ctx->K * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

/* ---- acceleration ---- */
static int check_secant(mchip_context *ctx, int j)
{
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!ctx->K) return fail(ctx, MCHIP_ERR_STATE, "no model set%s", nullptr);
	if (j < 0 || j >= ctx->nsec) return fail(ctx, MCHIP_ERR_INVALID, "secant index out of range%s", nullptr);
	return MCHIP_OK;
}

/* the secant buffers as model state (mod->u_pklm / v_pklm / u_etaik / v_etaik, multiclust.h:284-293): written and read whole, in
 * the parameter slots' own flat order, so that a quasi-Newton run with q > 1 can be resumed from a recorded state */
int mchip_set_secant(mchip_context *ctx, int which, int j, const double *p_part, const double *q_part)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, j);
	if (rc) return rc;
	if (!p_part || !q_part || (which != 0 && which != 1)) return fail(ctx, MCHIP_ERR_INVALID, "set_secant: bad arguments%s", nullptr);
	const size_t KT = (size_t)ctx->K * ctx->T;
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(ctx->d_stage, p_part, KT * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	hipLaunchKernelGGL(k_transpose_kt_to_tk, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_stage, which ? ctx->d_vp[j] : ctx->d_up[j], ctx->K, ctx->T);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(which ? ctx->d_vq[j] : ctx->d_uq[j], q_part, (size_t)ctx->nq * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}
int mchip_get_secant(mchip_context *ctx, int which, int j, double *p_part, double *q_part)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, j);
	if (rc) return rc;
	if (!p_part || !q_part || (which != 0 && which != 1)) return fail(ctx, MCHIP_ERR_INVALID, "get_secant: bad arguments%s", nullptr);
	const size_t KT = (size_t)ctx->K * ctx->T;
	HIPCHK(hipSetDevice(ctx->device));
	hipLaunchKernelGGL(k_transpose_tk_to_kt, dim3(nblk(KT)), dim3(256), 0, ctx->stream, which ? ctx->d_vp[j] : ctx->d_up[j], ctx->d_stage, ctx->K, ctx->T);
	HIPCHK(hipGetLastError());
	HIPCHK(hipMemcpyAsync(p_part, ctx->d_stage, KT * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipMemcpyAsync(q_part, which ? ctx->d_vq[j] : ctx->d_uq[j], (size_t)ctx->nq * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	return MCHIP_OK;
}

int mchip_secant(mchip_context *ctx, int which, int j, int to, int from)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, j);
	if (rc) return rc;
	if ((rc = check_slot(ctx, to)) || (rc = check_slot(ctx, from))) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	const size_t KT = (size_t)ctx->K * ctx->T;
	double *dp = which ? ctx->d_vp[j] : ctx->d_up[j];
	double *dq = which ? ctx->d_vq[j] : ctx->d_uq[j];
	hipLaunchKernelGGL(k_diff, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_p[to], ctx->d_p[from], dp, KT);
	hipLaunchKernelGGL(k_diff, dim3(nblk(ctx->nq)), dim3(256), 0, ctx->stream, ctx->d_q[to], ctx->d_q[from], dq, (size_t)ctx->nq);
	HIPCHK(hipGetLastError());
	return MCHIP_OK;
}

/* the dot products' eta parts land in d_scalars[16 + x], the p parts in d_scalars[20 + x] */
static void dots_enqueue(mchip_context *ctx, const double *uq, const double *vq, const double *u2q,
			 const double *up, const double *vp, const double *u2p, int mode, int nout, const int *stop = nullptr)
{
	const size_t KT = (size_t)ctx->K * ctx->T;
	const int gq = (int)((ctx->nq + 4095) / 4096) > 512 ? 512 : (int)((ctx->nq + 4095) / 4096);
	const int gp = (int)((KT + 4095) / 4096) > 512 ? 512 : (int)((KT + 4095) / 4096);
	double *part_q = ctx->d_redpart, *part_p = ctx->d_redpart + 3 * 512;
	hipLaunchKernelGGL(k_dots, dim3(gq), dim3(MCHIP_BLOCK), 0, ctx->stream, uq, vq, u2q, (size_t)ctx->nq, mode, part_q, stop);
	hipLaunchKernelGGL(k_dots, dim3(gp), dim3(MCHIP_BLOCK), 0, ctx->stream, up, vp, u2p, KT, mode, part_p, stop);
	hipLaunchKernelGGL(k_reduce_dots, dim3(2 * nout), dim3(MCHIP_BLOCK), 0, ctx->stream, part_q, gq, part_p, gp, nout, ctx->d_scalars, stop);
}

static int dots_common(mchip_context *ctx, const double *uq, const double *vq, const double *u2q,
		       const double *up, const double *vp, const double *u2p, int mode, int nout, double *out)
{
	dots_enqueue(ctx, uq, vq, u2q, up, vp, u2p, mode, nout);
	HIPCHK(hipGetLastError());
	double tmp[8];
	int rc = fetch_scalars(ctx, 16, 8, tmp);
	if (rc) return rc;
	for (int x = 0; x < nout; x++) out[x] = tmp[x] + tmp[4 + x];	/* eta terms first, then p (accel_em.c:143-184) */
	return MCHIP_OK;
}

int mchip_step_dots(mchip_context *ctx, int j, double *out3)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, j);
	if (rc) return rc;
	if (!out3) return MCHIP_ERR_INVALID;
	HIPCHK(hipSetDevice(ctx->device));
	return dots_common(ctx, ctx->d_uq[j], ctx->d_vq[j], nullptr, ctx->d_up[j], ctx->d_vp[j], nullptr, 0, 3, out3);
}

int mchip_secant_dots(mchip_context *ctx, int j1, int j2, double *out2)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, j1);
	if (rc) return rc;
	if ((rc = check_secant(ctx, j2))) return rc;
	if (!out2) return MCHIP_ERR_INVALID;
	HIPCHK(hipSetDevice(ctx->device));
	return dots_common(ctx, ctx->d_uq[j1], ctx->d_vq[j2], ctx->d_uq[j2], ctx->d_up[j1], ctx->d_vp[j2], ctx->d_up[j2], 1, 2, out2);
}

static int project_slot(mchip_context *ctx, int to, const int *stop = nullptr)
{
	if (!ctx->do_projection) return MCHIP_OK;
	hipLaunchKernelGGL(k_project_p, dim3(nblk((size_t)ctx->L * ctx->K)), dim3(MCHIP_BLOCK), 0, ctx->stream,
			   ctx->L, ctx->K, ctx->d_toff, ctx->d_p[to], ctx->p_lb, ctx->d_flags, stop);
	ctx->kt->project_q(ctx->qstride ? ctx->I : 1, ctx->K, ctx->d_q[to], ctx->eta_lb, stop, ctx->stream);
	HIPCHK(hipGetLastError());
	return MCHIP_OK;
}

int mchip_accel_update(mchip_context *ctx, int to, int base, int j, double s, int qn_form)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, j);
	if (rc) return rc;
	ctx->s_cache_slot = -1;
	if ((rc = check_slot(ctx, to)) || (rc = check_slot(ctx, base))) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	const size_t KT = (size_t)ctx->K * ctx->T;
	hipLaunchKernelGGL(k_accel_update, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_p[base], ctx->d_up[j], ctx->d_vp[j], ctx->d_p[to], KT, s, qn_form);
	hipLaunchKernelGGL(k_accel_update, dim3(nblk(ctx->nq)), dim3(256), 0, ctx->stream, ctx->d_q[base], ctx->d_uq[j], ctx->d_vq[j], ctx->d_q[to], (size_t)ctx->nq, s, qn_form);
	HIPCHK(hipGetLastError());
	ctx->empty_rows_nan[to] = ctx->empty_rows_nan[base];	/* x + (anything of NaN secants) is NaN in the reference */
	return project_slot(ctx, to);
}

/* One accelerated cycle = accelerated_em_step (accel_em.c:35-114) with one secant pair and no back-tracking, enqueued
 * without a host round trip: slot A holds the cycle's starting iterate, B = A+1 its first EM iterate and later the
 * extrapolated point, C = A+2 the second EM iterate.  Every decision the reference takes on the host is taken by a
 * one-thread kernel (stop rule twice, step size, accept test); the cycle's outcome is copied back to A so that the next
 * cycle has the same slot roles, which is what lets one captured graph serve every cycle. */
static int accel_cycle_enqueue(mchip_context *ctx, int A, int scheme)
{
	const int B = (A + 1) % 3, C = (A + 2) % 3;
	const int *stop = &ctx->d_run->stopped;
	int *cyc = ctx->d_cyc;
	const size_t KT = (size_t)ctx->K * ctx->T, nq = (size_t)ctx->nq;
	int rc;
	/* em_2_steps (em_alg.c:1072-1211): E(A) M(->B) stop, u = B - A; E(B) M(->C) stop, v = C - B */
	if ((rc = run_estep(ctx, A, B, 1, stop, cyc + 1, true))) return rc;
	hipLaunchKernelGGL(k_stop_check, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_run, ctx->d_scalars, ctx->d_llpart, ctx->ll_parts);
	if ((rc = run_estep(ctx, B, C, 1, stop, nullptr, true))) return rc;
	hipLaunchKernelGGL(k_stop_check, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_run, ctx->d_scalars, ctx->d_llpart, ctx->ll_parts);
	/* emll = log_likelihood(findex = C) -> d_scalars[1] (accel_em.c:53).  Admixture model: where the dual individual pass
	 * exists, emll is taken below, by the pass that visits the extrapolated point (same kernel arithmetic, same bits) */
	mchip_pass_args dual = pass_args(ctx, B);
	dual.stop = stop;
	dual.P2 = ctx->d_p[C];
	dual.Q2 = ctx->d_q[C];
	dual.llpart2 = ctx->d_llpart2;
	const bool use_dual = ctx->admixture && !ctx->knob.no_dual && ctx->kt->dual_available(dual);
	if (!ctx->admixture) {
		if ((rc = run_mixture(ctx, C, C, 0, 1, stop, 1))) return rc;	/* logL_mixture */
	} else if (!use_dual) {
		mchip_pass_args a = pass_args(ctx, C);
		a.stop = stop;
		prof_mark(ctx, MCHIP_KERN_LOGLIK, true);
		ctx->kt->loglik(a, ctx->stream);
		prof_mark(ctx, MCHIP_KERN_LOGLIK, false);
		hipLaunchKernelGGL(k_reduce_sum, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_llpart, ctx->kt->ind_ll_parts(a), ctx->d_scalars + 1, stop);
	}
	/* step size (accel_em.c:130-243) -> d_scalars[24] */
	{	/* (the secants u = B - A, v = C - B are formed inside the kernels that use them: nothing stores them in a batched cycle) */
		const int gq = (int)((ctx->nq + 4095) / 4096) > 512 ? 512 : (int)((ctx->nq + 4095) / 4096);
		const int gp = (int)((KT + 4095) / 4096) > 512 ? 512 : (int)((KT + 4095) / 4096);
		double *part_q = ctx->d_redpart, *part_p = ctx->d_redpart + 3 * 512;
		hipLaunchKernelGGL(k_dots_slots, dim3(gq + gp), dim3(MCHIP_BLOCK), 0, ctx->stream, ctx->d_q[A], ctx->d_q[B], ctx->d_q[C], nq, part_q, (unsigned)gq,
				   ctx->d_p[A], ctx->d_p[B], ctx->d_p[C], KT, part_p, stop);
		/* step size (accel_em.c:130-243) -> d_scalars[24], by the launch that adds up the partial dot products */
		hipLaunchKernelGGL(k_reduce_dots_step, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, part_q, gq, part_p, gp, ctx->d_scalars, scheme, cyc, stop);
	}
	/* accelerated_update (accel_em.c:422-551): B = A - 2 s u + s^2 (v - u) (or A + u + s v), projected; its log likelihood,
	 * taken by the pass that also leaves the S-side sums -> d_scalars[2] */
	hipLaunchKernelGGL(k_accel_update_slots, dim3(nblk(nq + KT)), dim3(256), 0, ctx->stream, ctx->d_q[A], ctx->d_q[B], ctx->d_q[C], ctx->d_q[B], nq,
			   ctx->d_p[A], ctx->d_p[B], ctx->d_p[C], ctx->d_p[B], KT, scheme == 4, ctx->d_scalars + 24, stop);
	if ((rc = project_slot(ctx, B, stop))) return rc;
	if (!ctx->admixture) {
		if ((rc = run_mixture(ctx, B, B, 0, 1, stop, 2))) return rc;	/* nothing of this pass serves the next E step */
	} else {
		if (use_dual) {
			prof_mark(ctx, MCHIP_KERN_DUAL, true);
			ctx->kt->accum_q_dual(dual, ctx->stream);
			prof_mark(ctx, MCHIP_KERN_DUAL, false);
		} else {
			mchip_pass_args a = pass_args(ctx, B);
			a.stop = stop;
			prof_mark(ctx, MCHIP_KERN_ACCUM_Q, true);
			ctx->kt->accum_q(a, ctx->stream);
			prof_mark(ctx, MCHIP_KERN_ACCUM_Q, false);
		}
	}
	/* the two log likelihoods' sums and accept iff ll > emll, one launch; the outcome goes back to slot A */
	if (ctx->admixture)
		hipLaunchKernelGGL(k_reduce_accept, dim3(1), dim3(MCHIP_BLOCK), 0, ctx->stream, use_dual ? ctx->d_llpart2 : (const double *)nullptr, ctx->d_llpart,
				   ctx->kt->ind_ll_parts(dual), ctx->d_scalars, cyc, stop);
	else
		hipLaunchKernelGGL(k_accept, dim3(1), dim3(64), 0, ctx->stream, ctx->d_scalars, cyc, stop);
	hipLaunchKernelGGL(k_select_copy, dim3(nblk(nq + KT)), dim3(256), 0, ctx->stream, ctx->d_q[A], ctx->d_q[B], ctx->d_q[C], nq,
			   ctx->d_p[A], ctx->d_p[B], ctx->d_p[C], KT, cyc, stop);
	HIPCHK(hipGetLastError());
	return MCHIP_OK;
}

int mchip_accel_run(mchip_context *ctx, int slot, int scheme, int n_cycles, mchip_run_state *state)
{
	MCHIP_ENTRY();
	int rc = check_slot(ctx, slot);
	if (rc) return rc;
	if (!state || n_cycles < 1 || scheme < 1 || scheme > 4) return fail(ctx, MCHIP_ERR_INVALID, "accel_run: bad arguments%s", nullptr);
	if ((ctx->admixture && !ctx->sparse) || ctx->nsec < 1)
		return fail(ctx, MCHIP_ERR_UNSUPPORTED, "accel_run: needs a secant pair; loci with more than 32 alleles run cycle by cycle%s", nullptr);
	if (ctx->qstride) ctx->empty_rows_nan[0] = ctx->empty_rows_nan[1] = ctx->empty_rows_nan[2] = 1;	/* every slot is written by the cycle's M steps */
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipMemcpyAsync(ctx->d_run, state, sizeof *state, hipMemcpyHostToDevice, ctx->stream));
	/* does Spart hold the S-side sums of this very slot (mchip_loglik_prefetch, or the accepted cycle that ended the
	 * previous call)?  The first E step of the batch is told through the device flag the later cycles set themselves */
	const int cyc0[4] = { 0, (ctx->admixture && ctx->s_cache_slot == slot) ? 1 : 0, 0, 0 };
	HIPCHK(hipMemcpyAsync(ctx->d_cyc, cyc0, sizeof cyc0, hipMemcpyHostToDevice, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));	/* cyc0 is a stack array */
	ctx->s_cache_slot = -1;
	bool use_graph = !ctx->profiling && !ctx->knob.no_graph;
	hipGraphExec_t &exec = ctx->cycle_graph[slot][scheme];
	if (use_graph && !exec) {
		hipGraph_t graph = nullptr;
		if (MCHIP_WAIT(hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeThreadLocal)) == hipSuccess) {
			rc = accel_cycle_enqueue(ctx, slot, scheme);
			const hipError_t e = MCHIP_WAIT(hipStreamEndCapture(ctx->stream, &graph));
			if (rc || e != hipSuccess || !graph || MCHIP_WAIT(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0)) != hipSuccess) exec = nullptr;
			if (graph) (void)MCHIP_WAIT(hipGraphDestroy(graph));
			(void)hipGetLastError();
		}
		if (!exec) use_graph = false;
	}
	for (int c = 0; c < n_cycles; c++) {
		if (use_graph) HIPCHK(hipGraphLaunch(exec, ctx->stream));
		else if ((rc = accel_cycle_enqueue(ctx, slot, scheme))) return rc;
	}
	int cyc_out[4] = { 0, 0, 0, 0 };
	HIPCHK(hipMemcpyAsync(state, ctx->d_run, sizeof *state, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipMemcpyAsync(cyc_out, ctx->d_cyc, sizeof cyc_out, hipMemcpyDeviceToHost, ctx->stream));
	HIPCHK(hipStreamSynchronize(ctx->stream));
	/* a batch that ran to its end on an accepted cycle leaves that cycle's sums for whoever continues from this slot */
	ctx->s_cache_slot = (ctx->admixture && !state->stopped && cyc_out[1]) ? slot : -1;
	ctx->have_ll = 1;
	return MCHIP_OK;
}

int mchip_multisecant_update(mchip_context *ctx, int to, int base, int u_index, int n_terms,
			     const int *v_index, const double *coef_a, const double *coef_b)
{
	MCHIP_ENTRY();
	int rc = check_secant(ctx, u_index);
	if (rc) return rc;
	ctx->s_cache_slot = -1;
	if ((rc = check_slot(ctx, to)) || (rc = check_slot(ctx, base))) return rc;
	if (n_terms < 0 || (n_terms && (!v_index || !coef_a || !coef_b))) return MCHIP_ERR_INVALID;
	for (int t = 0; t < n_terms; t++) if ((rc = check_secant(ctx, v_index[t]))) return rc;
	HIPCHK(hipSetDevice(ctx->device));
	const size_t KT = (size_t)ctx->K * ctx->T;
	hipLaunchKernelGGL(k_add, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_p[base], ctx->d_up[u_index], ctx->d_p[to], KT);
	hipLaunchKernelGGL(k_add, dim3(nblk(ctx->nq)), dim3(256), 0, ctx->stream, ctx->d_q[base], ctx->d_uq[u_index], ctx->d_q[to], (size_t)ctx->nq);
	for (int t = 0; t < n_terms; t++) {
		hipLaunchKernelGGL(k_axpy2, dim3(nblk(KT)), dim3(256), 0, ctx->stream, ctx->d_vp[v_index[t]], ctx->d_p[to], KT, coef_a[t], coef_b[t]);
		hipLaunchKernelGGL(k_axpy2, dim3(nblk(ctx->nq)), dim3(256), 0, ctx->stream, ctx->d_vq[v_index[t]], ctx->d_q[to], (size_t)ctx->nq, coef_a[t], coef_b[t]);
	}
	HIPCHK(hipGetLastError());
	ctx->empty_rows_nan[to] = ctx->empty_rows_nan[base];
	return project_slot(ctx, to);
}

/* ---- measurement hooks ---- */
int mchip_profile_begin(mchip_context *ctx)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	HIPCHK(hipSetDevice(ctx->device));
	ctx->profiling = 1;
	ctx->ev_used = 0;
	ctx->ev_kind.clear();
	HIPCHK(hipEventRecord(ctx->ev_begin, ctx->stream));
	return MCHIP_OK;
}

int mchip_profile_end(mchip_context *ctx, double *total_ms, double *kernel_ms, int *launches)
{
	MCHIP_ENTRY();
	if (!ctx) return MCHIP_ERR_INVALID;
	if (!ctx->profiling) return fail(ctx, MCHIP_ERR_STATE, "profile_end without profile_begin%s", nullptr);
	HIPCHK(hipSetDevice(ctx->device));
	HIPCHK(hipEventRecord(ctx->ev_end, ctx->stream));
	HIPCHK(hipEventSynchronize(ctx->ev_end));
	ctx->profiling = 0;
	float ms = 0;
	HIPCHK(hipEventElapsedTime(&ms, ctx->ev_begin, ctx->ev_end));
	if (total_ms) *total_ms = ms;
	double km[MCHIP_KERN_COUNT] = {0}, longest[MCHIP_KERN_COUNT] = {0};
	int kl[MCHIP_KERN_COUNT] = {0};
	std::vector<float> each(ctx->ev_kind.size(), 0.0f);
	for (size_t p = 0; 2 * p + 1 < ctx->ev_used && p < ctx->ev_kind.size(); p++) {
		HIPCHK(hipEventElapsedTime(&each[p], ctx->ev_pool[2 * p], ctx->ev_pool[2 * p + 1]));
		if (each[p] > longest[ctx->ev_kind[p]]) longest[ctx->ev_kind[p]] = each[p];
	}
	/* a launch that returned at once (the S-side pass of a batched accelerated cycle whose sums were already in place, any
	 * kernel after the stopping rule fired) did no pass over the data: it is not a launch of that kernel for these figures */
	for (size_t p = 0; 2 * p + 1 < ctx->ev_used && p < ctx->ev_kind.size(); p++) {
		const int kind = ctx->ev_kind[p];
		if (each[p] < 0.05 * longest[kind]) continue;
		km[kind] += each[p];
		kl[kind]++;
	}
	for (int x = 0; x < MCHIP_KERN_COUNT; x++) {
		if (kernel_ms) kernel_ms[x] = km[x];
		if (launches) launches[x] = kl[x];
	}
	return MCHIP_OK;
}

/* ---- where the process stands (mchip_progress.h) ---- */
int mchip_progress_note(const char *what)
{
	mchip_progress_slot *s = mchip_progress_my_slot();
	s->note.store(what, std::memory_order_relaxed);
	s->since_ns.store(mchip_progress_now(), std::memory_order_relaxed);
	mchip_progress_events.fetch_add(1, std::memory_order_relaxed);
	return MCHIP_OK;
}

int mchip_progress_report(char *buf, int buf_len, unsigned long long *events)
{
	if (events) *events = mchip_progress_events.load(std::memory_order_relaxed);
	if (!buf || buf_len <= 0) return MCHIP_OK;
	int n = progress_next_slot.load(std::memory_order_relaxed), used = 0;
	if (n > MCHIP_PROGRESS_SLOTS) n = MCHIP_PROGRESS_SLOTS;
	const long long now = mchip_progress_now();
	buf[0] = 0;
	for (int x = 0; x < n && used < buf_len - 1; x++) {
		const mchip_progress_slot &s = mchip_progress_slots[x];
		const char *entry = s.entry.load(std::memory_order_relaxed), *wait = s.wait.load(std::memory_order_relaxed);
		const char *note = s.note.load(std::memory_order_relaxed);
		const int w = snprintf(buf + used, (size_t)(buf_len - used), "thread %ld: %s%s%s%s%s%s, %.1f s since its last event\n",
				       s.tid.load(std::memory_order_relaxed), entry ? "in " : "outside the library", entry ? entry : "",
				       wait ? ", waiting in " : "", wait ? wait : "", note ? "; last phase: " : "", note ? note : "",
				       1e-9 * (double)(now - s.since_ns.load(std::memory_order_relaxed)));
		if (w < 0) break;
		used += w < buf_len - used ? w : buf_len - used - 1;
	}
	return MCHIP_OK;
}

}  /* extern "C" */
