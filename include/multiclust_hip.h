/*
 * multiclust_hip.h -- C-ABI of the MI355X-native EM hot path of MULTICLUST.
 *
 * Drop-in boundary.  The reference has no FFI: the seam is the set of C functions
 * its driver/init/writer layers call on the EM layer, all taking (options*, data*, model*)
 * (reference multiclust.h:371-388).  Each entry point below names the reference function(s) it replaces
 * (file:line under the reference checkout).  Conventions:
 *   - extern "C", plain pointers and sizes; no pointer into device memory crosses this boundary;
 *   - every function returns int, 0 = MCHIP_OK (reference: NO_ERROR, message.h:21); nothing here calls
 *     exit() -- the host keeps the reference's exit(0)-on-NaN/decrease semantics (em_alg.c:106-120);
 *   - an opaque context owns one HIP stream and all device buffers for (device, data set, K); contexts
 *     share nothing, so independent initialisations / bootstrap replicates can run one per stream;
 *   - parameters cross the boundary in the reference's own flat order:
 *       P[k][l][m]  -> double[K*T],  T = sum_l uniquealleles[l], index k*T + T_off[l] + m  (vpklm[slot][k][l][m])
 *       Q[i][k]     -> double[I*K]   (vetaik[slot][i][k]);  double[K] when eta is constrained / mixture (vetak[slot])
 *   - slots 0..2 are the reference's triple buffers vpklm[3]/vetaik[3] (multiclust.h:266-271).
 */
#ifndef MULTICLUST_HIP_H
#define MULTICLUST_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCHIP_ABI_VERSION 2	/* 2: mchip_profile_end fills MCHIP_PROF_KINDS = 4 entries (was 3); entry points added since 1 only */
#define MCHIP_MISSING 0xFF	/* genotype byte for a missing allele copy (reference MISSING = -9, multiclust.h:140) */
#define MCHIP_MAX_K 64		/* clusters supported by the K-specialised kernels (cost ~ 2K+6 per cell up to K ~ 20; above
				 * that the 2K-4K doubles a lane holds cost occupancy, above 32 they spill) */
#define MCHIP_MAX_SECANTS 3	/* options::q <= 3 without LAPACK (multiclust.c:847-851) */

enum mchip_status {
	MCHIP_OK = 0,
	MCHIP_ERR_INVALID = 1,		/* bad argument / shape */
	MCHIP_ERR_NO_DEVICE = 2,	/* no HIP device: the product path fails loudly, there is no CPU fallback */
	MCHIP_ERR_HIP = 3,		/* a HIP runtime call failed; see mchip_last_error() */
	MCHIP_ERR_ALLOC = 4,
	MCHIP_ERR_STATE = 5,		/* called out of order (no genotypes / no model yet) */
	MCHIP_ERR_UNSUPPORTED = 6
};

typedef struct mchip_context mchip_context;

int mchip_abi_version(void);
int mchip_device_count(int *count);
/* One context per (device, stream).  Replaces make_model/allocate_model_for_k ownership (multiclust.c:1087,1181). */
int mchip_create(mchip_context **ctx, int device);
int mchip_destroy(mchip_context *ctx);
const char *mchip_last_error(const mchip_context *ctx);
int mchip_synchronize(mchip_context *ctx);

/*
 * Upload one data set.  Replaces dat->IL / dat->L_alleles / dat->ILM as built by read_file.c:443-663:
 * geno[i][l][a] (I*L*ploidy bytes) is the index of allele copy a of individual i at locus l in that
 * locus's ascending allele list (the reference's own index form dat->ila, read_file.c:374-401), or
 * MCHIP_MISSING.  uniquealleles[l] = M_l includes the phantom trailing slot the reference creates for
 * loci with missing data (read_file.c:527-533); indices must be < M_l.  Called again per bootstrap
 * replicate (bootstrap.c:35-41 swaps dat->ILM).
 */
int mchip_set_genotypes(mchip_context *ctx, int I, int L, int ploidy,
			const int32_t *uniquealleles, const uint8_t *geno);
/* Counted on the device when a data set is installed: cells (i, l, m) with ILM[i][l][m] > 0 -- the cells the reference's E step
 * visits (em_alg.c:338-342) and the unit the flop count of the path is stated in -- and non-missing allele copies. */
int mchip_data_counts(mchip_context *ctx, uint64_t *nonempty_cells, uint64_t *allele_copies);
/* Individuals of the data set held without a single observed allele copy (returns how many; *first = the first one's index or
 * -1).  Under the admixture model with individual mixing proportions the reference gives such a row 0 / 0 = NaN in its first M
 * step (em_alg.c:685-690) and, NaN then sitting in the secants, its step size is NaN and every accelerated cycle falls back to
 * its EM iterate (accel_em.c:58-62): mchip_get_q reports the row as NaN, the host side (mc_step_size) does the falling back. */
int mchip_empty_individuals(const mchip_context *ctx, int *first);
/* the data set currently held, back in the upload form [I][L][ploidy] */
int mchip_get_genotypes(mchip_context *ctx, uint8_t *geno);
/*
 * A parametric-bootstrap data set generated on the device instead of uploaded: parametric_bootstrap_admixture
 * (bootstrap.c:84-124).  Same shape arguments as mchip_set_genotypes; allele copy j (i, l, n order) uses draws 2j and
 * 2j+1 of the rand() stream described by `window` (see mchip_mstep_from_rand_partition): source cluster by the
 * inverse-CDF walk over q[i][.] (or q[.] when eta_constrained), then the allele by the walk over p[k][l][.], with
 * r = rand() / RAND_MAX and the reference's left-to-right partial sums.  q is [I][K] (or [K]), p is [K][T] (the H0
 * MLEs, multiclust.c:562-581).  Consumes 2*I*L*ploidy draws; every copy is simulated, as in the reference's default
 * build.  Like mchip_set_genotypes it drops any model: call mchip_set_model next.  When the context already holds a data set
 * of the same shape and allele lists (the previous replicate), every device buffer is re-used, the observed haplotypes
 * installed by mchip_set_init_genotypes stay in force (the reference's dat->IL stays in place across replicates,
 * bootstrap.c:35-41), and an mchip_set_model with unchanged arguments re-uses the model's buffers too.
 */
int mchip_simulate_genotypes(mchip_context *ctx, int I, int L, int ploidy, const int32_t *uniquealleles,
			     const uint32_t *window, int K, int eta_constrained, const double *q, const double *p);
/*
 * The data set `src` holds, copied into `ctx` on the device (both contexts on one device).  run_bootstrap fits the null and the
 * alternative model to the SAME simulated data set (multiclust.c:675-708: one parametric_bootstrap(), then estimate_model()
 * over both K): the second model's context takes the first one's instead of generating it again.  Like
 * mchip_simulate_genotypes it keeps the init genotypes in force when the shape is unchanged and drops any model.
 */
int mchip_copy_genotypes(mchip_context *ctx, const mchip_context *src);
/*
 * The genotype the hard-partition M step reads, when it is not the data set itself.  While bootstrapping, the
 * reference's random_allele_partition still reads the observed haplotypes dat->IL (rnd_init.c:471;
 * parametric_bootstrap replaces only dat->ILM, bootstrap.c:35-41), so every fit to a simulated data set starts from
 * the observed alleles.  After this call mchip_mstep_from_partition / _from_rand_partition on this context do the
 * same; geno = NULL (and any new data set) goes back to the data set itself.  Same form as mchip_set_genotypes.
 */
int mchip_set_init_genotypes(mchip_context *ctx, const uint8_t *geno);

/*
 * Allocate parameter ring, secant buffers and workspaces for K.  Replaces allocate_model_for_k
 * (multiclust.c:1181-1265) and free_model_data (1288).  admixture/eta_constrained/do_projection are
 * options::admixture / eta_constrained / do_projection; the lower bounds are options::eta_lower_bound and
 * p_lower_bound after synchronize() (multiclust.c:812-815); n_secants is options::q.
 */
int mchip_set_model(mchip_context *ctx, int K, int admixture, int eta_constrained, int do_projection,
		    double eta_lower_bound, double p_lower_bound, int n_secants);

/* Parameter slots: mod->vpklm[slot], mod->vetaik[slot] / mod->vetak[slot].
 * Rows of individuals without a single observed copy (mchip_empty_individuals): the reference holds 0 / 0 = NaN for them from the
 * first M step on (em_alg.c:685-690); the device holds a finite 1 / K (nothing such a row is multiplied into carries a count) and
 * remembers PER SLOT whether the slot's rows stand for that NaN: set by whatever writes the slot from an M step (mchip_em_step's
 * `to`, mchip_em_run, every slot of mchip_accel_run, mchip_mstep_from_*partition), inherited by mchip_copy_slot,
 * mchip_accel_update and mchip_multisecant_update from the slot they start from, cleared by mchip_init_from_allele_centers
 * (counts start at 1 there: the row is 1 / K in the reference too).  mchip_get_q reports such rows as NaN; mchip_set_q accepts
 * them back -- a NaN row of such an individual is stored as 1 / K and the slot keeps reporting NaN (a get_q / set_q round trip
 * is safe), a finite row is stored and reported as given. */
int mchip_set_p(mchip_context *ctx, int slot, const double *p);
int mchip_get_p(mchip_context *ctx, int slot, double *p);
int mchip_set_q(mchip_context *ctx, int slot, const double *q);
int mchip_get_q(mchip_context *ctx, int slot, double *q);
int mchip_q_length(const mchip_context *ctx, int *n);	/* I*K or K */
int mchip_p_length(const mchip_context *ctx, int *n);	/* K*T */

/*
 * One EM iteration: E step on slot `from`, M step into slot `to` (may be equal: in place, as the
 * unaccelerated reference does).  Replaces e_step_admixture_orig + m_step_admixture_orig
 * (em_alg.c:291-486, 592-754) or e_step_mixture + m_step_mixture (763-1011) as sequenced by em_step
 * (195-207); d_iklm is never materialised.  *loglik = log likelihood of the `from` parameters (what the
 * reference's E step returns).  loglik may be NULL: the step is then only enqueued (no host sync);
 * fetch the value later with mchip_last_loglik().
 */
int mchip_em_step(mchip_context *ctx, int from, int to, double *loglik);
int mchip_last_loglik(mchip_context *ctx, double *loglik);
/*
 * A batch of in-place EM iterations with the stopping rule evaluated on the device: n_steps times
 * { em_step(slot -> slot); stop() } exactly as em() sequences them for the unaccelerated case (em_alg.c:78-88, 101-182),
 * enqueued without a host round trip per iteration; once the rule fires (convergence, iteration cap, NaN, decrease) the
 * remaining steps of the batch are no-ops, so parameters, n_iter and logL are those of the stopping iteration.
 * `state` is in/out: the host seeds it with model::logL / n_iter and reads back where the loop stands.
 * Available for every model (admixture with individual or shared mixing proportions, mixture).
 */
typedef struct mchip_run_state {
	double logL;		/* model::logL: log likelihood of the previous iteration (in), of the last executed one (out) */
	double bad_loglik;	/* the offending value when fatal != 0 */
	double abs_error, rel_error;	/* options::abs_error / rel_error */
	int n_iter;		/* model::n_iter */
	int max_iter;		/* options::max_iter (0 = unlimited) */
	int stopped, converged, iter_stop;
	int fatal;		/* 0, 1 = NaN log likelihood, 2 = log likelihood decrease (em_alg.c:106-120) */
} mchip_run_state;
int mchip_em_run(mchip_context *ctx, int slot, int n_steps, mchip_run_state *state);
/*
 * The same for the accelerated loop: n_cycles times accelerated_em_step (accel_em.c:35-114) as em() sequences it
 * (em_alg.c:84-88) for one secant pair and no back-tracking -- scheme 1, 2, 3 = SQUAREM S1-S3, 4 = QN with q = 1
 * (options::accel_scheme, multiclust.h:125-131) -- enqueued without a host round trip: em_2_steps with both stop() calls,
 * log_likelihood of the second EM iterate, step_size, accelerated_update with its log_likelihood, accept iff ll > emll,
 * all decided on the device.  `slot` holds the starting iterate (model::pindex) and, on return, the iterate the reference
 * would report (its pindex: the accepted extrapolation, else the second EM iterate, else -- when the stopping rule fired
 * inside em_2_steps -- the cycle's starting iterate, accel_em.c:44-45); the other two slots are scratch.  Same arithmetic as
 * driving the cycle call by call (mchip_em_step, mchip_secant, mchip_loglik, mchip_step_dots, mchip_accel_update,
 * mchip_loglik_prefetch).  Every model (admixture with individual or shared mixing proportions, mixture) with
 * n_secants >= 1; MCHIP_ERR_UNSUPPORTED without a secant pair, or for an admixture data set with more than 32 alleles at a
 * locus (the caller then drives the cycles call by call).
 */
int mchip_accel_run(mchip_context *ctx, int slot, int scheme, int n_cycles, mchip_run_state *state);

/* E step only (em_e_step's trailing E step, em_alg.c:226,230): refreshes the expected counts, returns logL. */
int mchip_e_step(mchip_context *ctx, int slot, double *loglik);
/* log_likelihood(): logL_admixture / logL_mixture (log_likelihood.c:96-147, 157-232). */
int mchip_loglik(mchip_context *ctx, int slot, double *loglik);
/* The same value, computed by the pass that also accumulates the E step's per-individual sums and keeps them: an
 * mchip_em_step from this slot that follows (no parameter write in between) skips that pass.  accelerated_update's
 * log_likelihood(tindex) (accel_em.c:544) is followed, whenever the extrapolated point is accepted, by exactly such
 * an E step at the head of the next cycle (em_alg.c:1089-1098). */
int mchip_loglik_prefetch(mchip_context *ctx, int slot, double *loglik);

/*
 * First M step from a hard allele partition: random_initialize_admixture (rnd_init.c:349-357) =
 * random_allele_partition (456-482) + m_step_admixture.  assign[i][l][a] in [0,K) is the cluster drawn for
 * allele copy a (the host draws it with the libc-compatible stream); d_iklm = 1 (not += 1) per matching copy.
 */
int mchip_mstep_from_partition(mchip_context *ctx, const uint8_t *assign, int to);
/*
 * The same with the partition drawn on the device: assign[j] = rand() % K for j = 0 .. I*L*ploidy-1 in the
 * reference's i, l, a order (rnd_init.c:456-467), where rand() is glibc's TYPE_3 generator
 * x_j = x_{j-31} + x_{j-3} (mod 2^32), rand() = x_j >> 1, and window[t] = x_{j0-31+t}, t = 0..30, are the 31 words
 * behind the first draw (oldest first; the host's generator state after srand() and any draws made so far).  The
 * caller advances its own copy of the stream by I*L*ploidy draws.  Bit-identical to drawing on the host and
 * calling mchip_mstep_from_partition; removes I*L*ploidy host rand() calls and the upload per initialisation.
 */
int mchip_mstep_from_rand_partition(mchip_context *ctx, const uint32_t *window, int to);

/*
 * One Rand-EM candidate: random_allele_center's partition + initialize_parameters_admixture (rnd_init.c:496-705) into slot
 * `to`.  The host walks the loci in order and draws each locus's center alleles from the rand() stream exactly as the
 * reference does (K draws plus uniqueness retries when the locus has at least K alleles, none otherwise), which also tells it
 * how many copies of the locus match no center and therefore where each locus's rand() % K draws lie in the stream:
 *   centers[l*K + k]  allele index of cluster k's center at locus l, 0xFF = none (the reference's -1);
 *   draw_offset[l]    position, counted from `window`, of the first draw that a non-matching copy of locus l consumes
 *                     (copies take them in i, a order; missing copies match no center);
 *   window, n_draws   the 31 words behind the candidate's first draw (as mchip_mstep_from_rand_partition) and the length
 *                     of the candidate's span of the stream.
 * The device assigns every copy (identity with a center, else its draw), counts -- eta_ik = (1 + copies of i in k) /
 * (ploidy L + K), missing copies included; p_klm = (1 + copies of allele m in k) / sum_m -- and normalises; nothing is
 * projected.  Reads the observed haplotypes when mchip_set_init_genotypes is in force (the reference reads dat->IL).
 */
int mchip_init_from_allele_centers(mchip_context *ctx, const uint8_t *centers, const uint64_t *draw_offset, const uint32_t *window,
				   uint64_t n_draws, int to);
/* parameters of slot `from` copied to slot `to`: Rand-EM keeps its best candidate (rnd_init.c:431-436 keeps the partition) */
int mchip_copy_slot(mchip_context *ctx, int to, int from);

/*
 * What the writers read from diklm / vik (write_file.c:359-381,446-459,531-542,593-598):
 * admixture: sik[i][k] = sum_{l,m} d_iklm of the last E step executed; mixture: vik[i][k].
 */
int mchip_get_expected_counts(mchip_context *ctx, double *sik);

/* ---- acceleration (accel_em.c, em_alg.c:1072-1211) ---- */
/* u (which=0) or v (which=1) secant j := x[to] - x[from], for p and eta (em_alg.c:1104-1161). */
int mchip_secant(mchip_context *ctx, int which, int j, int to, int from);
/* The secant buffers as model state: mod->u_pklm[j] / v_pklm[j] (which = 0 / 1) with u_etaik[j] / v_etaik[j] (or u_etak / v_etak),
 * multiclust.h:284-293, allocated by allocate_model_for_k (multiclust.c:1231-1258); p_part in the flat order of a P slot
 * (double[K*T]), q_part in that of a Q slot.  Together with the parameter slots and model::delta_index they are the whole state
 * of a quasi-Newton run with q > 1 between two cycles (em_alg.c:1171 rotates the slot the next pair goes to). */
int mchip_set_secant(mchip_context *ctx, int which, int j, const double *p_part, const double *q_part);
int mchip_get_secant(mchip_context *ctx, int which, int j, double *p_part, double *q_part);
/* utu, utvu, vutvu of secant pair j, eta terms then p terms (accel_em.c:143-184). out[3]. */
int mchip_step_dots(mchip_context *ctx, int j, double *out3);
/* u_{j1}.u_{j2} and u_{j1}.v_{j2} (quasi-Newton secant matrix, accel_em.c:291-310). out[2]. */
int mchip_secant_dots(mchip_context *ctx, int j1, int j2, double *out2);
/*
 * accelerated_update (accel_em.c:444-541): x[to] = x[base] - 2 s u_j + s^2 (v_j - u_j)  (qn_form = 0, SQUAREM)
 *                                       or x[base] + u_j + s v_j                         (qn_form = 1, QN q=1),
 * then simplex projection of every (k,l) block and every individual when do_projection.
 */
int mchip_accel_update(mchip_context *ctx, int to, int base, int j, double s, int qn_form);
/*
 * qn_accelerated_update (accel_em.c:364-415): x[to] = x[base] + u_{u_index}; then for t = 0..n_terms-1, in
 * order, x[to] += v_{v_index[t]} * coef_a[t] * coef_b[t]  (Ainv[j*q+n] and cutu[n]); then projection.
 */
int mchip_multisecant_update(mchip_context *ctx, int to, int base, int u_index, int n_terms,
			     const int *v_index, const double *coef_a, const double *coef_b);

/* ---- the path's one exchange step, for ONE process driving several GPUs ----
 * Independent fits (random initialisations multiclust.c:516-653, bootstrap replicates 681-701) run one per context on
 * different devices with no data-path collective; afterwards a single RCCL all-reduce over xGMI makes every device's
 * copy of the per-unit result table complete (rows are disjoint: op 0 = sum) or picks the best log likelihood
 * (op 1 = max).  host_bufs[d] is device d's table (count doubles), reduced in place.  RCCL is loaded lazily; without
 * it mchip_comm_create returns MCHIP_ERR_UNSUPPORTED.  (One process per GPU, as bench.py runs, exchanges through
 * torch.distributed instead.) */
typedef struct mchip_comm mchip_comm;
int mchip_comm_create(mchip_comm **comm, int n_devices, const int *devices);
int mchip_comm_all_reduce(mchip_comm *comm, double *const *host_bufs, int count, int op);
int mchip_comm_destroy(mchip_comm *comm);
const char *mchip_comm_last_error(const mchip_comm *comm);
/* what the communicator is and has done: devices it spans, the version of the RCCL it loaded (ncclGetVersion's code, 0 when the
 * library does not say) and the all-reduces completed on it -- the evidence a caller (or a test) has that the exchange went
 * through RCCL.  Any of the three pointers may be NULL. */
int mchip_comm_info(const mchip_comm *comm, int *n_devices, int *rccl_version, unsigned long long *n_reductions);

/* ---- measurement hooks (bench.py): HIP events on the context's own stream ---- */
int mchip_profile_begin(mchip_context *ctx);
/* total_ms: begin..end on the stream.  kernel_ms[MCHIP_PROF_KINDS] / launches[MCHIP_PROF_KINDS]: summed
 * durations and launch counts of the streaming passes over the genotype matrix, each launch bracketed
 * by its own event pair: [0] column pass of an EM step (N-side sums + logL), [1] individual pass
 * (S-side sums), [2] stand-alone log-likelihood pass, [3] dual individual pass of a batched accelerated cycle (the S-side
 * sums of the extrapolated point and the log likelihood of the second EM iterate in one pass; then [2] has no launches).
 * Launches that returned at once (batched runs: after the stopping
 * rule fired, or the S-side pass whose sums were already in place) took under 5 % of the kind's longest launch and are
 * left out of both figures. */
#define MCHIP_PROF_KINDS 4
int mchip_profile_end(mchip_context *ctx, double *total_ms, double *kernel_ms, int *launches);
/* device properties the bench reports (name, CU count, memory) */
int mchip_device_info(mchip_context *ctx, char *name, int name_len, int *compute_units, double *hbm_bytes);

/* ---- where the process stands: the record a watchdog reads when a run makes no progress ----
 * The reference is one thread of plain C that blocks on nothing but its own loops (em_alg.c:78-88, log_likelihood.c:212-221);
 * this library blocks on the HIP runtime (stream synchronisation behind every entry point that returns a value, copies,
 * allocation, graph capture, teardown).  Every entry point and every such runtime call records itself while it runs, and each of
 * those moments counts one event.  mchip_progress_report writes one line per host thread that has used the library -- the entry
 * point it is in, the runtime call it waits in (with source line), the last phase the host named, seconds since its last event
 * -- into buf (any length; truncated, NUL-terminated) and the process-wide event count into *events; either may be NULL / 0.
 * mchip_progress_note names a phase of the caller's own (a static string: the reader, the writers) and counts as an event.
 * multiclust's opt-in watchdog (MC_WATCHDOG_S=<seconds>, host/mc_fit.c) polls the count and, when it stands still for that long,
 * prints the report and leaves with _exit(3). */
int mchip_progress_report(char *buf, int buf_len, unsigned long long *events);
int mchip_progress_note(const char *what);

#ifdef __cplusplus
}
#endif
#endif
