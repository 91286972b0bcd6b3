/*
 * mc_host.h -- plain-C host side of the MI355X EM hot path.
 *
 * Mirrors the reference's EM-layer interface (reference multiclust.h:371-388: em, em_step, em_e_step,
 * em_2_steps, stop, converged, log_likelihood, accelerated_em_step, ...) with an mc_ prefix and the same
 * (options*, data*, model*) argument convention, the same ring-index / iteration-count / convergence
 * semantics and the same stderr lines; all array arithmetic goes through the C-ABI in
 * include/multiclust_hip.h (one mchip_context per model).  Differences that are deliberate:
 *   - fatal numerics (NaN logL, logL decrease) set model::fatal and stop instead of exit(0) inside the
 *     library (reference em_alg.c:106-120); the CLI turns fatal into the reference's exit(0).
 *   - parameters live on the device; mc_model_get_p/q fetch slot copies in the reference's flat order.
 */
#ifndef MC_HOST_H
#define MC_HOST_H

#include <stdint.h>
#include <stdio.h>
#include <time.h>
#include "multiclust_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* acceleration schemes (reference multiclust.h:125-131; 5, 6 = QN with q = 2, 3, multiclust.c:818-820) */
enum { MC_NONE = 0, MC_SQS1, MC_SQS2, MC_SQS3, MC_QN };
/* verbosity (reference message.h:45-53) */
enum { MC_ABSOLUTE_SILENCE = 0, MC_SILENT, MC_QUIET, MC_MINIMAL, MC_RESTRAINED, MC_TALKATIVE, MC_VERBOSE, MC_DEBUG };
/* initialisation procedures (reference multiclust.h:107-111) */
enum { MC_INIT_NOTHING = 0, MC_RAND_EM };
/* fatal conditions the reference answers with exit(0) */
enum { MC_FATAL_NONE = 0, MC_FATAL_NAN = 1, MC_FATAL_DECREASE = 2, MC_FATAL_DEVICE = 3 };

typedef struct mc_options {	/* subset of reference struct _options used by the EM layer (multiclust.h:155-215) */
	int admixture;
	int eta_constrained;
	int do_projection;
	int accel_scheme;
	int q;			/* number of secant conditions (QN) */
	int n_init_iter;
	int max_iter;
	unsigned int n_seconds;
	int adjust_step;
	int verbosity;
	double abs_error;
	double rel_error;
	double lower_bound;
	double eta_lower_bound;
	double p_lower_bound;
	unsigned int seed;
	int initialization_procedure;	/* MC_INIT_NOTHING (random initialisation) or MC_RAND_EM (multiclust.h:107-111) */
	int n_rand_em_init;		/* candidates per Rand-EM initialisation (-m, multiclust.c:936,1547) */
} mc_options;

typedef struct mc_data {	/* flat form of reference struct _data's genotype fields (multiclust.h:223-237) */
	int I, L, ploidy;
	const int32_t *uniquealleles;	/* [L] */
	const uint8_t *geno;		/* [I][L][ploidy] allele indices, MCHIP_MISSING = 0xFF */
	const uint8_t *init_geno;	/* NULL, or the observed data set while `geno` is a bootstrap replicate: the random
					 * partition of the admixture model keeps reading dat->IL (rnd_init.c:471) */
} mc_data;

typedef struct mc_model {	/* EM-layer state of reference struct _model (multiclust.h:259-360) */
	int K;
	int pindex, findex, tindex;
	int delta_index;
	double logL;
	int n_iter, converged, stopped, accel_step, iter_stop, time_stop;
	int fatal;
	clock_t start;
	double seconds_run;
	double A[9], Ainv[9], cutu[3];
	double last_emll, last_step, last_ll;	/* diagnostics of the last accelerated cycle */
	int last_accepted;
	mchip_context *dev;
	int owns_dev;
	void *init_cache;		/* per-locus allele counts of the observed haplotypes (Rand-EM), built on first use */
} mc_model;

/* glibc-compatible rand() stream (TYPE_3 additive feedback): "same seed" means the same draws as the
 * reference's srand()/rand() (multiclust.c:1592-1596, rnd_init.c:467) on any libc. */
typedef struct mc_rng { int32_t r[31]; int f, b; } mc_rng;
void mc_srand(mc_rng *g, unsigned int seed);
int mc_rand(mc_rng *g);

/* advance the stream by n draws in O(31^2 log n): the generator is the linear recurrence
 * x_j = x_{j-31} + x_{j-3} (mod 2^32), so x^n mod (x^31 - x^28 - 1) gives the state n steps ahead.  This is
 * what lets unit u of a sharded run start exactly where the serial program would. */
void mc_rng_jump(mc_rng *g, uint64_t n);

void mc_make_options(mc_options *opt);				/* defaults of make_options, multiclust.c:902-978 */
int mc_synchronize(mc_options *opt, const mc_data *dat);	/* lower bounds + q, multiclust.c:812-820 */

/* allocate_model_for_k (multiclust.c:1181): creates the device context on `device`, uploads dat, sizes for K */
int mc_model_create(mc_model **mod, const mc_options *opt, const mc_data *dat, int K, int device);
void mc_model_free(mc_model *mod);

/* Parametric bootstrap (bootstrap.c:31-175).  mc_bootstrap_genotypes draws one data set on the host from the fitted
 * parameters q ([I][K] or [K]) and p ([K][T]) of a K-cluster model, consuming `rng` exactly as the reference consumes
 * rand(); mc_bootstrap_draws is the number of draws that takes.  For the admixture model the same data set can be
 * generated on the device instead of uploaded: fill an mc_simulation with mc_simulation_begin (which also moves rng
 * past the data set's draws) and create the replicate's models with mc_model_create_simulated. */
typedef struct mc_simulation { uint32_t window[31]; int K; const double *q, *p; } mc_simulation;
void mc_bootstrap_genotypes(const mc_options *opt, const mc_data *dat, int K, const double *q, const double *p,
			    mc_rng *rng, uint8_t *geno);
uint64_t mc_bootstrap_draws(const mc_options *opt, const mc_data *dat);
void mc_simulation_begin(mc_simulation *sim, const mc_options *opt, const mc_data *dat, int K, const double *q,
			 const double *p, mc_rng *rng);
int mc_model_create_simulated(mc_model **mod, const mc_options *opt, const mc_data *dat, int K, int device,
			      const mc_simulation *sim);
int mc_model_resimulate(mc_model *mod, const mc_options *opt, const mc_data *dat, const mc_simulation *sim);
int mc_model_share_simulated(mc_model **mod, const mc_options *opt, const mc_data *dat, int K, int device, const mc_model *like);
int mc_model_get_genotypes(mc_model *mod, uint8_t *geno);
const char *mc_model_error(const mc_model *mod);
int mc_model_set_p(mc_model *mod, int slot, const double *p);
int mc_model_get_p(mc_model *mod, int slot, double *p);
int mc_model_set_q(mc_model *mod, int slot, const double *q);
int mc_model_get_q(mc_model *mod, int slot, double *q);
int mc_model_get_expected_counts(mc_model *mod, double *sik);

/* initialize_model (rnd_init.c:54-89) for the admixture model, random allele partition (349-357,456-482) */
int mc_initialize_model(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng);
void mc_reset_model_state(mc_model *mod);	/* multiclust.c:518-524 + rnd_init.c:58-71 */
/* random_initialize_mixture (rnd_init.c:103-110): random_individual_center + initialize_parameters_mixture, on the host */
int mc_initialize_mixture(const mc_data *dat, mc_model *mod, mc_rng *rng);
/* Rand-EM (rnd_init.c:123-160 mixture, 412-444 admixture): n_rand_em_init candidates -- random centers, parameters from the
 * partition, one EM iteration plus an E step (em_e_step) -- and the parameters of the candidate with the best log likelihood.
 * Admixture candidates are partitioned and counted on the device (mchip_init_from_allele_centers); the host only walks the
 * loci to draw the centers.  Consumes `rng` exactly as the reference consumes rand(). */
int mc_randem_initialize(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng);
/* Moves `rng` past n initialisations without performing them: where unit n of a sharded run starts in the serial stream.
 * A jump for the random allele partition; for Rand-EM and the mixture model the number of draws depends on the draws
 * themselves (center retries) and on the data (copies that match no center), so the host-side walk is replayed. */
int mc_skip_initializations(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng, int n);
/* starts[u], u = 0..n_units: the generator at the start of unit u of a sharded run (one walk over the units; `mod` supplies K and
 * holds the allele-count cache of the Rand-EM walk: a zeroed mc_model with K set will do, mc_init_cache_free() afterwards) */
int mc_unit_starts(const mc_options *opt, const mc_data *dat, mc_model *mod, const mc_rng *base, int n_units, mc_rng *starts);
void mc_init_cache_free(mc_model *mod);

void mc_em(const mc_options *opt, const mc_data *dat, mc_model *mod);			/* em_alg.c:44 */
int mc_em_step(const mc_options *opt, const mc_data *dat, mc_model *mod);		/* em_alg.c:195 */
double mc_em_e_step(const mc_options *opt, const mc_data *dat, mc_model *mod);		/* em_alg.c:219 */
int mc_em_2_steps(mc_model *mod, const mc_data *dat, const mc_options *opt);		/* em_alg.c:1072 */
int mc_stop(const mc_options *opt, mc_model *mod, double loglik);			/* em_alg.c:101 */
int mc_converged(const mc_options *opt, mc_model *mod, double loglik);			/* em_alg.c:163 */
double mc_log_likelihood(const mc_options *opt, const mc_data *dat, mc_model *mod, int which);	/* log_likelihood.c:56 */
int mc_accelerated_em_step(const mc_options *opt, const mc_data *dat, mc_model *mod);	/* accel_em.c:35 */
double mc_step_size(const mc_options *opt, const mc_data *dat, mc_model *mod);		/* accel_em.c:130 */
double mc_accelerated_update(const mc_options *opt, const mc_data *dat, mc_model *mod, double s);	/* accel_em.c:422 */
double mc_qn_accelerated_update(const mc_options *opt, const mc_data *dat, mc_model *mod);	/* accel_em.c:262 */
double mc_aic(double max_logL, int no_parameters);			/* log_likelihood.c:70 */
double mc_bic(double max_logL, int no_parameters, int I);		/* log_likelihood.c:82 */
int mc_no_parameters(const mc_options *opt, const mc_data *dat, int K);	/* multiclust.c:1267-1276 */

/* ---- several initialisations: maximize_likelihood (multiclust.c:471-656) ---- */
typedef struct mc_unit_result {	/* what one initialisation + em() leaves behind */
	int unit;
	double logL;
	int converged, n_iter, time_stop, iter_stop, pindex, fatal;
	double seconds_run;	/* CPU seconds since the model's start stamp when the fit stopped (em_alg.c:147) */
} mc_unit_result;

typedef struct mc_summary {	/* state maintained across initialisations (multiclust.h:337-353) */
	int n_init, n_total_iter, n_max_iter, n_maxll_times, n_maxll_init, ever_converged, best_unit;
	double max_logL, first_max_logL, aic, bic;
} mc_summary;

void mc_summary_reset(mc_summary *s);						/* multiclust.c:477-486 */
/* multiclust.c:534-560; results must be fed in unit order to reproduce the serial program */
void mc_summary_add(const mc_options *opt, mc_summary *s, const mc_unit_result *r, int no_parameters, int I);
/* rand() draws one admixture initialisation consumes (rnd_init.c:460-467: one per allele copy, missing included) */
uint64_t mc_draws_per_init(const mc_options *opt, const mc_data *dat, int K);
/* initialisation `unit` of a run seeded with `seed`: jump the stream to unit * draws_per_init, initialise, em() */
int mc_fit_unit(const mc_options *opt, const mc_data *dat, mc_model *mod, unsigned int seed, int unit, mc_unit_result *out);


/* ---- one bootstrap replicate: the other unit of sharding (run_bootstrap, multiclust.c:675-708) ---- */
typedef struct mc_replicate_result {
	int replicate;
	double logL_H0, logL_HA, ts;	/* best log likelihood of the null_K and alt_K fits, and their difference */
	int n_iter;			/* EM iterations of all fits of the replicate */
	int fatal;
} mc_replicate_result;
/* models[2] (may be NULL): the caller's null_K and alt_K models, created on first use and re-used for every later replicate
 * (free them with mc_model_free when the replicates are done); NULL: models are created and freed inside the call */
int mc_fit_replicate(const mc_options *opt, const mc_data *dat, int device, const mc_rng *base, int b, int null_K, int alt_K,
		     int n_init, int mle_K, const double *mle_q, const double *mle_p, mc_replicate_result *out, mc_model **models);

/* ---- opt-in watchdog (mc_watchdog.c): nothing in the reference corresponds -- it has nothing to wait for ----
 * mc_watchdog_start(s): a detached thread that polls the library's event count (mchip_progress_report) and, when it has stood
 * still for s seconds, prints where every thread stands (library record + /proc/self/task) on stderr and leaves with _exit(3).
 * mc_watchdog_from_env(): the same with s = $MC_WATCHDOG_S, nothing when the variable is unset.  mc_watchdog_report(): the
 * report alone (used by the tests). */
int mc_watchdog_start(double seconds);
int mc_watchdog_from_env(void);
void mc_watchdog_report(FILE *fp, double quiet_seconds);

#ifdef __cplusplus
}
#endif
#endif
