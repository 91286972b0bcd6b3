/*
 * mc_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C, flat arrays) of MULTICLUST's EM hot path, each function citing the
 * reference file:line whose arithmetic and operation order it follows.  It is the *checker* for the
 * HIP path: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 * The product (multiclust_amd/, include/) never links, imports or calls anything in oracle/.
 *
 * Parity of this restatement is PINNED: tests/test_oracle_golden.py checks it bit-for-bit (or to the
 * stated ulp bound) against golden vectors dumped from the reference itself (oracle/ref_harness.c,
 * oracle/make_fixtures.py -> tests/golden/).
 */
#ifndef MC_ORACLE_H
#define MC_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MCO_MISSING_IDX 0xFF

enum { MCO_NONE = 0, MCO_SQS1, MCO_SQS2, MCO_SQS3, MCO_QN /* 4 = QN q=1, 5 = q2, 6 = q3 */ };

typedef struct mco_data mco_data;
typedef struct mco_model mco_model;

typedef struct mco_options {
	int admixture;		/* options::admixture */
	int eta_constrained;	/* options::eta_constrained (-c) */
	int do_projection;	/* options::do_projection (default 1, multiclust.c:942) */
	int accel_scheme;	/* 0 none, 1-3 SQUAREM, 4-6 QN q=1..3 (multiclust.c:818-820) */
	int n_init_iter;	/* options::n_init_iter */
	int max_iter;		/* options::max_iter (-T) */
	int adjust_step;	/* options::adjust_step (-g) */
	double abs_error;	/* options::abs_error (-E), default 1e-4 */
	double rel_error;	/* options::rel_error (-e), default 0 */
	double lower_bound;	/* eta_lower_bound == p_lower_bound after synchronize() */
	int fused;		/* 0: materialise d_iklm exactly as the reference; 1: fused re-association (GPU order) */
} mco_options;

/* multiclust.c:812-813 */
double mco_lower_bound(double user_lower_bound, int I, int ploidy);

/* glibc 2.35 srand()/rand() (TYPE_3 additive feedback), restated so results do not depend on the host libc */
typedef struct mco_rng { int32_t r[31]; int f, b; } mco_rng;
void mco_srand(mco_rng *g, unsigned int seed);
int mco_rand(mco_rng *g);

/* data: geno[I][L][ploidy] allele index into locus l's ascending allele list, 0xFF = missing
 * (equivalent of dat->IL + dat->L_alleles; counts ILM are derived as read_file.c:651-657 does). */
mco_data *mco_data_create(int I, int L, int ploidy, const int32_t *uniquealleles, const uint8_t *geno);
void mco_data_free(mco_data *d);
int mco_data_T(const mco_data *d);
const int32_t *mco_data_ilm(const mco_data *d);	/* [I][T] */

mco_model *mco_model_create(const mco_data *d, const mco_options *o, int K);
void mco_model_free(mco_model *m);
/* parameter slots 0..2 (model::vpklm / vetaik / vetak). p: [K][T]; q: [I][K] or [K]. */
double *mco_model_p(mco_model *m, int slot);
double *mco_model_q(mco_model *m, int slot);
double *mco_model_sik(mco_model *m);	/* [I][K]: sum_lm d_iklm of the last E-step (admixture) or vik (mixture) */
double *mco_model_u_p(mco_model *m, int j);
double *mco_model_v_p(mco_model *m, int j);
double *mco_model_u_q(mco_model *m, int j);	/* eta parts of the secant pairs (u_etaik / u_etak, v_etaik / v_etak) */
double *mco_model_v_q(mco_model *m, int j);
int mco_model_delta_index(const mco_model *m);
void mco_model_set_delta_index(mco_model *m, int delta_index);	/* with the secants: the state of a QN run with q > 1 between cycles */
int mco_model_q_len(const mco_model *m);
void mco_model_reset(mco_model *m);	/* initialize_model() state reset, rnd_init.c:58-71 + multiclust.c:518-524 */
double mco_model_logL(const mco_model *m);
int mco_model_n_iter(const mco_model *m);
int mco_model_converged(const mco_model *m);
int mco_model_pindex(const mco_model *m);
int mco_model_findex(const mco_model *m);
int mco_model_tindex(const mco_model *m);
int mco_model_fatal(const mco_model *m);	/* 1: NaN logL, 2: logL decrease (reference would exit(0), em_alg.c:106-120) */

/* simplex.c:109-143 */
void mco_michelot_project(double *x, int len, double sum, double min);

/* rnd_init.c:349-357,456-482: random allele partition (one rand()%K per allele copy) + first M-step into slot tindex */
void mco_random_initialize_admixture(const mco_data *d, const mco_options *o, mco_model *m, mco_rng *g);
/* same, from a given assignment assign[I][L][ploidy] (k per allele copy) */
void mco_initialize_from_partition(const mco_data *d, const mco_options *o, mco_model *m, const uint8_t *assign);

/* Rand-EM (the reference carries it but its command line cannot select it, multiclust.c:935): random_allele_center
 * (rnd_init.c:496-583) fills ilk[I][L][ploidy]; initialize_parameters_admixture (603-705); random_individual_center (192-259)
 * fills I_K[I]; initialize_parameters_mixture (268-339); randem_initialize_* (123-160, 412-444) keeps the best of n candidates
 * scored by em_e_step (em_alg.c:219-233) and writes their log likelihoods to ll_out (may be NULL) */
void mco_random_allele_center(const mco_data *d, int K, mco_rng *g, uint8_t *ilk);
void mco_initialize_parameters_admixture(const mco_data *d, const mco_options *o, mco_model *m, const uint8_t *ilk);
void mco_random_individual_center(const mco_data *d, int K, mco_rng *g, int *I_K);
void mco_initialize_parameters_mixture(const mco_data *d, mco_model *m, const int *I_K);
void mco_randem_initialize(const mco_data *d, const mco_options *o, mco_model *m, mco_rng *g, int n_rand_em_init, double *ll_out);
double mco_em_e_step(const mco_data *d, const mco_options *o, mco_model *m);

/* em_alg.c:195-207: E (findex) + M (tindex) + stop(); returns stop flag */
int mco_em_step(const mco_data *d, const mco_options *o, mco_model *m);
/* E step only (findex): returns logL, refreshes sik */
double mco_e_step(const mco_data *d, const mco_options *o, mco_model *m);
/* log_likelihood.c:56-62 */
double mco_log_likelihood(const mco_data *d, const mco_options *o, mco_model *m, int which);
/* em_alg.c:1072-1211 */
int mco_em_2_steps(const mco_data *d, const mco_options *o, mco_model *m);
/* accel_em.c:130-243 */
double mco_step_size(const mco_data *d, const mco_options *o, mco_model *m);
/* accel_em.c:422-551 */
double mco_accelerated_update(const mco_data *d, const mco_options *o, mco_model *m, double s);
/* accel_em.c:35-114; trace (may be NULL): emll, s, ll, accepted */
int mco_accelerated_em_step(const mco_data *d, const mco_options *o, mco_model *m, double *trace4);
/* em_alg.c:44-90 */
void mco_em(const mco_data *d, const mco_options *o, mco_model *m);

/* maximize_likelihood bookkeeping over initialisations (multiclust.c:477-486, 534-560) */
typedef struct mco_summary {
	int n_init, n_total_iter, n_max_iter, n_maxll_times, n_maxll_init, ever_converged, best_unit;
	double max_logL, first_max_logL;
} mco_summary;
void mco_summary_reset(mco_summary *s);
void mco_summary_add(const mco_options *o, mco_summary *s, int unit, double logL, int converged, int n_iter, int time_stop);
/* n_units initialisations from one continuing rand() stream, each followed by em(); per_unit[u] = {logL, converged,
 * n_iter, pindex} */
void mco_maximize_likelihood(const mco_data *d, const mco_options *o, mco_model *m, mco_rng *g, int n_units,
			     double *per_unit, mco_summary *s);

#ifdef __cplusplus
}
#endif
#endif
