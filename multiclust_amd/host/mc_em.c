/*
 * mc_em.c -- host control flow of the EM hot path over the C-ABI (see mc_host.h).
 * Sequencing follows reference em_alg.c:44-233,1072-1211 and accel_em.c:35-551; every array operation
 * is a call into libmulticlust_hip.so.  No arithmetic on genotype-sized data happens on the host.
 */
#include "mc_host.h"

#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

static const char *accel_abbrev[] = { "EM", "S1", "S2", "S3", "QN" };

/* ------------------------------------------------------------------ libc-compatible rand() */
void mc_srand(mc_rng *g, unsigned int seed)
{
	int32_t word;
	if (seed == 0) seed = 1;
	word = (int32_t)seed;
	g->r[0] = word;
	for (int i = 1; i < 31; i++) {
		long hi = word / 127773, lo = word % 127773;
		word = (int32_t)(16807 * lo - 2836 * hi);
		if (word < 0) word += 2147483647;
		g->r[i] = word;
	}
	g->f = 3;
	g->b = 0;
	for (int i = 0; i < 310; i++) (void)mc_rand(g);
}

int mc_rand(mc_rng *g)
{
	uint32_t v = (uint32_t)g->r[g->f] + (uint32_t)g->r[g->b];
	g->r[g->f] = (int32_t)v;
	if (++g->f >= 31) g->f = 0;
	if (++g->b >= 31) g->b = 0;
	return (int)(v >> 1);
}

/* ------------------------------------------------------------------ options */
void mc_make_options(mc_options *opt)
{
	memset(opt, 0, sizeof *opt);
	opt->seed = 1234567;
	opt->max_iter = 0;
	opt->rel_error = 0;
	opt->abs_error = 1e-4;
	opt->lower_bound = opt->eta_lower_bound = opt->p_lower_bound = 1e-8;
	opt->do_projection = 1;
	opt->q = 1;
	opt->verbosity = MC_MINIMAL;
	opt->initialization_procedure = MC_INIT_NOTHING;	/* multiclust.c:935 */
	opt->n_rand_em_init = 50;				/* multiclust.c:936 */
}

int mc_synchronize(mc_options *opt, const mc_data *dat)
{
	/* multiclust.c:812-815 */
	double alt = 1.0 / dat->I / dat->ploidy - 0.5 / dat->I / dat->ploidy;
	if (alt < opt->lower_bound) opt->lower_bound = alt;
	opt->eta_lower_bound = opt->lower_bound;
	opt->p_lower_bound = opt->lower_bound;
	/* multiclust.c:818-851 */
	if (opt->accel_scheme >= MC_QN) {
		opt->adjust_step = 0;
		opt->q = opt->accel_scheme - MC_SQS3;
		if (opt->q > MCHIP_MAX_SECANTS) {
			fprintf(stderr, "ERROR [mc_em.c::mc_synchronize]: Cannot use acceleration methods greater than 6 (QN3) "
				"without linking to lapack.\n");
			return 1;
		}
	}
	return 0;
}

int mc_no_parameters(const mc_options *opt, const mc_data *dat, int K)
{
	int n = (!opt->admixture || opt->eta_constrained) ? (K - 1) : dat->I * (K - 1);
	for (int l = 0; l < dat->L; l++) n += (dat->uniquealleles[l] - 1) * K;
	return n;
}
double mc_aic(double max_logL, int no_parameters) { return -2 * max_logL + 2 * no_parameters; }
double mc_bic(double max_logL, int no_parameters, int I) { return -2 * max_logL + no_parameters * log((double)I); }

/* ------------------------------------------------------------------ model */
static int dev_fail(mc_model *mod, int rc, const char *what)
{
	if (rc) {
		fprintf(stderr, "ERROR [mc_em.c::%s]: device call failed (%d): %s\n", what, rc,
			mod && mod->dev ? mchip_last_error(mod->dev) : "no context");
		if (mod) { mod->fatal = MC_FATAL_DEVICE; mod->stopped = 1; }
	}
	return rc;
}

void mc_reset_model_state(mc_model *mod)
{
	mod->n_iter = 0;
	mod->logL = -INFINITY;		/* rnd_init.c:59 */
	mod->converged = 0;
	mod->stopped = 0;
	mod->iter_stop = 0;
	mod->time_stop = 0;
	mod->accel_step = 0;
	mod->fatal = 0;
	mod->pindex = mod->findex = mod->tindex = 0;
	mod->delta_index = 0;
	mod->start = clock();
	mod->seconds_run = 0;
}

static int model_create(mc_model **out, const mc_options *opt, const mc_data *dat, int K, int device, const mc_simulation *sim,
			const mc_model *like)
{
	mc_model *mod = calloc(1, sizeof *mod);
	int rc;
	*out = NULL;
	if (!mod) return MCHIP_ERR_ALLOC;
	mod->K = K;
	if ((rc = mchip_create(&mod->dev, device))) {
		fprintf(stderr, "ERROR [mc_em.c::mc_model_create]: cannot create a HIP context on device %d (status %d); "
			"this build has no CPU path\n", device, rc);
		free(mod);
		return rc;
	}
	mod->owns_dev = 1;
	if (like)
		rc = mchip_copy_genotypes(mod->dev, like->dev);
	else if (sim)
		rc = mchip_simulate_genotypes(mod->dev, dat->I, dat->L, dat->ploidy, dat->uniquealleles, sim->window, sim->K,
					      opt->eta_constrained, sim->q, sim->p);
	else
		rc = mchip_set_genotypes(mod->dev, dat->I, dat->L, dat->ploidy, dat->uniquealleles, dat->geno);
	/* fits to a bootstrap data set initialise from the observed haplotypes (rnd_init.c:471): dat->geno when the data set
	 * was generated on the device, dat->init_geno when the caller uploaded a replicate as dat->geno */
	if (!rc && opt->admixture && (sim || like || dat->init_geno))
		rc = mchip_set_init_genotypes(mod->dev, (sim || like) ? dat->geno : dat->init_geno);
	if (rc || (rc = mchip_set_model(mod->dev, K, opt->admixture, opt->eta_constrained, opt->do_projection,
					opt->eta_lower_bound, opt->p_lower_bound, opt->accel_scheme ? opt->q : 0))) {
		fprintf(stderr, "ERROR [mc_em.c::mc_model_create]: %s\n", mchip_last_error(mod->dev));
		mchip_destroy(mod->dev);
		free(mod);
		return rc;
	}
	mc_reset_model_state(mod);
	*out = mod;
	return 0;
}

int mc_model_create(mc_model **out, const mc_options *opt, const mc_data *dat, int K, int device)
{
	return model_create(out, opt, dat, K, device, NULL, NULL);
}

int mc_model_create_simulated(mc_model **out, const mc_options *opt, const mc_data *dat, int K, int device, const mc_simulation *sim)
{
	if (!sim || !opt->admixture) return MCHIP_ERR_INVALID;	/* the mixture model's replicate is drawn on the host */
	return model_create(out, opt, dat, K, device, sim, NULL);
}

/* The second model of a bootstrap replicate: fitted to the data set `like` already holds (generated once per replicate,
 * as parametric_bootstrap() is called once, multiclust.c:690); *out == NULL creates the model, otherwise it is re-used. */
int mc_model_share_simulated(mc_model **out, const mc_options *opt, const mc_data *dat, int K, int device, const mc_model *like)
{
	int rc;
	if (!like || !opt->admixture) return MCHIP_ERR_INVALID;
	if (!*out) return model_create(out, opt, dat, K, device, NULL, like);
	rc = mchip_copy_genotypes((*out)->dev, like->dev);
	if (!rc) rc = mchip_set_model((*out)->dev, (*out)->K, opt->admixture, opt->eta_constrained, opt->do_projection,
				      opt->eta_lower_bound, opt->p_lower_bound, opt->accel_scheme ? opt->q : 0);
	if (rc) fprintf(stderr, "ERROR [mc_em.c::mc_model_share_simulated]: %s\n", mchip_last_error((*out)->dev));
	mc_reset_model_state(*out);
	return rc;
}

/* The next bootstrap replicate into a model that already holds one of the same observed data set (same K, same options):
 * the device keeps every buffer and the observed haplotypes the initialisation reads; only the data set is generated anew. */
int mc_model_resimulate(mc_model *mod, const mc_options *opt, const mc_data *dat, const mc_simulation *sim)
{
	int rc;
	if (!mod || !sim || !opt->admixture) return MCHIP_ERR_INVALID;
	rc = mchip_simulate_genotypes(mod->dev, dat->I, dat->L, dat->ploidy, dat->uniquealleles, sim->window, sim->K,
				      opt->eta_constrained, sim->q, sim->p);
	if (!rc) rc = mchip_set_model(mod->dev, mod->K, opt->admixture, opt->eta_constrained, opt->do_projection,
				      opt->eta_lower_bound, opt->p_lower_bound, opt->accel_scheme ? opt->q : 0);
	if (rc) fprintf(stderr, "ERROR [mc_em.c::mc_model_resimulate]: %s\n", mchip_last_error(mod->dev));
	mc_reset_model_state(mod);
	return rc;
}

int mc_model_get_genotypes(mc_model *mod, uint8_t *geno) { return mchip_get_genotypes(mod->dev, geno); }

/* ------------------------------------------------------------------ parametric bootstrap (bootstrap.c:76-175)
 * In the default build every (i, l) receives `ploidy` simulated copies: the "missing stays missing" count copied at
 * bootstrap.c:87 is zeroed again by the loop that follows it (m_start = 0) before n_end is taken.  Admixture: two
 * rand() per copy (source cluster, then allele); mixture: one per individual, then one per copy. */
static int cdf_walk(const double *w, int n, double r)
{
	int j = 0;
	double sum = 0;
	while (j < n && r > sum) sum += w[j++];
	return j ? j - 1 : 0;
}

void mc_bootstrap_genotypes(const mc_options *opt, const mc_data *dat, int K, const double *q, const double *p,
			    mc_rng *rng, uint8_t *geno)
{
	const int indiv = opt->admixture && !opt->eta_constrained;
	int T = 0;
	int *toff = malloc(sizeof(int) * ((size_t)dat->L + 1));
	if (!toff) return;
	for (int l = 0; l < dat->L; l++) { toff[l] = T; T += dat->uniquealleles[l]; }
	toff[dat->L] = T;
	for (int i = 0; i < dat->I; i++) {
		int kmix = 0;
		if (!opt->admixture) kmix = cdf_walk(q, K, (double)mc_rand(rng) / 2147483647.0);
		for (int l = 0; l < dat->L; l++)
			for (int n = 0; n < dat->ploidy; n++) {
				int j = kmix;
				if (opt->admixture) j = cdf_walk(indiv ? q + (size_t)i * K : q, K, (double)mc_rand(rng) / 2147483647.0);
				geno[((size_t)i * dat->L + l) * dat->ploidy + n] =
					(uint8_t)cdf_walk(p + (size_t)j * T + toff[l], dat->uniquealleles[l], (double)mc_rand(rng) / 2147483647.0);
			}
	}
	free(toff);
}

uint64_t mc_bootstrap_draws(const mc_options *opt, const mc_data *dat)
{
	const uint64_t copies = (uint64_t)dat->I * dat->L * dat->ploidy;
	return opt->admixture ? 2 * copies : (uint64_t)dat->I + copies;
}

void mc_simulation_begin(mc_simulation *sim, const mc_options *opt, const mc_data *dat, int K, const double *q,
			 const double *p, mc_rng *rng)
{
	for (int t = 0; t < 31; t++) sim->window[t] = (uint32_t)rng->r[(rng->f + t) % 31];
	sim->K = K; sim->q = q; sim->p = p;
	mc_rng_jump(rng, mc_bootstrap_draws(opt, dat));
}

void mc_model_free(mc_model *mod)
{
	if (!mod) return;
	mc_init_cache_free(mod);
	if (mod->owns_dev && mod->dev) mchip_destroy(mod->dev);
	free(mod);
}

const char *mc_model_error(const mc_model *mod) { return mod && mod->dev ? mchip_last_error(mod->dev) : ""; }
int mc_model_set_p(mc_model *mod, int slot, const double *p) { return mchip_set_p(mod->dev, slot, p); }
int mc_model_get_p(mc_model *mod, int slot, double *p) { return mchip_get_p(mod->dev, slot, p); }
int mc_model_set_q(mc_model *mod, int slot, const double *q) { return mchip_set_q(mod->dev, slot, q); }
int mc_model_get_q(mc_model *mod, int slot, double *q) { return mchip_get_q(mod->dev, slot, q); }
int mc_model_get_expected_counts(mc_model *mod, double *sik) { return mchip_get_expected_counts(mod->dev, sik); }

/* assign[j] = rand() % K for j = 0..n-1 in stream order (rnd_init.c:467).  Large draws are split over host threads:
 * thread t starts from the stream jumped ahead to its first draw (mc_rng_jump), so the result is the serial one. */
typedef struct draw_job { uint8_t *out; size_t n; int K; mc_rng rng; } draw_job;

static void *draw_main(void *arg)
{
	draw_job *j = arg;
	for (size_t x = 0; x < j->n; x++) j->out[x] = (uint8_t)(mc_rand(&j->rng) % j->K);
	return NULL;
}

static void draw_partition(uint8_t *assign, size_t n, int K, mc_rng *rng)
{
	int nt = 1;
	if (n >= ((size_t)1 << 24)) {
		long cpus = sysconf(_SC_NPROCESSORS_ONLN);
		nt = cpus > 16 ? 16 : (cpus < 1 ? 1 : (int)cpus);
		if (getenv("MC_INIT_THREADS")) nt = atoi(getenv("MC_INIT_THREADS")) > 0 ? atoi(getenv("MC_INIT_THREADS")) : nt;
	}
	if (nt <= 1) {
		for (size_t j = 0; j < n; j++) assign[j] = (uint8_t)(mc_rand(rng) % K);
		return;
	}
	draw_job jobs[64];
	pthread_t th[64];
	int joinable[64] = { 0 };	/* (a pthread_t has no "none" value) */
	if (nt > 64) nt = 64;
	const size_t per = (n + (size_t)nt - 1) / (size_t)nt;
	for (int t = 0; t < nt; t++) {
		const size_t lo = (size_t)t * per, hi = lo + per < n ? lo + per : n;
		jobs[t].out = assign + lo;
		jobs[t].n = lo < n ? hi - lo : 0;
		jobs[t].K = K;
		jobs[t].rng = *rng;
		mc_rng_jump(&jobs[t].rng, (uint64_t)lo);
		if (pthread_create(&th[t], NULL, draw_main, &jobs[t])) draw_main(&jobs[t]);	/* no thread to be had: in this one */
		else joinable[t] = 1;
	}
	for (int t = 0; t < nt; t++) if (joinable[t]) pthread_join(th[t], NULL);
	mc_rng_jump(rng, (uint64_t)n);		/* the caller's stream ends where the serial loop would */
}

/* test hook for tests/test_shard_cpu.py */
void mc_test_draw_partition(uint8_t *assign, size_t n, int K, mc_rng *rng) { draw_partition(assign, n, K, rng); }

int mc_initialize_model(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng)
{
	/* rnd_init.c:54-89 reset, then random_initialize_admixture (349-357): one rand() % K per allele copy in
	 * i, l, a order (456-467), first M step on the device */
	size_t n = (size_t)dat->I * dat->L * dat->ploidy;
	uint8_t *assign;
	int rc;
	mod->n_iter = 0;
	mod->logL = -INFINITY;
	mod->converged = 0;
	if (opt->accel_scheme) mod->pindex = mod->tindex = mod->findex = 0;
	if (opt->initialization_procedure == MC_RAND_EM)	/* rnd_init.c:80,85: opt-in, the reference's own -m never selects it */
		return dev_fail(mod, mc_randem_initialize(opt, dat, mod, rng), "mc_initialize_model");
	if (!opt->admixture)
		return dev_fail(mod, mc_initialize_mixture(dat, mod, rng), "mc_initialize_model");
	if (!getenv("MC_HOST_INIT")) {
		/* the partition is drawn on the device from this point of the stream; the host copy of the stream moves on
		 * by the same number of draws */
		uint32_t window[31];
		for (int t = 0; t < 31; t++) window[t] = (uint32_t)rng->r[(rng->f + t) % 31];
		rc = mchip_mstep_from_rand_partition(mod->dev, window, mod->tindex);
		mc_rng_jump(rng, (uint64_t)n);
		return dev_fail(mod, rc, "mc_initialize_model");
	}
	if (!(assign = malloc(n))) return MCHIP_ERR_ALLOC;
	draw_partition(assign, n, mod->K, rng);
	rc = mchip_mstep_from_partition(mod->dev, assign, mod->tindex);
	free(assign);
	return dev_fail(mod, rc, "mc_initialize_model");
}

/* ------------------------------------------------------------------ stopping rules */
int mc_converged(const mc_options *opt, mc_model *mod, double loglik)
{
	/* em_alg.c:163-182 */
	int stop = 1;
	double abs_diff = 0, rel_diff = 0;
	if (opt->abs_error) abs_diff = fabs(loglik - mod->logL);
	if (opt->rel_error) rel_diff = abs_diff / fabs(mod->logL);
	if (opt->abs_error && abs_diff > opt->abs_error) stop &= 0;
	if (opt->rel_error && rel_diff > opt->rel_error) stop &= 0;
	if (stop) mod->converged = 1;
	return stop;
}

static int stop_condition(const mc_options *opt, mc_model *mod, double loglik)
{
	/* em_alg.c:145-161 */
	mod->seconds_run = ((double)clock() - mod->start) / CLOCKS_PER_SEC;
	if (opt->max_iter && mod->n_iter > opt->max_iter) {
		mod->iter_stop = 1;
		return 1;
	}
	if (opt->n_seconds && mod->seconds_run > opt->n_seconds) {
		mod->time_stop = 1;
		return 1;
	}
	return mc_converged(opt, mod, loglik);
}

int mc_stop(const mc_options *opt, mc_model *mod, double loglik)
{
	/* em_alg.c:101-143 */
	mod->n_iter++;
	if (isnan(loglik)) {
		fprintf(stderr, "ERROR [em_alg.c::stop(107)]: nan\n");
		mod->fatal = MC_FATAL_NAN;
		mod->stopped = 1;
		return 1;
	}
	mod->stopped = stop_condition(opt, mod, loglik);
	if (loglik < mod->logL && !mod->stopped) {
		fprintf(stderr, "ERROR [em_alg.c::stop(116)]: log likelihood decrease (%f < %f; %e)\n",
			loglik, mod->logL, (loglik - mod->logL) / loglik);
		mod->fatal = MC_FATAL_DECREASE;
		mod->stopped = 1;
		return 1;
	}
	if (opt->verbosity > MC_MINIMAL) {
		fprintf(stderr, "%4d (%s", mod->n_iter,
			mod->accel_step ? accel_abbrev[opt->accel_scheme < MC_QN ? opt->accel_scheme : MC_QN] : "EM");
		if (mod->accel_step && opt->accel_scheme > MC_QN) fprintf(stderr, "%d", opt->q);
		fprintf(stderr, "): %.2f (delta): %.5g\n", loglik, loglik - mod->logL);
	}
	mod->accel_step = 0;
	mod->logL = loglik;
	return mod->stopped;
}

/* ------------------------------------------------------------------ EM steps */
int mc_em_step(const mc_options *opt, const mc_data *dat, mc_model *mod)
{
	/* em_alg.c:195-207 */
	double ll = NAN;
	(void)dat;
	if (dev_fail(mod, mchip_em_step(mod->dev, mod->findex, mod->tindex, &ll), "mc_em_step")) return 1;
	return mc_stop(opt, mod, ll);
}

double mc_em_e_step(const mc_options *opt, const mc_data *dat, mc_model *mod)
{
	/* em_alg.c:219-233: E, M, E; returns the log likelihood after the step */
	double ll = NAN;
	(void)opt; (void)dat;
	if (dev_fail(mod, mchip_em_step(mod->dev, mod->findex, mod->tindex, NULL), "mc_em_e_step")) return NAN;
	if (dev_fail(mod, mchip_e_step(mod->dev, mod->tindex, &ll), "mc_em_e_step")) return NAN;
	return ll;
}

double mc_log_likelihood(const mc_options *opt, const mc_data *dat, mc_model *mod, int which)
{
	double ll = NAN;
	(void)opt; (void)dat;
	if (dev_fail(mod, mchip_loglik(mod->dev, which, &ll), "mc_log_likelihood")) return NAN;
	return ll;
}

int mc_em_2_steps(mc_model *mod, const mc_data *dat, const mc_options *opt)
{
	/* em_alg.c:1072-1211 */
	mod->findex = mod->pindex;
	mod->tindex = (mod->findex + 1) % 3;
	for (int j = 0; j < 2; j++) {
		if (mc_em_step(opt, dat, mod)) return 1;
		if (dev_fail(mod, mchip_secant(mod->dev, j, mod->delta_index, mod->tindex, mod->findex), "mc_em_2_steps")) return 1;
		mod->findex = mod->tindex;
		mod->tindex = (mod->findex + 1) % 3;
		if (mod->tindex == mod->pindex) mod->tindex = (mod->tindex + 1) % 3;
	}
	mod->delta_index = (mod->delta_index + 1) % opt->q;
	return 0;
}

/* An individual without a single observed allele copy has mixing proportions 0 / 0 = NaN in the reference from its first M step
 * on (em_alg.c:685-690).  Its secants are NaN, so step_size() is NaN and accelerated_em_step() takes its "invalid step size:
 * fall back on EM step" exit every cycle (accel_em.c:58-62); with q > 1 qn_accelerated_update() moves to a NaN point whose log
 * likelihood is NaN and refused (accel_em.c:97).  The device holds a finite row for such an individual (mchip_get_q reports it
 * as NaN), so the NaN has to be supplied here. */
static int carries_nan_rows(const mc_options *opt, mc_model *mod)
{
	return opt->admixture && !opt->eta_constrained && mchip_empty_individuals(mod->dev, NULL) > 0;
}

double mc_step_size(const mc_options *opt, const mc_data *dat, mc_model *mod)
{
	/* accel_em.c:130-243 */
	double d[3] = { NAN, NAN, NAN }, s;
	(void)dat;
	if (carries_nan_rows(opt, mod)) return NAN;	/* u'u, u'(v-u), (v-u)'(v-u) all hold NaN terms in the reference: "invalid step size" */
	if (dev_fail(mod, mchip_step_dots(mod->dev, mod->delta_index, d), "mc_step_size")) return NAN;
	const double utu = d[0], utvu = d[1], vutvu = d[2];
	if (opt->accel_scheme == MC_SQS1) s = utu / utvu;
	else if (opt->accel_scheme == MC_SQS2) s = utvu / vutvu;
	else if (opt->accel_scheme == MC_SQS3) {
		if (sqrt(utu) < 1e-8) return NAN;
		s = -sqrt(utu / vutvu);
	} else if (opt->accel_scheme == MC_QN) s = -utu / utvu;
	else s = -1;
	if (opt->accel_scheme < MC_QN && s > -1) s = -1;
	return s;
}

double mc_accelerated_update(const mc_options *opt, const mc_data *dat, mc_model *mod, double s)
{
	/* accel_em.c:422-551 */
	double ll;
	(void)dat;
	mod->delta_index = mod->delta_index ? mod->delta_index - 1 : opt->q - 1;
	if (dev_fail(mod, mchip_accel_update(mod->dev, mod->tindex, mod->pindex, mod->delta_index, s,
					     opt->accel_scheme == MC_QN), "mc_accelerated_update")) return NAN;
	/* log_likelihood(tindex), accel_em.c:544; if this point is accepted the next cycle's first E step reads the same
	 * slot, so the device keeps the per-individual sums of this pass for it */
	ll = NAN;
	if (dev_fail(mod, mchip_loglik_prefetch(mod->dev, mod->tindex, &ll), "mc_accelerated_update")) return NAN;
	mod->delta_index = (mod->delta_index + 1) % opt->q;
	return ll;
}

double mc_qn_accelerated_update(const mc_options *opt, const mc_data *dat, mc_model *mod)
{
	/* accel_em.c:262-419, q = 2 or 3 */
	const int q = opt->q;
	int vindex = mod->delta_index ? mod->delta_index - 1 : q - 1;
	int uindex = vindex ? vindex - 1 : q - 1;
	int q1, q2, j, n, nt = 0;
	int v_index[9];
	double ca[9], cb[9], det, d2[2];
	double *A = mod->A, *Ainv = mod->Ainv;
	if (carries_nan_rows(opt, mod)) return NAN;

	q1 = mod->delta_index;
	j = 0;
	do {
		q2 = mod->delta_index;
		n = 0;
		do {
			if (dev_fail(mod, mchip_secant_dots(mod->dev, q1, q2, d2), "mc_qn_accelerated_update")) return NAN;
			mod->cutu[n] = d2[0];
			A[j * q + n] = d2[0] - d2[1];
			n++;
			q2 = (q2 + 1) % q;
		} while (q2 != mod->delta_index);
		q1 = (q1 + 1) % q;
		j++;
	} while (q1 != mod->delta_index);

	if (q == 1) {
		Ainv[0] = 1 / A[0];
	} else if (q == 2) {
		det = A[0] * A[3] - A[1] * A[2];
		Ainv[0] = A[3] / det;
		Ainv[3] = A[0] / det;
		Ainv[1] = -A[1] / det;
		Ainv[2] = -A[2] / det;
	} else {
		det = A[0] * (A[4] * A[8] - A[5] * A[7])
			- A[1] * (A[8] * A[3] - A[5] * A[6])
			+ A[2] * (A[3] * A[7] - A[4] * A[6]);
		Ainv[0] = (A[4] * A[8] - A[5] * A[7]) / det;
		Ainv[1] = (A[2] * A[7] - A[1] * A[8]) / det;
		Ainv[2] = (A[1] * A[5] - A[2] * A[4]) / det;
		Ainv[3] = (A[5] * A[6] - A[3] * A[8]) / det;
		Ainv[4] = (A[0] * A[8] - A[2] * A[6]) / det;
		Ainv[5] = (A[2] * A[3] - A[0] * A[5]) / det;
		Ainv[6] = (A[3] * A[7] - A[4] * A[6]) / det;
		Ainv[7] = (A[1] * A[6] - A[0] * A[7]) / det;
		Ainv[8] = (A[0] * A[4] - A[1] * A[3]) / det;
	}
	q1 = mod->delta_index;
	j = 0;
	do {
		n = 0;
		q2 = mod->delta_index;
		do {
			v_index[nt] = q1;
			ca[nt] = Ainv[j * q + n];
			cb[nt] = mod->cutu[n];
			nt++;
			q2 = (q2 + 1) % q;
			n++;
		} while (q2 != mod->delta_index);
		q1 = (q1 + 1) % q;
		j++;
	} while (q1 != mod->delta_index);
	if (dev_fail(mod, mchip_multisecant_update(mod->dev, mod->tindex, mod->pindex, uindex, nt, v_index, ca, cb),
		     "mc_qn_accelerated_update")) return NAN;
	return mc_log_likelihood(opt, dat, mod, mod->tindex);
}

int mc_accelerated_em_step(const mc_options *opt, const mc_data *dat, mc_model *mod)
{
	/* accel_em.c:35-114 */
	int n_adjust = 0;
	double emll, ll = 0, s = 0;

	mc_em_2_steps(mod, dat, opt);
	if (mod->stopped) return 1;
	emll = mc_log_likelihood(opt, dat, mod, mod->findex);
	if (mod->fatal) return 1;
	mod->last_emll = emll; mod->last_step = 0; mod->last_ll = 0; mod->last_accepted = 0;
	if (opt->accel_scheme <= MC_QN) {
		s = mc_step_size(opt, dat, mod);
		if (mod->fatal) return 1;
		mod->last_step = s;
		if (isnan(s) || isinf(s)) goto EM_EXIT;
	}
	do {
		if (opt->accel_scheme <= MC_QN)
			ll = mc_accelerated_update(opt, dat, mod, s);
		else
			ll = mc_qn_accelerated_update(opt, dat, mod);
		if (mod->fatal) return 1;
		if (opt->adjust_step && ll < emll) {
			if (opt->verbosity > MC_MINIMAL)
				fprintf(stderr, "after attempt %d (of %d) of accel ll is %f, EM ll is %f, step is %f\n",
					n_adjust, opt->adjust_step, ll, emll, s);
			s = (s - 1) / 2;
		}
	} while (n_adjust++ < opt->adjust_step && ll < emll && s < -1);
	mod->last_step = s; mod->last_ll = ll;
	if (opt->verbosity > MC_TALKATIVE)
		fprintf(stderr, "accelerated_em_step (%s): aEM %f %s EM %f (step size: %f)\n",
			ll > emll ? accel_abbrev[opt->accel_scheme < MC_QN ? opt->accel_scheme : MC_QN] : "EM",
			ll, ll > emll ? ">" : "<", emll, s);
	if (ll > emll) {
		mod->pindex = mod->tindex;
		mod->accel_step = 1;
		mod->last_accepted = 1;
		return 0;
	}
EM_EXIT:
	mod->pindex = mod->findex;
	return 0;
}

/* Unaccelerated loop of em() (em_alg.c:61-64,78-88) in batches whose stopping rule runs on the device
 * (mchip_em_run): same arithmetic, same stopping iteration, no host round trip per iteration.  Used when nothing has to
 * be printed per iteration and no wall-clock limit applies; returns -1 when the device path does not cover the model. */
#define MC_EM_BATCH 32
static int em_batched(const mc_options *opt, mc_model *mod)
{
	while (!mod->stopped) {
		mchip_run_state st;
		memset(&st, 0, sizeof st);
		st.logL = mod->logL;
		st.abs_error = opt->abs_error;
		st.rel_error = opt->rel_error;
		st.n_iter = mod->n_iter;
		st.max_iter = opt->max_iter;
		const int rc = mchip_em_run(mod->dev, mod->findex, MC_EM_BATCH, &st);
		if (rc == MCHIP_ERR_UNSUPPORTED) return -1;
		if (dev_fail(mod, rc, "mc_em")) return 0;
		mod->n_iter = st.n_iter;
		mod->logL = st.logL;
		mod->accel_step = 0;
		if (st.converged) mod->converged = 1;
		if (st.iter_stop) mod->iter_stop = 1;
		mod->stopped = st.stopped;
		mod->seconds_run = ((double)clock() - mod->start) / CLOCKS_PER_SEC;
		if (st.fatal == 1) {
			fprintf(stderr, "ERROR [em_alg.c::stop(107)]: nan\n");
			mod->fatal = MC_FATAL_NAN;
		} else if (st.fatal == 2) {
			fprintf(stderr, "ERROR [em_alg.c::stop(116)]: log likelihood decrease (%f < %f; %e)\n",
				st.bad_loglik, st.logL, (st.bad_loglik - st.logL) / st.bad_loglik);
			mod->fatal = MC_FATAL_DECREASE;
		}
	}
	return 0;
}

/* The accelerated loop of em() (em_alg.c:84-88) in batches of cycles that run on the device without a host round trip
 * (mchip_accel_run): one secant pair, no back-tracking, nothing printed per iteration, no wall-clock limit.  Returns -1
 * when the device path does not cover the model (the caller then drives the cycles one by one). */
#define MC_ACCEL_BATCH 8
static int em_accel_batched(const mc_options *opt, mc_model *mod)
{
	const int scheme = opt->accel_scheme;		/* 1..3 SQUAREM, 4 = QN with q = 1 */
	if (carries_nan_rows(opt, mod)) return -1;	/* every cycle falls back to its EM iterate: the host loop does that */
	while (!mod->stopped) {
		mchip_run_state st;
		memset(&st, 0, sizeof st);
		st.logL = mod->logL;
		st.abs_error = opt->abs_error;
		st.rel_error = opt->rel_error;
		st.n_iter = mod->n_iter;
		st.max_iter = opt->max_iter;
		const int rc = mchip_accel_run(mod->dev, mod->pindex, scheme, MC_ACCEL_BATCH, &st);
		if (rc == MCHIP_ERR_UNSUPPORTED) return -1;
		if (dev_fail(mod, rc, "mc_em")) return 0;
		mod->n_iter = st.n_iter;
		mod->logL = st.logL;
		mod->findex = mod->tindex = mod->pindex;	/* the reported iterate; the other two slots are scratch */
		if (st.converged) mod->converged = 1;
		if (st.iter_stop) mod->iter_stop = 1;
		if (st.fatal == 1) {
			fprintf(stderr, "ERROR [em_alg.c::stop(107)]: nan\n");
			mod->fatal = MC_FATAL_NAN;
		} else if (st.fatal == 2) {
			fprintf(stderr, "ERROR [em_alg.c::stop(116)]: log likelihood decrease (%f < %f; %e)\n",
				st.bad_loglik, st.logL, (st.bad_loglik - st.logL) / st.bad_loglik);
			mod->fatal = MC_FATAL_DECREASE;
		}
		if (st.stopped) mod->stopped = 1;
	}
	mod->seconds_run = ((double)clock() - mod->start) / CLOCKS_PER_SEC;
	return 0;
}

void mc_em(const mc_options *opt, const mc_data *dat, mc_model *mod)
{
	/* em_alg.c:44-90 */
	int stop = 0;
	if (mod->K == 1) {
		mc_em_step(opt, dat, mod);
		if (!mod->fatal) mod->logL = mc_log_likelihood(opt, dat, mod, mod->tindex);
		return;
	}
	/* (with -i the warm-up loop and the do/while are two loops: when the first stops on the -T cap the second still runs
	 * one more step, em_alg.c:61-88 -- the batched loop is a single loop, so it serves n_init_iter = 0 only) */
	if (!opt->accel_scheme && opt->n_init_iter <= 0 && opt->verbosity <= MC_MINIMAL && !opt->n_seconds &&
	    !getenv("MC_NO_BATCH") && em_batched(opt, mod) == 0)
		return;
	while (mod->n_iter < opt->n_init_iter && !stop)
		stop = mc_em_step(opt, dat, mod);
	for (int i = 1; i < opt->q; i++) {
		mc_em_2_steps(mod, dat, opt);
		mod->pindex = mod->findex;
	}
	if (mod->converged || mod->fatal) return;
	if (opt->accel_scheme >= MC_SQS1 && opt->accel_scheme <= MC_QN && opt->q == 1 && !opt->adjust_step && !stop &&
	    opt->verbosity <= MC_MINIMAL && !opt->n_seconds &&
	    !getenv("MC_NO_BATCH") && em_accel_batched(opt, mod) == 0)
		return;
	do {
		if (!opt->accel_scheme)
			stop = mc_em_step(opt, dat, mod);
		else
			stop = mc_accelerated_em_step(opt, dat, mod);
	} while (!stop);
}
