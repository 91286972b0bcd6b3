/*
 * mc_fit.c -- several initialisations of one model: the unit of multi-GPU sharding.
 * Restates the bookkeeping of maximize_likelihood (reference multiclust.c:471-656) so that per-unit results
 * produced on different GPUs, fed back in unit order, reproduce the serial program's summary; and the
 * jump-ahead of the libc-compatible rand() stream that lets unit u start where the serial program would.
 */
#include "mc_host.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static double now_ms(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6;
}

/* ------------------------------------------------------------------ rand() jump-ahead */
static void polymul_mod(const uint32_t *a, const uint32_t *b, uint32_t *out)
{
	/* (a * b) mod (x^31 - x^28 - 1) over Z/2^32 */
	uint32_t t[61];
	memset(t, 0, sizeof t);
	for (int i = 0; i < 31; i++) {
		if (!a[i]) continue;
		for (int j = 0; j < 31; j++) t[i + j] += a[i] * b[j];
	}
	for (int d = 60; d >= 31; d--) {
		t[d - 3] += t[d];	/* x^d = x^(d-3) + x^(d-31) */
		t[d - 31] += t[d];
	}
	memcpy(out, t, 31 * sizeof(uint32_t));
}

void mc_rng_jump(mc_rng *g, uint64_t n)
{
	uint32_t c[31], base[31], s[61], nw[31];
	if (!n) return;
	memset(c, 0, sizeof c);
	memset(base, 0, sizeof base);
	c[0] = 1;	/* x^0 */
	base[1] = 1;	/* x^1 */
	for (uint64_t e = n; e; e >>= 1) {
		if (e & 1) polymul_mod(c, base, c);
		polymul_mod(base, base, base);
	}
	/* s[t] = x_{j+t}, oldest first: slot f holds the value written 31 draws ago */
	for (int t = 0; t < 31; t++) s[t] = (uint32_t)g->r[(g->f + t) % 31];
	for (int t = 0; t < 30; t++) s[31 + t] = s[t] + s[28 + t];
	for (int t = 0; t < 31; t++) {
		uint32_t v = 0;
		for (int j = 0; j < 31; j++) v += c[j] * s[t + j];
		nw[t] = v;
	}
	for (int t = 0; t < 31; t++) g->r[(g->f + t) % 31] = (int32_t)nw[t];
}

/* ------------------------------------------------------------------ bookkeeping */
void mc_summary_reset(mc_summary *s)
{
	s->first_max_logL = -INFINITY;
	s->max_logL = -INFINITY;	/* estimate_model, multiclust.c:377 */
	s->n_init = 0;
	s->n_total_iter = 0;
	s->n_maxll_times = 0;
	s->n_maxll_init = -1;
	s->n_max_iter = 0;
	s->ever_converged = 0;
	s->best_unit = -1;
	s->aic = s->bic = INFINITY;
}

void mc_summary_add(const mc_options *opt, mc_summary *s, const mc_unit_result *r, int no_parameters, int I)
{
	if (r->converged) s->ever_converged = 1;
	if (r->converged || (!s->n_init && r->time_stop)) {
		s->n_total_iter += r->n_iter;
		if (s->n_max_iter < r->n_iter) s->n_max_iter = r->n_iter;
		s->n_init++;
	}
	/* converged(opt, mod, mod->first_max_logL), em_alg.c:163-182, with mod->logL = r->logL */
	int seen = 1;
	double abs_diff = 0, rel_diff = 0;
	if (opt->abs_error) abs_diff = fabs(s->first_max_logL - r->logL);
	if (opt->rel_error) rel_diff = abs_diff / fabs(r->logL);
	if (opt->abs_error && abs_diff > opt->abs_error) seen = 0;
	if (opt->rel_error && rel_diff > opt->rel_error) seen = 0;
	if (r->converged && seen) {
		s->n_maxll_times++;
	} else if (r->converged && r->logL > s->first_max_logL) {
		s->n_maxll_times = 1;
		s->first_max_logL = r->logL;
		s->n_maxll_init = s->n_init;
	}
	if (r->logL > s->max_logL) {	/* strict: ties keep the earlier unit */
		s->max_logL = r->logL;
		s->aic = mc_aic(s->max_logL, no_parameters);
		s->bic = mc_bic(s->max_logL, no_parameters, I);
		s->best_unit = r->unit;
	}
}

uint64_t mc_draws_per_init(const mc_options *opt, const mc_data *dat, int K)
{
	(void)opt; (void)K;
	return (uint64_t)dat->I * dat->L * dat->ploidy;
}

int mc_fit_unit(const mc_options *opt, const mc_data *dat, mc_model *mod, unsigned int seed, int unit, mc_unit_result *out)
{
	mc_rng rng;
	int rc;
	const int delta_keep = mod->delta_index;	/* not reset between initialisations (multiclust.c:518-524) */
	mc_srand(&rng, seed);
	if ((rc = mc_skip_initializations(opt, dat, mod, &rng, unit))) return rc;
	mc_reset_model_state(mod);
	mod->delta_index = delta_keep;
	if ((rc = mc_initialize_model(opt, dat, mod, &rng))) return rc;
	mc_em(opt, dat, mod);
	out->unit = unit;
	out->logL = mod->logL;
	out->converged = mod->converged;
	out->n_iter = mod->n_iter;
	out->time_stop = mod->time_stop;
	out->iter_stop = mod->iter_stop;
	out->pindex = mod->pindex;
	out->fatal = mod->fatal;
	out->seconds_run = mod->seconds_run;
	return 0;
}

/* One parametric-bootstrap replicate, the unit run_bootstrap shards (multiclust.c:675-708): data set b is generated on
 * `device` from the H0 fit (mle_K clusters, mle_q, mle_p) at the position replicate b has in the serial rand() stream
 * (`base` = the stream where the first replicate begins), then n_init initialisations of the null_K model and of the alt_K
 * model are fitted to it (only one when K = 1, multiclust.c:630), each from the stream where the serial program would be;
 * the test statistic is the difference of the two best log likelihoods. */
int mc_fit_replicate(const mc_options *opt, const mc_data *dat, int device, const mc_rng *base, int b, int null_K, int alt_K,
		     int n_init, int mle_K, const double *mle_q, const double *mle_p, mc_replicate_result *out, mc_model **models)
{
	const uint64_t per_init = mc_draws_per_init(opt, dat, alt_K);
	const uint64_t units0 = null_K == 1 ? 1 : (uint64_t)n_init, units1 = alt_K == 1 ? 1 : (uint64_t)n_init;
	const uint64_t per_replicate = mc_bootstrap_draws(opt, dat) + (units0 + units1) * per_init;
	mc_rng rng = *base;
	mc_simulation gen;
	int rc = 0;
	if (!opt->admixture) return MCHIP_ERR_UNSUPPORTED;	/* the mixture model's replicate is drawn on the host */
	if (opt->initialization_procedure == MC_RAND_EM) return MCHIP_ERR_UNSUPPORTED;	/* no closed form for a replicate's stream position */
	memset(out, 0, sizeof *out);
	out->replicate = b;
	mc_rng_jump(&rng, (uint64_t)b * per_replicate);
	mc_simulation_begin(&gen, opt, dat, mle_K, mle_q, mle_p, &rng);
	for (int h = 0; h < 2 && !rc; h++) {
		const int K = h ? alt_K : null_K;
		const uint64_t units = h ? units1 : units0;
		double best = -INFINITY;
		mc_model *mod = NULL;
		const int timing = getenv("MC_TIMING") != NULL;	/* diagnostics: where a replicate spends its time */
		double t0 = timing ? now_ms() : 0, t1, t2, t3;
		if (h && models && models[0]) {	/* the alternative model takes the data set the null model was just fitted to */
			if ((rc = mc_model_share_simulated(&models[1], opt, dat, K, device, models[0]))) break;
			mod = models[1];
		} else if (models && models[h]) {
			mod = models[h];
			if ((rc = mc_model_resimulate(mod, opt, dat, &gen))) break;
		} else {
			if ((rc = mc_model_create_simulated(&mod, opt, dat, K, device, &gen))) break;
			if (models) models[h] = mod;
		}
		t1 = timing ? now_ms() : 0;
		for (uint64_t u = 0; u < units; u++) {
			const int delta_keep = mod->delta_index;
			mc_reset_model_state(mod);
			mod->delta_index = delta_keep;
			if ((rc = mc_initialize_model(opt, dat, mod, &rng))) break;
			t2 = timing ? now_ms() : 0;
			mc_em(opt, dat, mod);
			t3 = timing ? now_ms() : 0;
			if (timing) fprintf(stderr, "replicate %d K=%d unit %d: create+generate %.1f ms, initialise %.1f ms, em (%d iterations) %.1f ms\n",
					    b, K, (int)u, t1 - t0, t2 - t1, mod->n_iter, t3 - t2);
			out->n_iter += mod->n_iter;
			if (mod->fatal) { out->fatal = mod->fatal; break; }
			if (mod->logL > best) best = mod->logL;
		}
		t2 = timing ? now_ms() : 0;
		if (!models) mc_model_free(mod);
		if (timing) fprintf(stderr, "replicate %d K=%d: model freed in %.1f ms, %.1f ms since the model was requested\n", b, K, now_ms() - t2, now_ms() - t0);
		if (h) out->logL_HA = best; else out->logL_H0 = best;
	}
	out->ts = out->logL_HA - out->logL_H0;
	return rc;
}
