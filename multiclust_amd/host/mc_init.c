/*
 * mc_init.c -- initialisation procedures beyond the random allele partition: the mixture model's random centers and
 * Rand-EM for both models (reference rnd_init.c:103-160, 192-339, 412-444, 496-705).  Rand-EM is an opt-in: the
 * reference's own command line never selects it (initialization_procedure stays NOTHING, multiclust.c:935,1547-1552).
 *
 * What stays on the host is what is sequential by construction: rand() is consumed locus by locus, and how much a locus
 * consumes depends on the centers drawn for it (uniqueness retries, copies that match no center).  That walk touches
 * only a table of allele counts per locus (built once per model from the observed haplotypes); the I*L*ploidy-sized work
 * -- assigning every copy, counting, normalising, the EM iteration that scores the candidate -- runs on the device.
 */
#include "mc_host.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ mixture model: random centers */
/* random_individual_center (rnd_init.c:192-259): K distinct center individuals, every other individual joins the center
 * with the smallest L1 distance between allele-count vectors (first minimum wins; a center joins itself) */
static void mixture_centers(const mc_data *dat, int K, mc_rng *rng, int *I_K, int assign)
{
	const int I = dat->I, L = dat->L, pl = dat->ploidy;
	int center[K];
	int cnt[256];
	if (K == 1) {
		if (assign) memset(I_K, 0, sizeof(int) * (size_t)I);
		return;
	}
	for (int k = 0; k < K; k++) {		/* rejection until distinct: a clash redraws and the comparison starts over (rnd_init.c:206-218) */
		int c = mc_rand(rng) % I;
		for (int j = 0; j < k;) {
			if (center[j] == c) { c = mc_rand(rng) % I; j = 0; }
			else j++;
		}
		center[k] = c;
	}
	if (!assign) return;
	for (int i = 0; i < I; i++) {
		I_K[i] = 0;
		if (i == center[0]) continue;
		double min_diff = INFINITY;
		for (int k = 0; k < K; k++) {
			if (i == center[k]) { I_K[i] = k; break; }
			double diff = 0;
			for (int l = 0; l < L; l++) {
				const int M = dat->uniquealleles[l];
				const uint8_t *gi = dat->geno + ((size_t)i * L + l) * pl;
				const uint8_t *gc = dat->geno + ((size_t)center[k] * L + l) * pl;
				for (int m = 0; m < M; m++) cnt[m] = 0;
				for (int a = 0; a < pl; a++) {
					if (gi[a] != MCHIP_MISSING) cnt[gi[a]]++;
					if (gc[a] != MCHIP_MISSING) cnt[gc[a]]--;
				}
				for (int m = 0; m < M; m++) diff += abs(cnt[m]);
			}
			if (diff < min_diff) { I_K[i] = k; min_diff = diff; }
		}
	}
}

/* initialize_parameters_mixture (rnd_init.c:268-339).  Kept quirk: the reference re-adds every individual's counts inside
 * its loop over k (296-318), so cluster k ends with 1 + (K-k) * (its allele counts) before normalisation. */
static int mixture_parameters(const mc_data *dat, mc_model *mod, const int *I_K)
{
	const int I = dat->I, L = dat->L, pl = dat->ploidy, K = mod->K;
	int *toff = malloc(sizeof(int) * ((size_t)L + 1));
	if (!toff) return MCHIP_ERR_ALLOC;
	toff[0] = 0;
	for (int l = 0; l < L; l++) toff[l + 1] = toff[l] + dat->uniquealleles[l];
	const int T = toff[L];
	double *eta = malloc(sizeof(double) * (size_t)K), *p = malloc(sizeof(double) * (size_t)K * T);
	int rc;
	if (!eta || !p) { free(eta); free(p); free(toff); return MCHIP_ERR_ALLOC; }
	for (int k = 0; k < K; k++) eta[k] = 1;
	for (int i = 0; i < I; i++) eta[I_K[i]]++;
	for (int k = 0; k < K; k++) eta[k] /= I + K;
	for (size_t x = 0; x < (size_t)K * T; x++) p[x] = 0.0;
	for (int i = 0; i < I; i++)
		for (int l = 0; l < L; l++)
			for (int a = 0; a < pl; a++) {
				const uint8_t m = dat->geno[((size_t)i * L + l) * pl + a];
				if (m != MCHIP_MISSING) p[(size_t)I_K[i] * T + toff[l] + m] += 1.0;
			}
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++) {
			double temp = 0.0;
			for (int m = 0; m < dat->uniquealleles[l]; m++) {
				double *e = &p[(size_t)k * T + toff[l] + m];
				*e = 1.0 + (K - k) * *e;
				temp += *e;
			}
			for (int m = 0; m < dat->uniquealleles[l]; m++) p[(size_t)k * T + toff[l] + m] /= temp;
		}
	rc = mchip_set_q(mod->dev, mod->tindex, eta);
	if (!rc) rc = mchip_set_p(mod->dev, mod->tindex, p);
	free(eta); free(p); free(toff);
	return rc;
}

int mc_initialize_mixture(const mc_data *dat, mc_model *mod, mc_rng *rng)
{
	int *I_K = calloc((size_t)dat->I, sizeof(int));
	if (!I_K) return MCHIP_ERR_ALLOC;
	mixture_centers(dat, mod->K, rng, I_K, 1);
	const int rc = mixture_parameters(dat, mod, I_K);
	free(I_K);
	return rc;
}

/* ------------------------------------------------------------------ admixture model: random allele centers */
typedef struct init_cache {
	int L;
	int64_t *toff;		/* [L+1] */
	uint32_t *count;	/* [T] copies of allele m at locus l among the observed haplotypes */
} init_cache;

void mc_init_cache_free(mc_model *mod)
{
	init_cache *c = mod ? mod->init_cache : NULL;
	if (!c) return;
	free(c->toff); free(c->count); free(c);
	mod->init_cache = NULL;
}

/* the haplotypes random_allele_center reads are dat->IL: the observed ones, also while a bootstrap data set is in place */
static init_cache *get_cache(const mc_data *dat, mc_model *mod)
{
	if (mod->init_cache) return mod->init_cache;
	const uint8_t *geno = dat->init_geno ? dat->init_geno : dat->geno;
	const int I = dat->I, L = dat->L, pl = dat->ploidy;
	init_cache *c = calloc(1, sizeof *c);
	if (!c) return NULL;
	c->L = L;
	c->toff = malloc(sizeof(int64_t) * ((size_t)L + 1));
	if (!c->toff) { free(c); return NULL; }
	c->toff[0] = 0;
	for (int l = 0; l < L; l++) c->toff[l + 1] = c->toff[l] + dat->uniquealleles[l];
	c->count = calloc((size_t)c->toff[L] + 1, sizeof(uint32_t));
	if (!c->count) { free(c->toff); free(c); return NULL; }
	for (int i = 0; i < I; i++) {
		const uint8_t *row = geno + (size_t)i * L * pl;
		for (int l = 0; l < L; l++)
			for (int a = 0; a < pl; a++) {
				const uint8_t m = row[(size_t)l * pl + a];
				if (m != MCHIP_MISSING) c->count[c->toff[l] + m]++;
			}
	}
	mod->init_cache = c;
	return c;
}

static void rng_skip(mc_rng *rng, uint64_t n)
{
	if (n > 40000) { mc_rng_jump(rng, n); return; }		/* a jump costs about 3e4 multiply-adds */
	for (uint64_t x = 0; x < n; x++) (void)mc_rand(rng);
}

/* The host's share of random_allele_center (rnd_init.c:496-583) for one candidate: the center alleles of every locus
 * (520-549: all alleles when the locus has fewer than K, else K distinct ones by rejection) and, from the allele counts,
 * how many copies match none of them and so take a rand() % K of their own (575-577) -- which fixes where each locus's draws
 * lie in the stream.  centers / offsets may be NULL (skipping an initialisation: only the stream moves).  Returns the length
 * of the candidate's span of the stream. */
static uint64_t walk_allele_centers(const mc_data *dat, const init_cache *c, int K, mc_rng *rng, uint8_t *centers, uint64_t *offsets)
{
	const uint64_t copies = (uint64_t)dat->I * dat->ploidy;
	uint64_t pos = 0, pending = 0;	/* draws the copies of earlier loci consume, not yet taken off the generator: only a locus
					 * that draws its centers needs the generator in place, so the skips of the loci between two such
					 * loci (every locus when K exceeds the allele counts: config 4) are made in one step */
	int center[K];
	if (K == 1) {		/* rnd_init.c:505-510: every copy to cluster 0, nothing drawn */
		if (centers) memset(centers, 0xFF, (size_t)dat->L);
		if (offsets) memset(offsets, 0, sizeof(uint64_t) * (size_t)dat->L);
		return 0;
	}
	for (int l = 0; l < dat->L; l++) {
		const int M = dat->uniquealleles[l];
		if (M < K) {
			for (int k = 0; k < M; k++) center[k] = k;
			for (int k = M; k < K; k++) center[k] = -1;
		} else {
			rng_skip(rng, pending);
			pending = 0;
			for (int k = 0; k < K; k++) {	/* K distinct slots by rejection, as above (rnd_init.c:530-548) */
				int pick = mc_rand(rng) % M;
				pos++;
				for (int j = 0; j < k;) {
					if (center[j] == pick) { pick = mc_rand(rng) % M; pos++; j = 0; }
					else j++;
				}
				center[k] = pick;
			}
		}
		uint64_t matched = 0;
		for (int k = 0; k < K && center[k] >= 0; k++) matched += c->count[c->toff[l] + center[k]];
		if (centers) for (int k = 0; k < K; k++) centers[(size_t)l * K + k] = center[k] < 0 ? 0xFF : (uint8_t)center[k];
		if (offsets) offsets[l] = pos;
		pending += copies - matched;
		pos += copies - matched;
	}
	rng_skip(rng, pending);
	return pos;
}

/* ------------------------------------------------------------------ Rand-EM */
static int randem_admixture(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng)
{
	const int K = mod->K, n_cand = K > 1 ? opt->n_rand_em_init : 1;
	const int t = mod->tindex, keep = (t + 1) % 3, cur = (t + 2) % 3;
	init_cache *c = get_cache(dat, mod);
	uint8_t *centers = malloc((size_t)dat->L * K);
	uint64_t *offsets = malloc(sizeof(uint64_t) * (size_t)dat->L);
	double max_logL = -INFINITY;
	int rc = 0, have_best = 0;
	if (!c || !centers || !offsets) { free(centers); free(offsets); return MCHIP_ERR_ALLOC; }
	for (int i = 0; i < n_cand && !rc; i++) {
		uint32_t window[31];
		for (int x = 0; x < 31; x++) window[x] = (uint32_t)rng->r[(rng->f + x) % 31];
		const uint64_t span = walk_allele_centers(dat, c, K, rng, centers, offsets);
		if ((rc = mchip_init_from_allele_centers(mod->dev, centers, offsets, window, span, t))) break;
		if ((rc = mchip_copy_slot(mod->dev, cur, t))) break;	/* the candidate's parameters, before the EM iteration moves them */
		const double ll = mc_em_e_step(opt, dat, mod);		/* rnd_init.c:429 */
		if (mod->fatal) { rc = MCHIP_ERR_HIP; break; }
		if (ll > max_logL) {					/* rnd_init.c:431-434 keeps the partition; same parameters */
			max_logL = ll;
			have_best = 1;
			rc = mchip_copy_slot(mod->dev, keep, cur);
		}
	}
	if (!rc && have_best) rc = mchip_copy_slot(mod->dev, t, keep);	/* rnd_init.c:436-439 */
	free(centers); free(offsets);
	return rc;
}

static int randem_mixture(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng)
{
	const int n_cand = mod->K > 1 ? opt->n_rand_em_init : 1;
	int *I_K = calloc((size_t)dat->I, sizeof(int)), *best = calloc((size_t)dat->I, sizeof(int));
	double max_logL = -INFINITY;
	int rc = 0;
	if (!I_K || !best) { free(I_K); free(best); return MCHIP_ERR_ALLOC; }
	for (int i = 0; i < n_cand && !rc; i++) {		/* rnd_init.c:132-155 */
		mixture_centers(dat, mod->K, rng, I_K, 1);
		if ((rc = mixture_parameters(dat, mod, I_K))) break;
		const double ll = mc_em_e_step(opt, dat, mod);
		if (mod->fatal) { rc = MCHIP_ERR_HIP; break; }
		if (ll > max_logL) {
			max_logL = ll;
			memcpy(best, I_K, sizeof(int) * (size_t)dat->I);
		}
	}
	if (!rc) rc = mixture_parameters(dat, mod, best);	/* rnd_init.c:156-159 */
	free(I_K); free(best);
	return rc;
}

int mc_randem_initialize(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng)
{
	return opt->admixture ? randem_admixture(opt, dat, mod, rng) : randem_mixture(opt, dat, mod, rng);
}

/* Where every unit of a sharded run starts in the stream: starts[u] = the generator after u initialisations, u = 0..n_units (the
 * last entry is where the serial program stands afterwards).  One walk over the units; workers that each replayed the units
 * before theirs (mc_skip_initializations) did O(units^2) walks between them when the draws are data dependent (Rand-EM). */
int mc_unit_starts(const mc_options *opt, const mc_data *dat, mc_model *mod, const mc_rng *base, int n_units, mc_rng *starts)
{
	mc_rng rng = *base;
	for (int u = 0; u <= n_units; u++) {
		starts[u] = rng;
		if (u < n_units) {
			const int rc = mc_skip_initializations(opt, dat, mod, &rng, 1);
			if (rc) return rc;
		}
	}
	return 0;
}

int mc_skip_initializations(const mc_options *opt, const mc_data *dat, mc_model *mod, mc_rng *rng, int n)
{
	const int randem = opt->initialization_procedure == MC_RAND_EM;
	const int n_cand = randem ? (mod->K > 1 ? opt->n_rand_em_init : 1) : 1;
	if (n <= 0) return 0;
	if (opt->admixture && !randem) {	/* one rand() per allele copy, missing ones included (rnd_init.c:460-467) */
		mc_rng_jump(rng, (uint64_t)n * mc_draws_per_init(opt, dat, mod->K));
		return 0;
	}
	if (opt->admixture) {
		init_cache *c = get_cache(dat, mod);
		if (!c) return MCHIP_ERR_ALLOC;
		for (int u = 0; u < n; u++)
			for (int i = 0; i < n_cand; i++) (void)walk_allele_centers(dat, c, mod->K, rng, NULL, NULL);
		return 0;
	}
	for (int u = 0; u < n; u++)		/* mixture: only the center draws consume the stream */
		for (int i = 0; i < n_cand; i++) mixture_centers(dat, mod->K, rng, NULL, 0);
	return 0;
}
