/*
 * ref_glue.c -- the reference-side binding INTEGRATION.md describes, as a real translation unit.
 *
 * TEST INFRASTRUCTURE (build container only): `make -C oracle glue` compiles this file against the reference's OWN headers
 * where they lie (/root/reference/multiclust.h -- nothing is copied) and links it with libmulticlust_host.so, so that a
 * renamed field of the reference's options / data / model structs, or a changed signature on our side, breaks the build
 * instead of silently outdating the document.  It is what a maintainer of the reference would add next to multiclust.c:
 * every function names the reference call site it serves.  Nothing here is loaded by the product or the tests' GPU path.
 */
#include "multiclust.h"		/* the reference's structs: options, data, model (multiclust.h:155-360) */
#include "mc_host.h"		/* our host side (multiclust_amd/host) */

/* [INTEGRATION.md section 2] dat->IL / dat->L_alleles (read_file.c:443-663) -> allele indices, MISSING -> 0xFF */
uint8_t *mcamd_flatten_genotypes(const data *dat)
{
	uint8_t *g = malloc((size_t)dat->I * dat->L * dat->ploidy);
	if (!g) return NULL;
	for (int i = 0; i < dat->I; i++)
		for (int l = 0; l < dat->L; l++)
			for (int a = 0; a < dat->ploidy; a++) {
				int v = dat->IL[dat->ploidy * i + a][l];
				uint8_t idx = MCHIP_MISSING;
				for (int m = 0; v != MISSING && m < dat->uniquealleles[l]; m++)
					if (dat->L_alleles[l][m] == v) { idx = (uint8_t)m; break; }
				g[((size_t)i * dat->L + l) * dat->ploidy + a] = idx;
			}
	return g;
}

/* [section 3] options / data as the EM layer sees them; call after the reference's synchronize() (multiclust.c:807) */
int mcamd_options(const options *opt, const data *dat, const uint8_t *flat_geno, double user_lower_bound, mc_options *mo, mc_data *md)
{
	mc_make_options(mo);
	mo->admixture = opt->admixture;
	mo->eta_constrained = opt->eta_constrained;
	mo->do_projection = opt->do_projection;
	mo->accel_scheme = opt->accel_scheme;
	mo->n_init_iter = opt->n_init_iter;
	mo->max_iter = opt->max_iter;
	mo->n_seconds = opt->n_seconds;
	mo->adjust_step = opt->adjust_step;
	mo->abs_error = opt->abs_error;
	mo->rel_error = opt->rel_error;
	mo->lower_bound = user_lower_bound;	/* 1e-8, or --bound: mc_synchronize applies multiclust.c:812-815 itself */
	mo->verbosity = opt->verbosity;
	mo->seed = opt->seed;
	mo->n_rand_em_init = opt->n_rand_em_init;
	mo->initialization_procedure = opt->initialization_procedure == RAND_EM ? MC_RAND_EM : MC_INIT_NOTHING;
	md->I = dat->I;
	md->L = dat->L;
	md->ploidy = dat->ploidy;
	md->uniquealleles = dat->uniquealleles;	/* includes the phantom slot of loci with missing data (read_file.c:527-533) */
	md->geno = flat_geno;
	md->init_geno = NULL;
	if (mc_synchronize(mo, md)) return INTERNAL_ERROR;
	if (mo->eta_lower_bound != opt->eta_lower_bound || mo->p_lower_bound != opt->p_lower_bound) return INTERNAL_ERROR;
	return NO_ERROR;
}

/* [section 3] where allocate_model_for_k() / free_model_data() are called (multiclust.c:392,422) */
int mcamd_allocate_model_for_k(const mc_options *mo, const mc_data *md, const model *mod, int device, mc_model **mm)
{
	return mc_model_create(mm, mo, md, mod->K, device) ? INTERNAL_ERROR : NO_ERROR;	/* no GPU: fails, there is no fallback */
}

/* [section 4] initialize_model(opt, dat, mod) (multiclust.c:527): rng was seeded with mc_srand(&rng, opt->seed) where
 * parse_options calls srand() (multiclust.c:1592-1596) */
int mcamd_initialize_model(const mc_options *mo, const mc_data *md, model *mod, mc_model *mm, mc_rng *rng)
{
	mc_reset_model_state(mm);
	if (mc_initialize_model(mo, md, mm, rng)) return INTERNAL_ERROR;
	mod->n_iter = 0;
	mod->logL = -INFINITY;
	mod->converged = 0;
	return NO_ERROR;
}

/* [section 4] em(opt, dat, mod) (multiclust.c:531) and the scalars maximize_likelihood() reads afterwards */
void mcamd_em(const mc_options *mo, const mc_data *md, model *mod, mc_model *mm)
{
	mc_em(mo, md, mm);
	if (mm->fatal) exit(0);		/* the reference's own reaction to NaN / decreasing log likelihood (em_alg.c:106-120) */
	mod->logL = mm->logL;
	mod->n_iter = mm->n_iter;
	mod->converged = mm->converged;
	mod->stopped = mm->stopped;
	mod->pindex = mm->pindex;
	mod->findex = mm->findex;
	mod->tindex = mm->tindex;
	mod->time_stop = mm->time_stop;
	mod->iter_stop = mm->iter_stop;
	mod->seconds_run = mm->seconds_run;
}

/* [section 4] what the writers and the MLE copies read: mod->vpklm[pindex], mod->vetaik[pindex] / vetak[pindex]
 * (multiclust.c:565-577, write_file.c:295,325) */
int mcamd_fetch_parameters(const options *opt, const data *dat, model *mod, mc_model *mm)
{
	int T = 0, rc = NO_ERROR;
	for (int l = 0; l < dat->L; l++) T += dat->uniquealleles[l];
	const size_t nq = (opt->admixture && !opt->eta_constrained) ? (size_t)dat->I * mod->K : (size_t)mod->K;
	double *p = malloc(sizeof(double) * (size_t)mod->K * T), *q = malloc(sizeof(double) * nq);
	if (!p || !q || mc_model_get_p(mm, mm->pindex, p) || mc_model_get_q(mm, mm->pindex, q)) {
		rc = INTERNAL_ERROR;
	} else {
		for (int k = 0; k < mod->K; k++) {
			int off = 0;
			for (int l = 0; l < dat->L; l++) {
				for (int m = 0; m < dat->uniquealleles[l]; m++)
					mod->vpklm[mod->pindex][k][l][m] = p[(size_t)k * T + off + m];
				off += dat->uniquealleles[l];
			}
		}
		if (opt->admixture && !opt->eta_constrained) {
			for (int i = 0; i < dat->I; i++)
				for (int k = 0; k < mod->K; k++) mod->vetaik[mod->pindex][i][k] = q[(size_t)i * mod->K + k];
		} else {
			for (int k = 0; k < mod->K; k++) mod->vetak[mod->pindex][k] = q[k];
		}
	}
	free(p); free(q);
	return rc;
}

/* [section 4] what partition_admixture / popq_admix / indivq_admix need from mod->diklm (write_file.c:350-382,446-459,
 * 531-542): its sums over (l, m), or vik for the mixture model (write_file.c:585-603) */
int mcamd_fetch_expected_counts(const options *opt, const data *dat, model *mod, mc_model *mm, double *sik /* [I][K] */)
{
	if (mc_model_get_expected_counts(mm, sik)) return INTERNAL_ERROR;
	if (!opt->admixture)
		for (int i = 0; i < dat->I; i++)
			for (int k = 0; k < mod->K; k++) mod->vik[i][k] = sik[(size_t)i * mod->K + k];
	return NO_ERROR;
}

/* [section 4] log_likelihood(opt, dat, mod, which) and em_e_step() (rnd_init.c:145,429) */
double mcamd_log_likelihood(const mc_options *mo, const mc_data *md, mc_model *mm, int which) { return mc_log_likelihood(mo, md, mm, which); }
double mcamd_em_e_step(const mc_options *mo, const mc_data *md, mc_model *mm) { return mc_em_e_step(mo, md, mm); }

/* [section 4] parametric_bootstrap() (multiclust.c:690, bootstrap.c:31) + the replicate's allocate_model_for_k(): the data set
 * is generated on the device from the H0 MLEs (mod->mle_etaik / mle_etak, mod->mle_pKLM flattened by the caller) */
int mcamd_bootstrap_model(const mc_options *mo, const mc_data *md, int null_K, const double *mle_q, const double *mle_p,
			  int K, int device, mc_rng *rng, mc_simulation *sim, int first_model_of_replicate, mc_model **mm)
{
	if (first_model_of_replicate) mc_simulation_begin(sim, mo, md, null_K, mle_q, mle_p, rng);
	if (*mm) return mc_model_resimulate(*mm, mo, md, sim) ? INTERNAL_ERROR : NO_ERROR;
	return mc_model_create_simulated(mm, mo, md, K, device, sim) ? INTERNAL_ERROR : NO_ERROR;
}
