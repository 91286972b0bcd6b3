/* ref_glue.h -- prototypes of oracle/glue/ref_glue.c (the reference-side binding of INTEGRATION.md) for ref_bind.c.
 * TEST INFRASTRUCTURE: included only after the reference's multiclust.h and our mc_host.h. */
#ifndef REF_GLUE_H
#define REF_GLUE_H
uint8_t *mcamd_flatten_genotypes(const data *dat);
int mcamd_options(const options *opt, const data *dat, const uint8_t *flat_geno, double user_lower_bound, mc_options *mo, mc_data *md);
int mcamd_allocate_model_for_k(const mc_options *mo, const mc_data *md, const model *mod, int device, mc_model **mm);
int mcamd_initialize_model(const mc_options *mo, const mc_data *md, model *mod, mc_model *mm, mc_rng *rng);
void mcamd_em(const mc_options *mo, const mc_data *md, model *mod, mc_model *mm);
int mcamd_fetch_parameters(const options *opt, const data *dat, model *mod, mc_model *mm);
int mcamd_fetch_expected_counts(const options *opt, const data *dat, model *mod, mc_model *mm, double *sik);
double mcamd_log_likelihood(const mc_options *mo, const mc_data *md, mc_model *mm, int which);
double mcamd_em_e_step(const mc_options *mo, const mc_data *md, mc_model *mm);
int mcamd_bootstrap_model(const mc_options *mo, const mc_data *md, int null_K, const double *mle_q, const double *mle_p,
			  int K, int device, mc_rng *rng, mc_simulation *sim, int first_model_of_replicate, mc_model **mm);
#endif
