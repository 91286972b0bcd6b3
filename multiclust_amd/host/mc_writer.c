/*
 * mc_writer.c -- result files of MULTICLUST (reference write_file.c:203-732), same names and row formats:
 *   <f>.<admix|mix>.K=<K>.out.txt      logL / AIC / BIC / count.K
 *   <f>.<admix|mix>.K=<K>.etaik.txt    (or .etak.txt when eta is shared)
 *   <f>.<admix|mix>.K=<K>.pklm.txt
 *   <f>_admix_indivq_<K>.indivq  /  <f>.mix.K=<K>.indivq
 *   <f>_admix_popq_<K>.popq      /  <f>_mix_popq.popq
 * The reference computes the popq / indivq / partition numbers by looping over diklm[i][k][l][m]; every one of
 * those loops is a sum over (l,m), so they are functions of S_ik = sum_lm d_iklm, which the device returns
 * (mchip_get_expected_counts).
 */
#include "mc_cli.h"

#include <stdlib.h>
#include <string.h>

static void stem(const mc_cli_options *opt, char *out, size_t n)
{
	if (opt->outfile_name) {
		snprintf(out, n, "%s", opt->outfile_name);
		return;
	}
	const size_t pl = strlen(opt->path);
	const int sep = pl && opt->path[pl - 1] != '/' && opt->path[pl - 1] != '\\';
	snprintf(out, n, "%s%s%s", opt->path, sep ? "/" : "", opt->filename_file);
}

static FILE *open_out(const char *path)
{
	FILE *fp = fopen(path, "w");
	if (!fp) fprintf(stderr, "ERROR [mc_writer.c::open_out]: could not open file '%s'\n", path);
	return fp;
}

void mc_partition(const mc_cli_data *dat, const mc_fit_view *fit, int *I_K, int *count_K)
{
	/* partition_admixture (write_file.c:350-382): argmax_k sum_lm d_iklm; partition_mixture (585-603): argmax_k vik.
	 * First maximum wins (strict >). */
	for (int k = 0; k < fit->K; k++) count_K[k] = 0;
	for (int i = 0; i < dat->I; i++) {
		const double *s = fit->sik + (size_t)i * fit->K;
		int best = 0;
		for (int k = 1; k < fit->K; k++)
			if (s[k] > s[best]) best = k;
		if (I_K) I_K[i] = best;
		count_K[best]++;
	}
}

int mc_write_results(const mc_cli_options *opt, const mc_cli_data *dat, const mc_fit_view *fit, const int *count_K)
{
	char base[4096], path[4200];
	const int K = fit->K, admix = opt->em.admixture;
	const char *mdl = admix ? "admix" : "mix";
	const int shared_eta = (!admix || opt->em.eta_constrained);
	FILE *fp;
	stem(opt, base, sizeof base);

	/* write_file_detail (write_file.c:203-348) */
	snprintf(path, sizeof path, "%s.%s.K=%d.out.txt", base, mdl, K);
	if (!(fp = open_out(path))) return 1;
	fprintf(fp, "logL = %f (%s)\n", fit->logL, fit->converged ? "converged" : "not converged");
	fprintf(fp, "AIC = %f\n", fit->aic);
	fprintf(fp, "BIC = %f\n\n", fit->bic);
	fprintf(fp, "count.K\n");
	for (int k = 0; k < K; k++) fprintf(fp, "%d ", count_K[k]);
	fprintf(fp, "\n\n");
	fclose(fp);

	if (shared_eta) {
		snprintf(path, sizeof path, "%s.%s.K=%d.etak.txt", base, mdl, K);
		if (!(fp = open_out(path))) return 1;
		fprintf(fp, "i\tk\tetak\n");
		for (int k = 0; k < K; k++) fprintf(fp, "%d\t%f\n", k, fit->q[k]);
		fprintf(fp, "\n");
	} else {
		snprintf(path, sizeof path, "%s.%s.K=%d.etaik.txt", base, mdl, K);
		if (!(fp = open_out(path))) return 1;
		fprintf(fp, "i\tk\tetaik\n");
		for (int i = 0; i < dat->I; i++)
			for (int k = 0; k < K; k++) fprintf(fp, "%d\t%d\t%f\n", i, k, fit->q[(size_t)i * K + k]);
		fprintf(fp, "\n");
	}
	fclose(fp);

	snprintf(path, sizeof path, "%s.%s.K=%d.pklm.txt", base, mdl, K);
	if (!(fp = open_out(path))) return 1;
	fprintf(fp, "k\tl\tm\tKLM\n");
	for (int k = 0; k < K; k++)
		for (int l = 0; l < dat->L; l++)
			for (int m = 0; m < dat->uniquealleles[l]; m++)
				fprintf(fp, "%d\t%d\t%d\t%f\n", k, l, m, fit->p[(size_t)k * dat->T + dat->toff[l] + m]);
	fprintf(fp, "\n");
	fclose(fp);

	/* popq (write_file.c:398-475 admixture, 618-690 mixture) */
	if (admix) snprintf(path, sizeof path, "%s_admix_popq_%d.popq", base, K);
	else snprintf(path, sizeof path, "%s_mix_popq.popq", base);
	if (!(fp = open_out(path))) return 1;
	double *vp = calloc((size_t)dat->numpops * K, sizeof *vp);
	if (!vp) { fclose(fp); return 1; }
	for (int k = 0; k < K; k++)
		for (int i = 0; i < dat->I; i++) vp[(size_t)dat->locale[i] * K + k] += fit->sik[(size_t)i * K + k];
	for (int n = 0; n < dat->numpops; n++) {
		fprintf(fp, "%s:\t", dat->pops[n]);
		for (int k = 0; k < K; k++) {
			const double denom = admix ? (double)(dat->ploidy * dat->L * dat->i_p[n]) : (double)dat->i_p[n];
			fprintf(fp, "%lf\t", vp[(size_t)n * K + k] / denom);
		}
		fprintf(fp, "%d\n", dat->i_p[n]);
	}
	free(vp);
	fclose(fp);

	/* indivq (write_file.c:492-569 admixture, 700-732 mixture) */
	if (admix) snprintf(path, sizeof path, "%s_admix_indivq_%d.indivq", base, K);
	else snprintf(path, sizeof path, "%s.mix.K=%d.indivq", base, K);
	if (!(fp = open_out(path))) return 1;
	for (int i = 0; i < dat->I; i++) {
		fprintf(fp, "%d\t%s\t(x)\t%s\t:", i, dat->names[i], dat->pops[dat->locale[i]]);
		for (int k = 0; k < K; k++) {
			double v;
			if (!admix) v = fit->sik[(size_t)i * K + k];					/* vik */
			else if (opt->em.eta_constrained || dat->missing_data)			/* write_file.c:525-542 */
				v = fit->sik[(size_t)i * K + k] / (double)(dat->ploidy * dat->L);
			else v = fit->q[(size_t)i * K + k];
			fprintf(fp, "\t%f", v);
		}
		fprintf(fp, "\n");
	}
	fclose(fp);
	return 0;
}
