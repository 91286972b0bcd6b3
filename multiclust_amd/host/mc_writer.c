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

#include <math.h>
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

int mc_read_afile(const char *path, int I, int **labels, int *pK)
{
	/* read_file.c:970-999 */
	FILE *fp = fopen(path, "r");
	int *lab, top = 0;
	*labels = NULL;
	*pK = 0;
	if (!fp) {
		fprintf(stderr, "ERROR [mc_writer.c::mc_read_afile]: could not open file '%s'\n", path);
		return MC_EXIT_FILE_OPEN_ERROR;
	}
	if (!(lab = malloc(sizeof(int) * (size_t)(I > 0 ? I : 1)))) { fclose(fp); return MC_EXIT_MEMORY_ALLOCATION; }
	for (int i = 0; i < I; i++) {
		if (fscanf(fp, "%d", &lab[i]) != 1) {
			fprintf(stderr, "ERROR [mc_writer.c::mc_read_afile]: format of partition file '%s': %d labels wanted, %d found\n", path, I, i);
			fclose(fp);
			free(lab);
			return MC_EXIT_FILE_FORMAT_ERROR;
		}
		lab[i]--;
		/* (a label below 1 indexes in front of the reference's contingency table: refused here) */
		if (lab[i] < 0) {
			fprintf(stderr, "ERROR [mc_writer.c::mc_read_afile]: partition file '%s': labels start at 1\n", path);
			fclose(fp);
			free(lab);
			return MC_EXIT_FILE_FORMAT_ERROR;
		}
		if (lab[i] > top) top = lab[i];
	}
	fclose(fp);
	*labels = lab;
	*pK = top + 1;
	return 0;
}

double mc_adjusted_rand(int n, int k1, int k2, const int *cl1, const int *cl2)
{
	/* multiclust.c:1903-1985, ADJUSTED_RAND_INDEX branch: the same sums in the same order */
	double *nmat = calloc((size_t)k1 * (size_t)k2 + 1, sizeof *nmat);
	double *sumtr = calloc((size_t)k1 + 1, sizeof *sumtr), *sumpr = calloc((size_t)k2 + 1, sizeof *sumpr);
	double nidot2sum = 0, nij2sum = 0, ndotj2sum = 0, index = NAN;
	if (nmat && sumtr && sumpr) {
		for (int i = 0; i < n; i++) nmat[(size_t)cl1[i] * k2 + cl2[i]] += 1;	/* (ints in the reference: exact either way) */
		for (int i = 0; i < k1; i++)
			for (int j = 0; j < k2; j++) sumtr[i] += nmat[(size_t)i * k2 + j];
		for (int j = 0; j < k2; j++)
			for (int i = 0; i < k1; i++) sumpr[j] += nmat[(size_t)i * k2 + j];
		for (int i = 0; i < k1; i++) nidot2sum += sumtr[i] * (sumtr[i] - 1) / 2;
		for (int i = 0; i < k1; i++)
			for (int j = 0; j < k2; j++) {
				/* nmat[i][j] * (nmat[i][j] - 1) / 2.0 with int nmat: an integer product (exact here up to any count; the
				 * reference's overflows past 46 340 individuals in one cell), then divided in double */
				const long long c = (long long)nmat[(size_t)i * k2 + j];
				nij2sum += (double)(c * (c - 1)) / 2.0;
			}
		for (int j = 0; j < k2; j++) ndotj2sum += sumpr[j] * (sumpr[j] - 1) / 2.0;
		const double term3 = nidot2sum * ndotj2sum / (n * (n - 1.) / 2.);
		const double term1 = nij2sum - term3;
		const double term2 = (nidot2sum + ndotj2sum) / 2 - term3;
		index = term1 / term2;
	}
	free(nmat); free(sumtr); free(sumpr);
	return index;
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
