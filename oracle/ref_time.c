/*
 * ref_time.c -- TEST / MEASUREMENT INFRASTRUCTURE ONLY (oracle side, never shipped, never on the product path).
 *
 * Times the *unmodified* reference EM (em(), em_alg.c:44: em_step / accelerated_em_step until stop()) on a sample handed
 * over as flat arrays, for bench.py's `cpu_baseline` leg (kind "reference").  Linked by oracle/Makefile against the
 * reference's own objects where they lie under /root/reference, like ref_harness.c; the binary (oracle/_ref/ref_time) is
 * what travels to the GPU box, no reference source does.
 *
 * The reference's reader is bypassed on purpose: read_file() bubble-sorts all haplotypes per locus twice
 * (read_file.c:518,577: O((I*ploidy)^2 * L), 13 minutes for config 2), which would turn a 20-second timing sample into
 * hours.  Instead the fields of `data` the EM path reads (em_alg.c, accel_em.c, log_likelihood.c, simplex.c: I, L, M,
 * ploidy, uniquealleles, L_alleles, ILM; allocate_model_for_k(): max_M) are filled here from the same (ua, geno) arrays
 * the HIP path gets, and the starting parameters are written into slot 0 (where initialize_model() leaves its own).  A
 * missing copy (0xFF) adds to no count, as in the reference's reader; the allele slot that reader appends to a locus with
 * missing values -- a column no copy ever matches -- is just another entry of ua[] if the caller wants it (the EM layer reads
 * L_alleles only to ask whether its FIRST entry is the missing code, em_alg.c / multiclust.c:1274: it never is here).
 *
 * usage: ref_time <dir> <I> <L> <ploidy> <K> <max_iter> -- <multiclust argv: -f x [-a [-c]] -k K [-s n]>
 *   <dir>/ua.i32 [L], geno.u8 [I][L][ploidy] (allele index), q0.f64 [I][K] ([K] with -c and for the mixture model), p0.f64 [K][T]
 * writes <dir>/q_ref.f64, p_ref.f64 (final iterate) and prints one JSON line.
 */
#define _POSIX_C_SOURCE 200809L
#include "multiclust.h"
#include <stdint.h>
#include <time.h>

int make_options(options **opt);
int make_data(data **dat);
int make_model(model **mod);
int parse_options(options *opt, data *dat, int argc, const char **argv);
int allocate_model_for_k(options *opt, model *mod, data *dat);
int synchronize(options *opt, data *dat, model *mod);

static void die(const char *msg) { fprintf(stderr, "ref_time: %s\n", msg); exit(2); }

static void *slurp(const char *dir, const char *name, size_t bytes)
{
	char path[4096];
	snprintf(path, sizeof path, "%s/%s", dir, name);
	FILE *f = fopen(path, "rb");
	if (!f) die("cannot open input file");
	void *buf = malloc(bytes ? bytes : 1);
	if (!buf || fread(buf, 1, bytes, f) != bytes) die("short input file");
	fclose(f);
	return buf;
}

static double now(void)
{
	struct timespec ts;
	clock_gettime(CLOCK_MONOTONIC, &ts);
	return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

int main(int argc, const char **argv)
{
	options *opt = NULL;
	data *dat = NULL;
	model *mod = NULL;
	int sep = -1;

	for (int i = 1; i < argc; i++)
		if (!strcmp(argv[i], "--")) { sep = i; break; }
	if (sep != 7) die("usage: ref_time <dir> <I> <L> <ploidy> <K> <max_iter> -- <args>");
	const char *dir = argv[1];
	int I = atoi(argv[2]), L = atoi(argv[3]), ploidy = atoi(argv[4]), K = atoi(argv[5]), max_iter = atoi(argv[6]);
	if (I < 1 || L < 1 || ploidy < 1 || K < 1 || max_iter < 1) die("bad dimensions");

	if (make_options(&opt) || make_data(&dat) || make_model(&mod)) die("make_* failed");
	if (parse_options(opt, dat, argc - sep, argv + sep)) die("parse_options failed");
	opt->write_files = 0;
	opt->verbosity = 1;		/* QUIET */
	opt->max_iter = max_iter;	/* stop_condition(): n_iter > max_iter (em_alg.c:148) */
	opt->abs_error = 1e-300;	/* never "converged" inside the sample */
	opt->rel_error = 0;
	if (opt->max_K != K) die("-k does not match K");

	double t0 = now();
	int *ua = slurp(dir, "ua.i32", (size_t)L * sizeof(int));
	uint8_t *geno = slurp(dir, "geno.u8", (size_t)I * L * ploidy);
	size_t T = 0;
	int M = 0;
	for (int l = 0; l < L; l++) {
		if (ua[l] < 1) die("locus without alleles in the sample");
		T += ua[l];
		if (ua[l] > M) M = ua[l];
	}
	dat->I = I; dat->L = L; dat->ploidy = ploidy; dat->M = M; dat->missing_data = 0;
	dat->uniquealleles = ua;
	dat->L_alleles = malloc(L * sizeof *dat->L_alleles);
	int *codes = malloc(T * sizeof *codes);
	for (size_t l = 0, t = 0; l < (size_t)L; t += ua[l], l++) {
		dat->L_alleles[l] = codes + t;
		for (int m = 0; m < ua[l]; m++) codes[t + m] = m + 1;	/* any code but MISSING */
	}
	dat->ILM = malloc(I * sizeof *dat->ILM);
	int **rows = malloc((size_t)I * L * sizeof *rows);
	int *counts = calloc((size_t)I * T, sizeof *counts);
	if (!dat->L_alleles || !codes || !dat->ILM || !rows || !counts) die("out of memory");
	for (int i = 0; i < I; i++) {
		dat->ILM[i] = rows + (size_t)i * L;
		size_t t = 0;
		for (int l = 0; l < L; t += ua[l], l++) {
			int *c = counts + (size_t)i * T + t;
			dat->ILM[i][l] = c;
			for (int a = 0; a < ploidy; a++) {
				uint8_t m = geno[((size_t)i * L + l) * ploidy + a];
				if (m == 0xFF) { dat->missing_data = 1; continue; }	/* a missing copy counts nowhere (read_file.c:600-620) */
				if (m >= ua[l]) die("allele index out of range");
				c[m]++;
			}
		}
	}
	free(geno);

	if (synchronize(opt, dat, mod)) die("synchronize failed");
	mod->K = K;
	dat->max_M = dat->M > mod->K ? dat->M : mod->K;
	if (allocate_model_for_k(opt, mod, dat)) die("allocate_model_for_k failed");
	double t_alloc = now() - t0;

	/* mixing proportions: one row per individual, or one row in all with -c and for the mixture model (multiclust.c:1199-1226) */
	const int shared_eta = !opt->admixture || opt->eta_constrained;
	double *q0 = slurp(dir, "q0.f64", (shared_eta ? (size_t)K : (size_t)I * K) * sizeof(double));
	double *p0 = slurp(dir, "p0.f64", (size_t)K * T * sizeof(double));
	if (shared_eta)
		for (int k = 0; k < K; k++) mod->vetak[0][k] = q0[k];
	else
		for (int i = 0; i < I; i++)
			for (int k = 0; k < K; k++) mod->vetaik[0][i][k] = q0[(size_t)i * K + k];
	for (int k = 0; k < K; k++) {
		size_t t = 0;
		for (int l = 0; l < L; t += ua[l], l++)
			for (int m = 0; m < ua[l]; m++) mod->vpklm[0][k][l][m] = p0[(size_t)k * T + t + m];
	}
	/* the state maximize_likelihood() and initialize_model() leave in front of em() (multiclust.c:518-524, rnd_init.c:57-70) */
	mod->n_iter = 0;
	mod->logL = -INFINITY;
	mod->converged = mod->stopped = mod->iter_stop = mod->time_stop = 0;
	mod->accel_step = 0;
	mod->delta_index = 0;
	mod->pindex = mod->findex = mod->tindex = 0;
	mod->start = clock();

	t0 = now();
	em(opt, dat, mod);
	double t_em = now() - t0;

	char path[4096];
	snprintf(path, sizeof path, "%s/q_ref.f64", dir);
	FILE *f = fopen(path, "wb");
	if (!f) die("cannot write q_ref.f64");
	if (shared_eta) fwrite(mod->vetak[mod->pindex], sizeof(double), K, f);
	else for (int i = 0; i < I; i++) fwrite(mod->vetaik[mod->pindex][i], sizeof(double), K, f);
	fclose(f);
	snprintf(path, sizeof path, "%s/p_ref.f64", dir);
	f = fopen(path, "wb");
	if (!f) die("cannot write p_ref.f64");
	for (int k = 0; k < K; k++)
		for (int l = 0; l < L; l++) fwrite(mod->vpklm[mod->pindex][k][l], sizeof(double), ua[l], f);
	fclose(f);

	printf("{\"I\": %d, \"L\": %d, \"T\": %zu, \"K\": %d, \"accel_scheme\": %d, \"n_iter\": %d, \"em_s\": %.6f, \"setup_s\": %.3f, "
	       "\"logL\": %.17g, \"lower_bound\": %.17g, \"iter_stop\": %d}\n",
	       I, L, T, K, opt->accel_scheme, mod->n_iter, t_em, t_alloc, mod->logL, opt->lower_bound, mod->iter_stop);
	return 0;
}
