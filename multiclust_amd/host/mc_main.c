/*
 * mc_main.c -- the `multiclust` command line on top of the MI355X EM hot path.
 *
 * Keeps the reference's observable surface: argv flags as parse_options() reads them (reference
 * multiclust.c:1396-1735; 2-3 letter disambiguation of --bound/--format/--impute/--missing/--projection/
 * --plus/--simulate, -I1), defaults of make_options (902-978), the K loop of estimate_model (365-452), the
 * initialisation loop and bookkeeping of maximize_likelihood (471-656), the per-initialisation stdout line
 * (618-627), print_model_state (718-793), run_bootstrap (675-708) and the -w repetition summary (201-347).
 * Deliberately absent (documented in DESIGN.md): -I/-I1 (allele-index mode gives different numbers from default
 * mode in the reference itself), --impute, --simulate, -x (not implemented in the reference either).
 * Extensions: --device <n> selects the HIP device; --streams <n> runs n fits at a time per GPU; --gpus <n> shards the initialisations of each K over n GPUs of
 * the node (one host thread and one context per GPU, units u = d, d+n, ..., each starting from the serial program's
 * rand() position by jump-ahead), with a single RCCL all-reduce of the per-unit result table, after which the serial
 * bookkeeping is replayed in unit order (admixture model, fixed number of initialisations).
 */
#include "mc_cli.h"

#include <errno.h>
#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static const char *accel_abbrev(const mc_cli_options *o, char *buf)
{
	static const char *ab[] = { "EM", "S1", "S2", "S3", "QN" };
	if (o->em.accel_scheme >= MC_QN) { sprintf(buf, "Q%d", o->em.q); return buf; }	/* multiclust.c:828 */
	return ab[o->em.accel_scheme];
}

static void usage(FILE *fp, const char *prog)
{
	fprintf(fp,
		"Usage: %s -f <file> [-a] [-k <K> | -1 <minK> -2 <maxK>] [options]\n"
		"  -a            admixture model (default: mixture)      -c  shared mixing proportions (with -a)\n"
		"  -k <n>        number of clusters K (default 6)        -1 <n>, -2 <n>  minimum / maximum K\n"
		"  -n <n>        random initialisations (default 50)     -r <seed>  random seed\n"
		"  -s <0..6>     acceleration: 0 EM, 1-3 SQUAREM, 4-6 quasi-Newton q=1..3\n"
		"  -E <d> -e <d> absolute (1e-4) / relative (0) log-likelihood convergence tolerance\n"
		"  -T <n> -t <m> iteration / time (minutes) limits        -i <n>  initial plain EM iterations\n"
		"  -g <n>        step-size back-tracking attempts         --bound <d>  parameter lower bound (1e-8)\n"
		"  --projection  disable the simplex projection           -p <n>  ploidy (default 2)\n"
		"  -b <n>        parametric bootstrap of H0: K-1 vs HA: K -u l <ll> | n <times>  target log likelihood / revisits\n"
		"  -w n <n> | t <min> | m <min>   repeat the fit for timing (no files written)\n"
		"  -d <dir> -o <stem>  output directory / file stem      -R  R-formatted STRUCTURE file\n"
		"  --missing <n> missing-data code (default -9)           -M  print only the maximum log likelihood\n"
		"  -v [level]    verbosity                                --device <n>  HIP device index\n"
		"  --gpus <n>    shard the initialisations over n GPUs (admixture; one RCCL all-reduce of the results)\n"
		"  --streams <n> n concurrent fits per GPU, each on its own stream (small data sets do not fill a GPU)\n"
		"  -P <file> -Q <file>  initial allele frequencies p[k][l][0] (L*K numbers, biallelic loci) and mixing proportions of the admixture model\n"
		"  --randem      Rand-EM initialisation: the best of -m <n> (50) candidates from random allele centers\n"
		"  -A <file>     a partition of the individuals (labels 1, 2, ...): the adjusted Rand index of the fitted one is reported\n", prog);
}

static int arg_int(int argc, const char **argv, int i, long *out)
{
	char *end;
	if (i >= argc) return 1;
	errno = 0;
	*out = strtol(argv[i], &end, 10);
	return errno || end == argv[i];
}
static int arg_dbl(int argc, const char **argv, int i, double *out)
{
	char *end;
	if (i >= argc) return 1;
	errno = 0;
	*out = strtod(argv[i], &end);
	return errno || end == argv[i];
}

static void defaults(mc_cli_options *o)
{
	memset(o, 0, sizeof *o);
	mc_make_options(&o->em);
	o->path = "./";
	o->min_K = o->max_K = 6;
	o->n_init = 50;
	o->n_rand_em_init = 50;
	o->missing_value = MC_MISSING;
	o->ploidy = 2;
	o->n_repeat = 1;
	o->write_files = 1;
	o->compact = 1;
	o->n_gpus = 1;
	o->n_streams = 1;
}

#define BAD(msg) do { fprintf(stderr, "ERROR [mc_main.c::parse_options]: %s (argument '%s'); try -h\n", msg, i < argc ? argv[i] : ""); return MC_EXIT_INVALID_CMD_ARGUMENT; } while (0)

static int parse_options(mc_cli_options *o, int argc, const char **argv)
{
	long v;
	double d;
	for (int i = 1; i < argc; i++) {
		if (strlen(argv[i]) < 2) BAD("malformed option");
		size_t j = 1;
		char a = argv[i][j];
		while (a == '-' && ++j < strlen(argv[i])) a = argv[i][j];
		const char *w = &argv[i][j];
		switch (a) {
		case 'a': o->em.admixture = 1; break;
		case 'b':
			if (!strncmp(w, "bou", 3)) { if (arg_dbl(argc, argv, ++i, &d) || d < 0) BAD("--bound"); o->em.lower_bound = d; }
			else { if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-b"); o->n_bootstrap = (int)v; }
			break;
		case 'c': o->em.eta_constrained = 1; break;
		case 'd':
			if (!strncmp(w, "dev", 3)) { if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("--device"); o->device = (int)v; }
			else { if (++i >= argc) BAD("-d"); o->path = argv[i]; }
			break;
		case 'e': if (arg_dbl(argc, argv, ++i, &d) || d < 0) BAD("-e"); o->em.rel_error = d; break;
		case 'E': if (arg_dbl(argc, argv, ++i, &d) || d < 0) BAD("-E"); o->em.abs_error = d; break;
		case 'f':
			if (!strncmp(w, "fo", 2)) { ++i; break; }	/* --format only matters to the data writer */
			if (++i >= argc) BAD("-f");
			o->filename = o->filename_file = argv[i];
			for (size_t x = strlen(o->filename); x-- > 1;)
				if (o->filename[x] == '/') { o->filename_file = &o->filename[x + 1]; break; }
			break;
		case 'g':
			if (!strncmp(w, "gp", 2)) { if (arg_int(argc, argv, ++i, &v) || v < 1 || v > 64) BAD("--gpus"); o->n_gpus = (int)v; }
			else { if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-g"); o->em.adjust_step = (int)v; }
			break;
		case 'h': usage(stdout, argv[0]); return 1;
		case 'i':
			if (!strncmp(w, "im", 2)) BAD("--impute is not supported by this build");
			if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-i");
			o->em.n_init_iter = (int)v;
			break;
		case 'I': BAD("-I / -I1 (alleles as indices) is not supported: the reference's own results differ in that mode");
		case '1': if (arg_int(argc, argv, ++i, &v) || v < 1) BAD("-1"); o->min_K = (int)v; break;
		case '2': if (arg_int(argc, argv, ++i, &v) || v < 1) BAD("-2"); o->max_K = (int)v; break;
		case 'k': if (arg_int(argc, argv, ++i, &v) || v < 1) BAD("-k"); o->min_K = o->max_K = (int)v; break;
		case 'm':
			if (!strncmp(w, "mi", 2)) { if (arg_int(argc, argv, ++i, &v)) BAD("--missing"); o->missing_value = (int)v; }
			else { if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-m"); o->n_rand_em_init = o->em.n_rand_em_init = (int)v; }
			break;
		case 'M': o->parallel = 1; o->n_repeat = 1; o->em.verbosity = MC_SILENT; break;
		case 'n': if (arg_int(argc, argv, ++i, &v)) BAD("-n"); o->n_init = (int)v; if (!v) o->n_repeat = 0; break;
		case 'o': if (++i >= argc) BAD("-o"); o->outfile_name = argv[i]; break;
		case 'p':
			if (!strncmp(w, "pr", 2)) o->em.do_projection = 0;
			else if (!strncmp(w, "pl", 2)) ;	/* --plus: data-writer option */
			else { if (arg_int(argc, argv, ++i, &v) || v < 1) BAD("-p"); o->ploidy = (int)v; }
			break;
		case 'P': if (++i >= argc) BAD("-P"); o->pfile = argv[i]; break;
		case 'Q': if (++i >= argc) BAD("-Q"); o->qfile = argv[i]; break;
		case 'A': if (++i >= argc) BAD("-A"); o->afile = argv[i]; break;	/* multiclust.c:1416-1418 */
		case 'R': o->R_format = 1; break;
		case 'r':
			/* extension: --randem selects the Rand-EM initialisation the reference carries but cannot reach
			 * (initialization_procedure stays NOTHING, multiclust.c:935); -m <n> is its number of candidates */
			if (!strncmp(w, "ra", 2)) { o->em.initialization_procedure = MC_RAND_EM; break; }
			if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-r");
			o->em.seed = (unsigned)v;
			o->seed_given = 1;
			break;
		case 'x': BAD("-x (block relaxation) is not implemented, as in the reference");
		case 's':
			if (!strncmp(w, "si", 2)) BAD("--simulate is not supported by this build");
			if (!strncmp(w, "st", 2)) { if (arg_int(argc, argv, ++i, &v) || v < 1 || v > 16) BAD("--streams"); o->n_streams = (int)v; break; }
			if (arg_int(argc, argv, ++i, &v) || v < 0 || v > 6) BAD("-s");
			o->em.accel_scheme = (int)v;
			break;
		case 't': if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-t"); o->em.n_seconds = 60u * (unsigned)v; break;
		case 'T': if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-T"); o->em.max_iter = (int)v; break;
		case 'u':
			while (++i < argc && argv[i][0] != '-') {
				if (argv[i][0] == 'l') { if (arg_dbl(argc, argv, ++i, &d)) BAD("-u l"); o->target_ll = 1; o->desired_ll = d; }
				else if (argv[i][0] == 'n') { if (arg_int(argc, argv, ++i, &v) || v < 0) BAD("-u n"); o->target_revisit = (int)v; }
				else BAD("-u");
			}
			i--;
			break;
		case 'v':
			if (i + 1 == argc) o->em.verbosity = MC_VERBOSE;
			else if (arg_int(argc, argv, i + 1, &v)) o->em.verbosity = MC_VERBOSE;
			else { o->em.verbosity = (int)v; i++; }
			break;
		case 'w':
			while (++i < argc && argv[i][0] != '-') {
				if (arg_int(argc, argv, i + 1, &v)) BAD("-w");
				if (argv[i][0] == 't') o->repeat_seconds = 60u * (unsigned)v;
				else if (argv[i][0] == 'm') o->max_repeat_seconds = 60u * (unsigned)v;
				else if (argv[i][0] == 'n') { if (v <= 0) BAD("-w n"); o->n_repeat = (int)v; }
				else BAD("-w");
				i++;
			}
			i--;
			o->write_files = 0;
			break;
		default:
			fprintf(stderr, "ERROR [mc_main.c::parse_options]: unknown option (argument '%s'); try -h\n", argv[i]);
			return MC_EXIT_INVALID_CMD_OPTION;
		}
	}
	if (!o->filename) {
		fprintf(stderr, "ERROR [mc_main.c::parse_options]: You must specify the data file with command line option '-f'.  Try '-h' for help.\n");
		return MC_EXIT_INVALID_CMDLINE;
	}
	return 0;
}

/* ---- state across initialisations / models (reference struct _model, multiclust.h:337-360) ---- */
typedef struct run_state {
	mc_summary sum;
	double max_logL_H0, ts_obs, ts_bs;
	int n_targetll_times, n_targetll_init, time_stop;
	int aic_K, bic_K, null_K, alt_K;
	double *mle_q, *mle_p;		/* H0 MLEs for the bootstrap (multiclust.c:562-581) */
	int mle_K;
	mc_rng rng;
	FILE *out;			/* stdout, or a replicate's buffer when bootstrap replicates run on several devices */
	mchip_comm **comm;		/* the run's RCCL communicator over devices device..device+n_gpus-1, created on first use */
	mc_model **sim_models;		/* [2] or NULL: this worker's H0 / HA models of the bootstrap data sets, kept from one replicate to
					 * the next (same shape, same K: the device re-uses every buffer, mc_model_resimulate) */
	/* -A (multiclust.c:602-612): the partition read from the file (labels - 1, pK of them), the MAP partition of the last fit that
	 * was partitioned (dat->I_K) and the adjusted Rand index of the two (model::arand: never reset, as in the reference) */
	int *partition_from_file, pK, *I_K;
	double arand;
} run_state;

/* sharded runs use n_gpus * n_streams workers (host thread + context + stream each); worker x sits on device
 * device + x % n_gpus, so consecutive units land on different GPUs first */
static int n_workers(const mc_cli_options *o) { return (o->n_gpus < 1 ? 1 : o->n_gpus) * (o->n_streams < 1 ? 1 : o->n_streams); }
static int worker_device(const mc_cli_options *o, int x) { return o->device + x % (o->n_gpus < 1 ? 1 : o->n_gpus); }

/* with one GPU its table is already complete (the workers are threads of this process); the rehearsal of the sharded
 * path on one GPU (MC_FORCE_SHARDED) still goes through RCCL */
static int exchange_needed(const mc_cli_options *o) { return o->n_gpus > 1 || getenv("MC_FORCE_SHARDED") != NULL; }

/* one communicator per run: creating it (ncclCommInitAll) costs seconds, an exchange microseconds */
static int get_comm(const mc_cli_options *o, run_state *st, mchip_comm **out)
{
	int rc = 0;
	if (!*st->comm) {
		const int n_dev = o->n_gpus < 1 ? 1 : o->n_gpus;
		int *devs = malloc(sizeof(int) * (size_t)n_dev);
		if (!devs) return MCHIP_ERR_ALLOC;
		for (int x = 0; x < n_dev; x++) devs[x] = o->device + x;
		rc = mchip_comm_create(st->comm, n_dev, devs);
		free(devs);
		if (rc) fprintf(stderr, "ERROR [mc_main.c]: cannot create the RCCL communicator (status %d)\n", rc);
	}
	*out = *st->comm;
	return rc;
}

/* the run's one exchange: RCCL all-reduce (sum) of the per-unit table; MC_TRACE_EXCHANGE=1 reports on stderr what
 * the communicator is and has done (tests/test_gpu_cli.py reads it: the one-GPU rehearsal has to go through RCCL, not past it) */
static int exchange_table(const mc_cli_options *o, run_state *st, double **tab, int count, const char *what)
{
	mchip_comm *comm = NULL;
	int rc = get_comm(o, st, &comm);
	if (rc) return rc;
	if ((rc = mchip_comm_all_reduce(comm, tab, count, 0))) { fprintf(stderr, "ERROR [mc_main.c]: %s\n", mchip_comm_last_error(comm)); return rc; }
	if (getenv("MC_TRACE_EXCHANGE")) {
		int n = 0, ver = 0;
		unsigned long long done = 0;
		mchip_comm_info(comm, &n, &ver, &done);
		fprintf(stderr, "exchange: RCCL %d all-reduce #%llu, %d doubles (%s) over %d device(s)\n", ver, done, count, what, n);
	}
	return 0;
}

static void print_model_state(const mc_cli_options *o, const mc_cli_data *d, const run_state *st, int K, int diff, int newline)
{
	char ab[16];
	(void)d;
	if (o->compact) {	/* multiclust.c:720-746 */
		fprintf(st->out, "%s %s %s %d %u %e %e %e %e %f %f %f ", o->filename, accel_abbrev(o, ab), o->em.admixture ? "admix" : "mix", K,
		       o->em.seed, o->em.eta_lower_bound, o->em.p_lower_bound, o->em.abs_error, o->em.rel_error,
		       st->sum.max_logL, st->sum.aic, st->sum.bic);
		if (o->afile) fprintf(st->out, "%f ", st->arand); else fprintf(st->out, "ND ");	/* multiclust.c:728-731 */
		fprintf(st->out, "%s %02d:%02d:%02d %d %d %d %d", st->sum.ever_converged ? "converged" : "not", diff / 3600, (diff % 3600) / 60,
		       diff % 60, st->sum.n_total_iter, st->sum.n_init, st->sum.n_maxll_init, st->sum.n_maxll_times);
		if (o->target_ll) fprintf(st->out, " %f %d %d", o->desired_ll, st->n_targetll_init, st->n_targetll_times);
		if (st->time_stop) fprintf(st->out, " time");
		if (newline) fprintf(st->out, "\n");
	} else {		/* multiclust.c:748-792 */
		fprintf(st->out, "Dataset: %s\nMethod/Model: %s, %s, K=%d\n", o->filename, accel_abbrev(o, ab), o->em.admixture ? "admix" : "mix", K);
		fprintf(st->out, "Convergence: ae=%e, re=%e\nBounds: e=%e, p=%e\n", o->em.abs_error, o->em.rel_error, o->em.eta_lower_bound, o->em.p_lower_bound);
		fprintf(st->out, "Total number of iterations: %d\nTotal time: %02d:%02d:%02d\n", st->sum.n_total_iter, diff / 3600, (diff % 3600) / 60, diff % 60);
		fprintf(st->out, "Iteration of max log likelihood: %d of %d\nNumber of times reach max log likelihood: %d\n", st->sum.n_maxll_init, st->sum.n_init, st->sum.n_maxll_times);
		fprintf(st->out, "Maximum log likelihood: %f\nAIC: %f\nBIC: %f\nConverged: %s\n", st->sum.max_logL, st->sum.aic, st->sum.bic, st->sum.ever_converged ? "yes" : "no");
		if (st->time_stop) fprintf(st->out, "WARNING: Fitting stopped because ran out of time\n");
	}
}

/* maximize_likelihood (multiclust.c:471-656) for one K */
/* initialize_model (rnd_init.c:54-89).  With -P and -Q the admixture model starts from the parameters in those files instead
 * of a random partition (rnd_init.c:74-76): read_qfile (read_file.c:880-922: I*K numbers in i, k order, or K with -c) and
 * read_pfile (924-959: "assumes biallelic locus": L*K numbers p[k][l][0] in l, k order, p[k][l][1] = 1 - p[k][l][0], any further
 * allele of a locus keeps what the slot held).  No rand() is consumed, so every initialisation starts from the same point. */
static int cli_initialize(const mc_cli_options *o, const mc_cli_data *d, const mc_data *md, mc_model *mod, mc_rng *rng)
{
	if (!(o->em.admixture && o->pfile && o->qfile)) return mc_initialize_model(&o->em, md, mod, rng);
	const int K = mod->K, nq = o->em.eta_constrained ? K : d->I * K;
	double *q = malloc(sizeof(double) * (size_t)nq), *p = malloc(sizeof(double) * (size_t)K * d->T);
	FILE *fq = fopen(o->qfile, "r"), *fp = fopen(o->pfile, "r");
	int rc = 0;
	mod->n_iter = 0;
	mod->logL = -INFINITY;
	mod->converged = 0;
	if (o->em.accel_scheme) mod->pindex = mod->tindex = mod->findex = 0;
	if (!q || !p) rc = MCHIP_ERR_ALLOC;
	if (!rc && (!fq || !fp)) { fprintf(stderr, "ERROR [mc_main.c::cli_initialize]: cannot open '%s'\n", fq ? o->pfile : o->qfile); rc = MC_EXIT_FILE_OPEN_ERROR; }
	for (int x = 0; !rc && x < nq; x++)
		if (fscanf(fq, "%lf", &q[x]) != 1) { fprintf(stderr, "ERROR [mc_main.c::cli_initialize]: format of '%s'\n", o->qfile); rc = MC_EXIT_FILE_FORMAT_ERROR; }
	if (!rc) rc = mc_model_get_p(mod, mod->tindex, p);
	for (int l = 0; !rc && l < d->L; l++)
		for (int k = 0; !rc && k < K; k++) {
			double v;
			if (fscanf(fp, "%lf", &v) != 1) { fprintf(stderr, "ERROR [mc_main.c::cli_initialize]: format of '%s'\n", o->pfile); rc = MC_EXIT_FILE_FORMAT_ERROR; break; }
			/* (the reference stores 1 - v in the locus's second allele slot whether or not there is one, read_file.c:950: a
			 * monomorphic locus makes it write past its allocation; refused here) */
			if (d->uniquealleles[l] < 2) { fprintf(stderr, "ERROR [mc_main.c::cli_initialize]: -P needs two alleles at every locus\n"); rc = MC_EXIT_INVALID_USER_SETUP; break; }
			p[(size_t)k * d->T + d->toff[l]] = v;
			p[(size_t)k * d->T + d->toff[l] + 1] = 1 - v;
		}
	if (!rc) rc = mc_model_set_q(mod, mod->tindex, q);
	if (!rc) rc = mc_model_set_p(mod, mod->tindex, p);
	if (fq) fclose(fq);
	if (fp) fclose(fp);
	free(q); free(p);
	return rc;
}

static int maximize_likelihood(const mc_cli_options *o, const mc_cli_data *d, const mc_data *md, mc_model *mod, run_state *st, int bootstrap)
{
	const int K = mod->K, nq = (o->em.admixture && !o->em.eta_constrained) ? d->I * K : K;
	const int npar = mc_no_parameters(&o->em, md, K);
	/* estimate_model resets the maximum per call, not per K (multiclust.c:377), and nothing resets AIC / BIC: a K whose fits do
	 * not beat the previous K's maximum reports that K's figures (and writes no files) */
	const double max_logL_keep = st->sum.max_logL, aic_keep = st->sum.aic, bic_keep = st->sum.bic;
	double *q = NULL, *p = NULL, *sik = NULL;
	int *count_K = NULL, rc = 0;
	mc_summary_reset(&st->sum);
	st->sum.max_logL = max_logL_keep;
	if (max_logL_keep > -INFINITY) { st->sum.aic = aic_keep; st->sum.bic = bic_keep; }
	st->n_targetll_times = 0;
	st->time_stop = 0;
	mod->start = clock();
	const clock_t start = mod->start;
	for (int i = 0; o->target_revisit || o->target_ll || o->em.n_seconds || i < o->n_init; i++) {
		const int delta_keep = mod->delta_index;
		mc_reset_model_state(mod);
		mod->delta_index = delta_keep;
		mod->start = start;			/* the time limit spans all initialisations (multiclust.c:488) */
		mchip_progress_note("maximize_likelihood: initialisation");
		if ((rc = cli_initialize(o, d, md, mod, &st->rng))) goto DONE;
		mchip_progress_note("maximize_likelihood: mc_em");
		mc_em(&o->em, md, mod);
		mchip_progress_note("maximize_likelihood: bookkeeping and result files");
		if (mod->fatal == MC_FATAL_DEVICE) { rc = MCHIP_ERR_HIP; goto DONE; }
		if (mod->fatal) exit(0);		/* the reference's reaction to NaN / decreasing logL (em_alg.c:106-120) */
		mc_unit_result r = { i, mod->logL, mod->converged, mod->n_iter, mod->time_stop, mod->iter_stop, mod->pindex, 0, mod->seconds_run };
		const double prev_max = st->sum.max_logL;
		mc_summary_add(&o->em, &st->sum, &r, npar, d->I);
		if (mod->logL > prev_max) {
			const int keep_mle = !bootstrap && o->n_bootstrap && K == st->null_K;
			const int part_only = o->afile && !o->write_files;	/* nothing is written, but -A wants the partition */
			if (keep_mle || (!bootstrap && o->write_files) || part_only) {
				if (!q) { q = malloc(sizeof(double) * (size_t)nq); p = malloc(sizeof(double) * (size_t)K * d->T);
					  sik = malloc(sizeof(double) * (size_t)d->I * K); count_K = malloc(sizeof(int) * (size_t)K); }
				if (!q || !p || !sik || !count_K) { rc = MCHIP_ERR_ALLOC; goto DONE; }
				if ((rc = mc_model_get_q(mod, mod->pindex, q)) || (rc = mc_model_get_p(mod, mod->pindex, p)) ||
				    (rc = mc_model_get_expected_counts(mod, sik))) goto DONE;
			}
			if (keep_mle) {		/* multiclust.c:562-581 */
				free(st->mle_q); free(st->mle_p);
				st->mle_q = malloc(sizeof(double) * (size_t)nq); st->mle_p = malloc(sizeof(double) * (size_t)K * d->T);
				memcpy(st->mle_q, q, sizeof(double) * (size_t)nq);
				memcpy(st->mle_p, p, sizeof(double) * (size_t)K * d->T);
				st->mle_K = K;
			}
			if (!bootstrap && o->write_files) {	/* multiclust.c:584-600 */
				mc_fit_view fv = { K, mod->converged, mod->logL, st->sum.aic, st->sum.bic, q, p, sik };
				mc_partition(d, &fv, st->I_K, count_K);
				if ((rc = mc_write_results(o, d, &fv, count_K))) goto DONE;
			}
			if (o->afile) {		/* multiclust.c:602-612 */
				if (part_only) {
					mc_fit_view fv = { K, mod->converged, mod->logL, st->sum.aic, st->sum.bic, q, p, sik };
					mc_partition(d, &fv, st->I_K, count_K);
				}
				/* (a bootstrap fit with result files on is not partitioned in the reference: dat->I_K is then the partition
				 * of an earlier fit, possibly with a cluster index this K does not have -- it indexes past its table there;
				 * the index is left as it was here) */
				int fits = 1;
				for (int x = 0; x < d->I; x++) if (st->I_K[x] >= K) fits = 0;
				if (fits) st->arand = mc_adjusted_rand(d->I, st->pK, K, st->partition_from_file, st->I_K);
			}
		}
		if (!bootstrap && o->em.verbosity > MC_QUIET && o->write_files)	/* multiclust.c:618-627 */
			fprintf(st->out, "K = %d, initialization = %d: %f (%s) in %3d iterations, %02d:%02d:%02d (%f; %d), seed: %u\n", K, i, mod->logL,
			       mod->converged ? "converged" : "not converged", mod->n_iter, (int)(mod->seconds_run / 3600),
			       (int)((((int)mod->seconds_run) % 3600) / 60), ((int)mod->seconds_run) % 60, st->sum.max_logL,
			       st->sum.n_maxll_times, o->em.seed);
		if (K == 1) break;
		if (mod->time_stop) { st->time_stop = 1; break; }
		if (o->target_revisit && st->sum.n_maxll_times >= o->target_revisit) break;
		if (o->target_ll) {		/* multiclust.c:643-652 */
			const int hit = mod->logL > o->desired_ll || (o->em.abs_error && fabs(o->desired_ll - mod->logL) <= o->em.abs_error);
			if (hit) {
				if (!st->n_targetll_times) st->n_targetll_init = st->sum.n_init;
				st->n_targetll_times++;
				if (!o->target_revisit || o->target_revisit <= st->n_targetll_times) break;
			}
		}
	}
DONE:
	free(q); free(p); free(sik); free(count_K);
	return rc;
}


/* ---- initialisations sharded over several GPUs of the node ---- */
typedef struct shard_worker {
	const mc_cli_options *o;
	const mc_cli_data *d;
	const mc_data *md;
	int K, index, n_dev, n_units, want_params;
	const mc_simulation *sim;	/* bootstrap replicate generated on the device, or NULL */
	const mc_rng *starts;		/* [n_units + 1]: the serial stream's state at the start of every unit (and after the last) */
	uint64_t draws;
	mc_unit_result *res;		/* [n_units], shared: worker d writes rows u = d, d + n_dev, ... */
	double best_logL;
	int best_unit, rc;
	double *q, *p, *sik;		/* parameters of this worker's best unit */
} shard_worker;

static void *shard_main(void *arg)
{
	shard_worker *w = arg;
	const int nq = (w->o->em.admixture && !w->o->em.eta_constrained) ? w->d->I * w->K : w->K;
	mc_model *mod = NULL;
	w->best_logL = -INFINITY;
	w->best_unit = -1;
	const int device = worker_device(w->o, w->index);
	if ((w->rc = w->sim ? mc_model_create_simulated(&mod, &w->o->em, w->md, w->K, device, w->sim)
			    : mc_model_create(&mod, &w->o->em, w->md, w->K, device))) return NULL;
	const clock_t start = clock();
	/* unit u starts where the serial stream stands after u initialisations (w->starts, walked once by the caller) */
	for (int u = w->index; u < w->n_units; u += w->n_dev) {
		mc_rng rng = w->starts[u];
		mc_reset_model_state(mod);
		mod->start = start;
		if ((w->rc = mc_initialize_model(&w->o->em, w->md, mod, &rng))) break;
		mc_em(&w->o->em, w->md, mod);
		if (mod->fatal == MC_FATAL_DEVICE) { w->rc = MCHIP_ERR_HIP; break; }
		mc_unit_result *r = &w->res[u];
		r->unit = u; r->logL = mod->logL; r->converged = mod->converged; r->n_iter = mod->n_iter;
		r->time_stop = mod->time_stop; r->iter_stop = mod->iter_stop; r->pindex = mod->pindex; r->fatal = mod->fatal;
		r->seconds_run = mod->seconds_run;
		if (mod->fatal) break;
		if (w->want_params && mod->logL > w->best_logL) {	/* strict: the earliest of equal units wins, as in the serial loop */
			if (!w->q) { w->q = malloc(sizeof(double) * (size_t)nq); w->p = malloc(sizeof(double) * (size_t)w->K * w->d->T);
				     w->sik = malloc(sizeof(double) * (size_t)w->d->I * w->K); }
			if (!w->q || !w->p || !w->sik) { w->rc = MCHIP_ERR_ALLOC; break; }
			if ((w->rc = mc_model_get_q(mod, mod->pindex, w->q)) || (w->rc = mc_model_get_p(mod, mod->pindex, w->p)) ||
			    (w->rc = mc_model_get_expected_counts(mod, w->sik))) break;
			w->best_logL = mod->logL;
			w->best_unit = u;
		}
	}
	mc_model_free(mod);
	return NULL;
}

#define RES_FIELDS 9
static int maximize_likelihood_sharded(const mc_cli_options *o, const mc_cli_data *d, const mc_data *md, int K, run_state *st, int bootstrap,
				       const mc_simulation *sim)
{
	const int n_dev = n_workers(o), n_gpus = o->n_gpus < 1 ? 1 : o->n_gpus, n_units = (K == 1) ? 1 : o->n_init;
	const int nq = (o->em.admixture && !o->em.eta_constrained) ? d->I * K : K;
	const int npar = mc_no_parameters(&o->em, md, K);
	const int keep_mle = !bootstrap && o->n_bootstrap && K == st->null_K;
	const double max_logL_keep = st->sum.max_logL, aic_keep = st->sum.aic, bic_keep = st->sum.bic;
	shard_worker *w = calloc((size_t)n_dev, sizeof *w);
	pthread_t *th = calloc((size_t)n_dev, sizeof *th);
	int *joinable = calloc((size_t)n_dev, sizeof *joinable);	/* (a pthread_t has no "none" value) */
	mc_unit_result *res = calloc((size_t)n_units, sizeof *res);
	double **tab = calloc((size_t)n_gpus, sizeof *tab);
	int *count_K = calloc((size_t)K, sizeof *count_K), rc = 0;
	mc_rng *starts = calloc((size_t)n_units + 1, sizeof *starts);
	if (!w || !th || !joinable || !res || !tab || !count_K || !starts) { rc = MCHIP_ERR_ALLOC; goto DONE; }
	{	/* where each unit starts in the rand() stream: one walk (a jump per unit for the random allele partition, the host-side
		 * replay of the center draws for Rand-EM), instead of every worker replaying the units before its own */
		mc_model walker;
		memset(&walker, 0, sizeof walker);
		walker.K = K;
		rc = mc_unit_starts(&o->em, md, &walker, &st->rng, n_units, starts);
		mc_init_cache_free(&walker);
		if (rc) goto DONE;
	}
	for (int x = 0; x < n_dev; x++) {
		w[x].o = o; w[x].d = d; w[x].md = md; w[x].K = K; w[x].index = x; w[x].n_dev = n_dev; w[x].n_units = n_units;
		w[x].want_params = keep_mle || (!bootstrap && o->write_files) || o->afile != NULL;
		w[x].sim = sim;
		w[x].starts = starts; w[x].draws = mc_draws_per_init(&o->em, md, K); w[x].res = res;
		if (pthread_create(&th[x], NULL, shard_main, &w[x])) shard_main(&w[x]);	/* no thread to be had: in this one */
		else joinable[x] = 1;
	}
	for (int x = 0; x < n_dev; x++) if (joinable[x]) pthread_join(th[x], NULL);
	for (int x = 0; x < n_dev; x++) if (w[x].rc) rc = w[x].rc;
	if (rc) goto DONE;
	st->rng = starts[n_units];	/* where the serial stream stands after these initialisations */

	/* the one exchange: every device's result table holds the rows its workers fitted; an RCCL all-reduce (sum)
	 * completes them all */
	for (int g = 0; g < n_gpus; g++)
		if (!(tab[g] = calloc((size_t)n_units * RES_FIELDS, sizeof(double)))) { rc = MCHIP_ERR_ALLOC; goto DONE; }
	for (int u = 0; u < n_units; u++) {
		double *row = tab[(u % n_dev) % n_gpus] + (size_t)u * RES_FIELDS;
		row[0] = res[u].logL; row[1] = res[u].converged; row[2] = res[u].n_iter; row[3] = res[u].time_stop;
		row[4] = res[u].iter_stop; row[5] = res[u].pindex; row[6] = res[u].fatal; row[7] = res[u].seconds_run; row[8] = 1.0;
	}
	if (exchange_needed(o) && (rc = exchange_table(o, st, tab, n_units * RES_FIELDS, "per-initialisation results"))) goto DONE;
	for (int x = 1; x < n_gpus; x++)
		if (memcmp(tab[0], tab[x], sizeof(double) * (size_t)n_units * RES_FIELDS)) { fprintf(stderr, "ERROR [mc_main.c]: devices disagree after the all-reduce\n"); rc = MCHIP_ERR_STATE; goto DONE; }

	/* replay the serial bookkeeping in unit order (multiclust.c:534-560, 618-627) */
	mc_summary_reset(&st->sum);
	st->sum.max_logL = max_logL_keep;
	if (max_logL_keep > -INFINITY) { st->sum.aic = aic_keep; st->sum.bic = bic_keep; }	/* as in maximize_likelihood above */
	st->time_stop = 0;
	for (int u = 0; u < n_units; u++) {
		const double *row = tab[0] + (size_t)u * RES_FIELDS;
		if (row[8] != 1.0) { fprintf(stderr, "ERROR [mc_main.c]: unit %d was fitted %g times\n", u, row[8]); rc = MCHIP_ERR_STATE; goto DONE; }
		if ((int)row[6]) exit(0);		/* NaN / decreasing logL: the reference's reaction (em_alg.c:106-120) */
		mc_unit_result r = { u, row[0], (int)row[1], (int)row[2], (int)row[3], (int)row[4], (int)row[5], 0, row[7] };
		mc_summary_add(&o->em, &st->sum, &r, npar, d->I);
		if (!bootstrap && o->em.verbosity > MC_QUIET && o->write_files)
			fprintf(st->out, "K = %d, initialization = %d: %f (%s) in %3d iterations, %02d:%02d:%02d (%f; %d), seed: %u\n", K, u, r.logL,
			       r.converged ? "converged" : "not converged", r.n_iter, (int)(r.seconds_run / 3600),
			       (int)((((int)r.seconds_run) % 3600) / 60), ((int)r.seconds_run) % 60, st->sum.max_logL, st->sum.n_maxll_times, o->em.seed);
	}
	/* the winner's parameters live on the device that fitted it */
	if (st->sum.best_unit >= 0 && (keep_mle || (!bootstrap && o->write_files) || o->afile)) {
		const shard_worker *own = &w[st->sum.best_unit % n_dev];
		if (own->best_unit != st->sum.best_unit) { fprintf(stderr, "ERROR [mc_main.c]: owner of the best unit does not hold it\n"); rc = MCHIP_ERR_STATE; goto DONE; }
		if (keep_mle) {
			free(st->mle_q); free(st->mle_p);
			st->mle_q = malloc(sizeof(double) * (size_t)nq); st->mle_p = malloc(sizeof(double) * (size_t)K * d->T);
			memcpy(st->mle_q, own->q, sizeof(double) * (size_t)nq);
			memcpy(st->mle_p, own->p, sizeof(double) * (size_t)K * d->T);
			st->mle_K = K;
		}
		if (!bootstrap && o->write_files) {
			mc_fit_view fv = { K, res[st->sum.best_unit].converged, st->sum.max_logL, st->sum.aic, st->sum.bic, own->q, own->p, own->sik };
			mc_partition(d, &fv, st->I_K, count_K);
			rc = mc_write_results(o, d, &fv, count_K);
		}
		if (!rc && o->afile) {	/* multiclust.c:602-612: the index of the serial loop's last improvement = the best unit's */
			if (!o->write_files) {
				mc_fit_view fv = { K, res[st->sum.best_unit].converged, st->sum.max_logL, st->sum.aic, st->sum.bic, own->q, own->p, own->sik };
				mc_partition(d, &fv, st->I_K, count_K);
			}
			int fits = 1;
			for (int x = 0; x < d->I; x++) if (st->I_K[x] >= K) fits = 0;
			if (fits) st->arand = mc_adjusted_rand(d->I, st->pK, K, st->partition_from_file, st->I_K);
		}
	}
DONE:
	if (w) for (int x = 0; x < n_dev; x++) { free(w[x].q); free(w[x].p); free(w[x].sik); }
	if (tab) for (int x = 0; x < n_gpus; x++) free(tab[x]);
	free(w); free(th); free(joinable); free(res); free(tab); free(count_K); free(starts);
	return rc;
}

static int shardable(const mc_cli_options *o)
{
	/* MC_FORCE_SHARDED=1 sends even --gpus 1 through the sharded path (threads, jump-ahead, RCCL exchange, replay): the
	 * single-GPU rehearsal used by tests/test_gpu_cli.py */
	const int force = getenv("MC_FORCE_SHARDED") != NULL;
	return (o->n_gpus > 1 || o->n_streams > 1 || (force && o->n_gpus == 1)) && o->em.admixture && !o->target_revisit && !o->target_ll && !o->em.n_seconds &&
	       !(o->pfile && o->qfile);	/* initial parameters from files: every unit would be the same fit */
}

/* estimate_model (multiclust.c:365-452): K = min_K..max_K, or H0 / HA when bootstrapping */
/* sim != NULL: the models are fitted to the bootstrap data set it describes, generated on each device (admixture) */
static int estimate_model(const mc_cli_options *o, const mc_cli_data *d, const mc_data *md, run_state *st, int bootstrap, int *total_iter,
			  const mc_simulation *sim)
{
	int K = o->n_bootstrap ? st->null_K : o->min_K, rc = 0;
	double min_aic = INFINITY, min_bic = INFINITY;
	const clock_t start = clock();
	st->sum.max_logL = -INFINITY;
	st->max_logL_H0 = -INFINITY;
	if (total_iter) *total_iter = 0;
	for (;;) {
		if (shardable(o)) {
			rc = maximize_likelihood_sharded(o, d, md, K, st, bootstrap, sim);
		} else {
			mc_model *mod = NULL;
			mc_model **slot = (sim && st->sim_models) ? &st->sim_models[K == st->alt_K ? 1 : 0] : NULL;
			if (slot && K == st->alt_K && st->alt_K != st->null_K && st->sim_models[0] && (!*slot || (*slot)->K == K)) {
				/* the alternative model takes the data set the null model of this replicate was just fitted to */
				if ((rc = mc_model_share_simulated(slot, &o->em, md, K, o->device, st->sim_models[0]))) return rc;
				mod = *slot;
			} else if (slot && *slot && (*slot)->K == K) {
				mod = *slot;
				if ((rc = mc_model_resimulate(mod, &o->em, md, sim))) return rc;
			} else {
				if ((rc = sim ? mc_model_create_simulated(&mod, &o->em, md, K, o->device, sim)
					      : mc_model_create(&mod, &o->em, md, K, o->device))) return rc;
				if (slot) { mc_model_free(*slot); *slot = mod; }
			}
			rc = maximize_likelihood(o, d, md, mod, st, bootstrap);
			if (!slot) mc_model_free(mod);
		}
		if (rc) return rc;
		if (o->n_repeat == 1 && o->em.verbosity)
			print_model_state(o, d, st, K, (int)(((double)clock() - start) / CLOCKS_PER_SEC), 1);
		if (total_iter) *total_iter += st->sum.n_total_iter;
		if (o->n_bootstrap && K == st->null_K) st->max_logL_H0 = st->sum.max_logL;
		if (min_aic > st->sum.aic) { min_aic = st->sum.aic; st->aic_K = K; }
		if (min_bic > st->sum.bic) { min_bic = st->sum.bic; st->bic_K = K; }
		if (o->n_bootstrap && K == st->null_K) K = st->alt_K;
		else if (!o->n_bootstrap && K < o->max_K) K++;
		else break;
	}
	if (o->n_bootstrap) {
		const double diff = st->sum.max_logL - st->max_logL_H0;
		if (diff <= 0) {
			fprintf(stderr, "ERROR [mc_main.c::estimate_model]: Null hypothesis likelihood exceeds alternative hypothesis likelihood.  "
				"Try increasing number of initializations (command-line option -n)\n");
			return MC_EXIT_INTERNAL_ERROR;		/* multiclust.c:437-443 */
		}
		if (!bootstrap) st->ts_obs = diff; else st->ts_bs = diff;
	}
	return 0;
}

/* ---- bootstrap replicates sharded over the GPUs of the node as whole units ----
 * Replicate b = data set + H0 fits + HA fits consumes a data-independent number of rand() draws, so device x takes
 * replicates x, x + n_dev, ... each from the stream jumped to where the serial program would be, generates the data
 * set on its own GPU and fits both models there; its stdout goes to a buffer.  One RCCL all-reduce completes the table
 * of test statistics on every device; the buffers are then printed in replicate order with the running p-value. */
typedef struct bs_worker {
	const mc_cli_options *o;
	const mc_cli_data *d;
	const mc_data *md;
	const run_state *st;		/* observed-data results: null_K, alt_K, H0 MLEs, ts_obs */
	int index, n_dev;
	uint64_t draws_per_replicate;
	double *ts;			/* [n_bootstrap], shared: worker x writes entries b = x, x + n_dev, ... */
	char **text;			/* [n_bootstrap] captured stdout of each replicate */
	int rc;
} bs_worker;

static void *bs_main(void *arg)
{
	bs_worker *w = arg;
	mc_cli_options ow = *w->o;
	ow.device = worker_device(w->o, w->index);
	ow.n_gpus = 0;			/* the fits of a replicate stay on this device, on this worker's stream */
	ow.n_streams = 1;
	mc_model *kept[2] = { NULL, NULL };
	int *own_I_K = malloc(sizeof(int) * (size_t)(w->d->I > 0 ? w->d->I : 1));	/* -A: the partition record is per worker */
	if (!own_I_K) { w->rc = MCHIP_ERR_ALLOC; return NULL; }
	memcpy(own_I_K, w->st->I_K, sizeof(int) * (size_t)w->d->I);
	for (int b = w->index; b < w->o->n_bootstrap; b += w->n_dev) {
		run_state ls = *w->st;
		ls.sim_models = kept;
		ls.I_K = own_I_K;
		mc_simulation gen;
		size_t len = 0;
		ls.mle_q = w->st->mle_q; ls.mle_p = w->st->mle_p;	/* read only */
		mc_rng_jump(&ls.rng, (uint64_t)b * w->draws_per_replicate);
		if (!(ls.out = open_memstream(&w->text[b], &len))) { w->rc = MCHIP_ERR_ALLOC; free(own_I_K); return NULL; }
		fprintf(ls.out, "Bootstrap dataset %d (of %d):", b + 1, w->o->n_bootstrap);
		mc_simulation_begin(&gen, &ow.em, w->md, ls.mle_K, ls.mle_q, ls.mle_p, &ls.rng);
		w->rc = estimate_model(&ow, w->d, w->md, &ls, 1, NULL, &gen);
		fclose(ls.out);
		if (w->rc) break;
		w->ts[b] = ls.ts_bs;
	}
	mc_model_free(kept[0]);
	mc_model_free(kept[1]);
	free(own_I_K);
	return NULL;
}

static int run_bootstrap_sharded(const mc_cli_options *o, const mc_cli_data *d, const mc_data *md, run_state *st, int *ntime_out)
{
	const int n_dev = n_workers(o), n_gpus = o->n_gpus < 1 ? 1 : o->n_gpus, B = o->n_bootstrap;
	const uint64_t per_init = mc_draws_per_init(&o->em, md, st->alt_K);
	const uint64_t units0 = st->null_K == 1 ? 1 : (uint64_t)o->n_init, units1 = (uint64_t)o->n_init;
	bs_worker *w = calloc((size_t)n_dev, sizeof *w);
	pthread_t *th = calloc((size_t)n_dev, sizeof *th);
	int *joinable = calloc((size_t)n_dev, sizeof *joinable);
	double *ts = calloc((size_t)B, sizeof *ts), **tab = calloc((size_t)n_gpus, sizeof *tab);
	char **text = calloc((size_t)B, sizeof *text);
	int rc = 0, ntime = 0;
	if (!w || !th || !joinable || !ts || !tab || !text) { rc = MCHIP_ERR_ALLOC; goto DONE; }
	for (int x = 0; x < n_dev; x++) {
		w[x].o = o; w[x].d = d; w[x].md = md; w[x].st = st; w[x].index = x; w[x].n_dev = n_dev;
		w[x].draws_per_replicate = mc_bootstrap_draws(&o->em, md) + (units0 + units1) * per_init;
		w[x].ts = ts; w[x].text = text;
		if (pthread_create(&th[x], NULL, bs_main, &w[x])) bs_main(&w[x]);
		else joinable[x] = 1;
	}
	for (int x = 0; x < n_dev; x++) if (joinable[x]) pthread_join(th[x], NULL);
	for (int x = 0; x < n_dev; x++) if (w[x].rc) rc = w[x].rc;
	if (rc) goto DONE;
	mc_rng_jump(&st->rng, (uint64_t)B * w[0].draws_per_replicate);
	/* the one exchange: rows (test statistic, fitted flag) of the replicates each device's workers own, summed over devices */
	for (int g = 0; g < n_gpus; g++)
		if (!(tab[g] = calloc((size_t)B * 2, sizeof(double)))) { rc = MCHIP_ERR_ALLOC; goto DONE; }
	for (int b = 0; b < B; b++) {
		double *row = tab[(b % n_dev) % n_gpus] + 2 * (size_t)b;
		row[0] = ts[b];
		row[1] = 1.0;
	}
	if (exchange_needed(o) && (rc = exchange_table(o, st, tab, B * 2, "bootstrap test statistics"))) goto DONE;
	for (int b = 0; b < B; b++) {
		if (tab[0][2 * b + 1] != 1.0) { fprintf(stderr, "ERROR [mc_main.c]: bootstrap replicate %d was fitted %g times\n", b, tab[0][2 * b + 1]); rc = MCHIP_ERR_STATE; goto DONE; }
		st->ts_bs = tab[0][2 * b];
		if (st->ts_bs >= st->ts_obs) ntime++;
		fputs(text[b] ? text[b] : "", stdout);
		printf(" test statistics bs=%f obs=%f (%f)\n", st->ts_bs, st->ts_obs, (double)ntime / (b + 1));
	}
	*ntime_out = ntime;
DONE:
	if (tab) for (int x = 0; x < n_gpus; x++) free(tab[x]);
	if (text) for (int b = 0; b < B; b++) free(text[b]);
	free(w); free(th); free(joinable); free(ts); free(tab); free(text);
	return rc;
}

int main(int argc, const char **argv)
{
	mc_cli_options o;
	mc_cli_data d;
	run_state st;
	int rc;
	/* Every line leaves the process when it is complete, also into a pipe (same bytes as the reference's, earlier): a run that
	 * stops making progress has then said how far it got.  MC_WATCHDOG_S=<s>: leave with status 3 and a report of where every
	 * thread stands once nothing has happened for s seconds (mc_watchdog.c). */
	setvbuf(stdout, NULL, _IOLBF, 0);
	mc_watchdog_from_env();
	mchip_progress_note("parse_options");
	defaults(&o);
	if ((rc = parse_options(&o, argc, argv))) return rc;	/* -h included: the reference's -h leaves with status 1 */
	mchip_progress_note("mc_read_structure");
	if ((rc = mc_read_structure(&o, &d))) return rc;
	mchip_progress_note("estimate_model");
	if (o.em.verbosity >= MC_TALKATIVE)
		fprintf(stderr, "INFO: Finished reading data: %d %d-ploid individuals at %d loci.\n", d.I, d.ploidy, d.L);
	mc_data md = { d.I, d.L, d.ploidy, d.uniquealleles, d.geno, NULL };
	/* synchronize (multiclust.c:807-893) */
	if (mc_synchronize(&o.em, &md)) return MC_EXIT_INVALID_USER_SETUP;
	if (d.I < o.max_K) { fprintf(stderr, "ERROR: Maximum number of clusters (%d) (set with command-line argument -k) cannot exceed the number of individuals (%d)\n", o.max_K, d.I); return MC_EXIT_INVALID_USER_SETUP; }
	if (o.n_bootstrap && o.max_K <= 1) { fprintf(stderr, "ERROR: When bootstrapping, maximum K (%d) (set with command-line argument -k) must exceed 1.\n", o.max_K); return MC_EXIT_INVALID_USER_SETUP; }
	if (o.min_K > o.max_K) { fprintf(stderr, "ERROR: Minimum K (%d) must not exceed maximum K (%d).\n", o.min_K, o.max_K); return MC_EXIT_INVALID_USER_SETUP; }
	if (!o.target_ll && !o.target_revisit && !o.em.n_seconds && !o.n_init) o.n_init = 1;
	if (!o.em.n_rand_em_init) o.em.initialization_procedure = MC_INIT_NOTHING;	/* -m 0 (multiclust.c:1549-1550) */
	memset(&st, 0, sizeof st);
	if (!(st.I_K = calloc((size_t)d.I, sizeof *st.I_K))) return MC_EXIT_MEMORY_ALLOCATION;
	if (o.afile && (rc = mc_read_afile(o.afile, d.I, &st.partition_from_file, &st.pK))) return rc;	/* synchronize's last step (multiclust.c:889-890) */
	st.out = stdout;
	mchip_comm *run_comm = NULL;
	st.comm = &run_comm;
	if (o.n_bootstrap) { st.null_K = o.max_K - 1; st.alt_K = o.max_K; }
	/* the reference seeds libc only when -r is given; otherwise rand() runs from glibc's default seed 1 although the
	 * banner prints 1234567 */
	mc_srand(&st.rng, o.seed_given ? o.em.seed : 1u);
	if ((o.n_gpus > 1 || o.n_streams > 1) && !shardable(&o))
		fprintf(stderr, "WARNING: --gpus / --streams apply to the admixture model with a fixed number of initialisations; running one fit at a time on one GPU\n");

	if (o.n_repeat > 1 || o.repeat_seconds) {	/* timed_model_estimation (multiclust.c:201-347): same lines, same statistics */
		const clock_t start = clock();
		double sum_ll = 0, sum_ll2 = 0, sum_init = 0, sum_init2 = 0, sum_iter = 0, sum_iter2 = 0, esec = 0;
		double sum_aic_K = 0, sum_aic_K2 = 0, sum_bic_K = 0, sum_bic_K2 = 0;
		double max_ll = -INFINITY, min_aic = 0, min_bic = 0, first_ll = -INFINITY;
		double max_ar = -1, max_ll_rand = 0, sum_ar = 0, sum_ar2 = 0;	/* adjusted Rand index (-A); without it model::arand stays 0 */
		int n = 0, conv = 0, reached = 0, first_hit = 0, max_init = 0, max_iter = 0, enough = o.repeat_seconds ? 0 : 1, total;
		char ab[16];
		while (n < o.n_repeat || !enough) {
			if ((rc = estimate_model(&o, &d, &md, &st, 0, &total, NULL))) goto END;
			if (st.sum.n_init > max_init) max_init = st.sum.n_init;
			if (st.sum.n_max_iter > max_iter) max_iter = st.sum.n_max_iter;
			if (st.sum.max_logL > max_ll) {
				mc_model probe;
				max_ll = st.sum.max_logL;
				min_aic = st.sum.aic;
				min_bic = st.sum.bic;
				max_ll_rand = st.arand;
				memset(&probe, 0, sizeof probe);
				probe.logL = st.sum.max_logL;	/* converged(opt, mod, first_ll) compares with the model's current logL */
				if (!mc_converged(&o.em, &probe, first_ll)) { first_ll = st.sum.max_logL; first_hit = n; }
			}
			if (st.arand > max_ar) max_ar = st.arand;	/* multiclust.c:252-253 */
			if (o.afile) { sum_ar += st.arand; sum_ar2 += st.arand * st.arand; }
			sum_init += st.sum.n_init; sum_init2 += (double)st.sum.n_init * st.sum.n_init;
			sum_iter += st.sum.n_total_iter; sum_iter2 += (double)st.sum.n_total_iter * st.sum.n_total_iter;
			sum_aic_K += st.aic_K; sum_aic_K2 += (double)st.aic_K * st.aic_K;
			sum_bic_K += st.bic_K; sum_bic_K2 += (double)st.bic_K * st.bic_K;
			sum_ll += st.sum.max_logL; sum_ll2 += st.sum.max_logL * st.sum.max_logL;
			n++;
			if (st.sum.ever_converged) conv++;
			if (st.n_targetll_times) reached++;
			esec = ((double)clock() - start) / CLOCKS_PER_SEC;
			if (o.em.verbosity > MC_SILENT) {
				print_model_state(&o, &d, &st, o.max_K, (int)esec, 0);
				printf(" %f %f %d %d %f", esec, esec / n, reached, conv, max_ll);
				if (o.target_ll) printf(" %f", o.desired_ll); else printf(" NA");
				printf(" %d %d %d %u\n", o.target_revisit, n, o.n_repeat, o.repeat_seconds);
			}
			if (!enough && esec > o.repeat_seconds) enough = 1;
			if (o.max_repeat_seconds && esec > o.max_repeat_seconds) break;
		}
		if (o.em.verbosity >= MC_SILENT) {
			printf("Data, Method, Model: %s, %s, %s\n", o.filename, accel_abbrev(&o, ab),
			       o.em.admixture && o.em.eta_constrained ? "admix constrained" : o.em.admixture ? "admix" : "mix");
			printf("Run: %e %e %e %e n=%d i=%d u=(%f,%d) w=(%d,%u)\n", o.em.abs_error, o.em.rel_error, o.em.eta_lower_bound,
			       o.em.p_lower_bound, o.n_init, o.em.n_init_iter, o.desired_ll, o.target_revisit, o.n_repeat, o.repeat_seconds);
			printf("Number of repetitions: %d of %d requested, %d converged, %d reach target\n", n, o.n_repeat, conv, reached);
			printf("Average time: %fs (total: %fs; target: %u)\n", esec / n, esec, o.repeat_seconds);
			printf("Average log likelihood: %f (+/- %f)\n", sum_ll / n, sqrt((sum_ll2 - sum_ll * sum_ll / n) / (n - 1)));
			printf("Maximum log likelihood: %f first hit at run %d (AIC %f; BIC %f; RAND: %f)\n", max_ll, first_hit, min_aic, min_bic, max_ll_rand);
			printf("Adjusted RAND: avg = %f +/- %f; max = %f\n", sum_ar / n, sqrt((sum_ar2 - sum_ar * sum_ar / n) / (n - 1)), max_ar);
			if (o.max_K != o.min_K) {
				printf("Average K (AIC): %f (+/- %f)\n", sum_aic_K / n, sqrt((sum_aic_K2 - sum_aic_K * sum_aic_K / n) / (n - 1)));
				printf("Average K (BIC): %f (+/- %f)\n", sum_bic_K / n, sqrt((sum_bic_K2 - sum_bic_K * sum_bic_K / n) / (n - 1)));
			} else {
				printf("Total initializations, iterations: %d, %d\n", (int)sum_init, (int)sum_iter);
				printf("Average initializations: %f (+/- %f) [%e, %e]\n", sum_init / n,
				       sqrt((sum_init2 - sum_init * sum_init / n) / (n - 1)), sum_init2, sum_init);
				printf("Average iterations: %f (+/- %f) [%e %e]\n", sum_iter / sum_init,
				       sqrt((sum_iter2 - sum_iter * sum_iter / sum_init) / (sum_init - 1)), sum_iter2, sum_iter);
				printf("Maximum initializations: %d\n", max_init);
				printf("Maximum iterations: %d\n", max_iter);
			}
		}
	} else if ((rc = estimate_model(&o, &d, &md, &st, 0, NULL, NULL))) {
		goto END;
	}
	if (o.parallel) printf("%f\n", st.sum.max_logL);	/* multiclust.c:143-145 */

	if (o.n_bootstrap) {	/* run_bootstrap (multiclust.c:675-708) */
		/* admixture: the replicate is generated on the device(s) from the stream position (no host data set, no upload);
		 * mixture (or MC_HOST_BOOTSTRAP): drawn on the host and uploaded like any data set */
		const int on_device = o.em.admixture && !getenv("MC_HOST_BOOTSTRAP");
		uint8_t *orig = d.geno, *sim = on_device ? NULL : malloc((size_t)d.I * d.L * d.ploidy);
		int ntime = 0;
		if ((!on_device && !sim) || !st.mle_q) { rc = MCHIP_ERR_ALLOC; goto END; }
		/* whole replicates per device when there is at least one for each; otherwise (or on one device) the replicates run in
		 * turn and --gpus shards the initialisations inside each */
		/* (Rand-EM draws a data-dependent number of values per initialisation: replicate b's place in the stream has no closed form) */
		const int by_replicate = on_device && shardable(&o) && o.n_bootstrap >= n_workers(&o) && o.em.initialization_procedure != MC_RAND_EM;
		if (by_replicate && (rc = run_bootstrap_sharded(&o, &d, &md, &st, &ntime))) goto END;
		mc_model *kept[2] = { NULL, NULL };
		if (on_device && !shardable(&o)) st.sim_models = kept;
		for (int b = 0; !by_replicate && b < o.n_bootstrap; b++) {
			printf("Bootstrap dataset %d (of %d):", b + 1, o.n_bootstrap);
			if (on_device) {
				mc_simulation gen;
				mc_simulation_begin(&gen, &o.em, &md, st.mle_K, st.mle_q, st.mle_p, &st.rng);
				rc = estimate_model(&o, &d, &md, &st, 1, NULL, &gen);
			} else {
				mc_bootstrap_genotypes(&o.em, &md, st.mle_K, st.mle_q, st.mle_p, &st.rng, sim);
				md.geno = d.geno = sim;
				md.init_geno = orig;
				rc = estimate_model(&o, &d, &md, &st, 1, NULL, NULL);
				md.geno = d.geno = orig;
				md.init_geno = NULL;
			}
			if (rc) { free(sim); mc_model_free(kept[0]); mc_model_free(kept[1]); goto END; }
			if (st.ts_bs >= st.ts_obs) ntime++;
			printf(" test statistics bs=%f obs=%f (%f)\n", st.ts_bs, st.ts_obs, (double)ntime / (b + 1));
		}
		free(sim);
		st.sim_models = NULL;
		mc_model_free(kept[0]);
		mc_model_free(kept[1]);
		/* the reference divides two ints here (multiclust.c:703); kept */
		printf("p-value to reject H0: K=%d is %f\n", st.null_K, (double)(ntime / o.n_bootstrap));
	}
END:
	mchip_progress_note("end of main: freeing");
	if (run_comm) mchip_comm_destroy(run_comm);
	free(st.mle_q); free(st.mle_p); free(st.I_K); free(st.partition_from_file);
	mc_free_data(&d);
	mchip_progress_note("returning from main: the HIP runtime's own teardown follows");
	return rc;
}
