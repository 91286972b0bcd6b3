/*
 * ref_harness.c -- TEST INFRASTRUCTURE ONLY (oracle side, never shipped, never on the product path).
 *
 * Drives the *unmodified* reference sources where they lie under /root/reference (compiled by
 * oracle/Makefile into oracle/_ref/, multiclust.c with -Dmain=ref_main) and dumps golden vectors at
 * full precision (raw little-endian doubles / int32 / uint8 + a JSON manifest) for tests/golden/.
 *
 * It only *calls* reference functions (all have external linkage): make_options/make_data/make_model,
 * parse_options, read_file, synchronize, allocate_model_for_k (multiclust.c:902,1007,1087,1396,807,1181),
 * initialize_model (rnd_init.c:54), em_step/em/em_2_steps (em_alg.c:195,44,1072), log_likelihood
 * (log_likelihood.c:56), step_size/accelerated_update/accelerated_em_step (accel_em.c:130,422,35),
 * michelot_project (simplex.c:109).  The traced SQUAREM cycle below re-sequences those calls in the
 * order accel_em.c:35-114 uses them so that emll / s / ll / accept can be recorded; the untraced
 * reference em() is run beside it and both must agree (checked here, abort on mismatch).
 *
 * usage: ref_harness <outdir> <n_em_steps> <snap,csv> <n_accel_cycles> -- <multiclust argv...>
 */
#define _POSIX_C_SOURCE 200809L	/* strdup under -std=c17 */
#include "multiclust.h"
#include <stdint.h>
#include <limits.h>

/* reference functions without a prototype in multiclust.h */
int make_options(options **opt);
int make_data(data **dat);
int make_model(model **mod);
int parse_options(options *opt, data *dat, int argc, const char **argv);
int allocate_model_for_k(options *opt, model *mod, data *dat);
int synchronize(options *opt, data *dat, model *mod);
void free_model_data(model *mod, options *opt);
double step_size(options *opt, data *dat, model *mod);
double accelerated_update(options *opt, data *dat, model *mod, double s);
double qn_accelerated_update(options *opt, data *dat, model *mod);
void michelot_project(double *, int, double, double);
int maximize_likelihood(options *opt, data *dat, model *mod, int bootstrap);
void random_allele_center(data *dat, model *mod);
void random_individual_center(data *dat, model *mod);
void random_individual_partition(data *dat, model *mod);
void initialize_parameters_admixture(options *opt, data *dat, model *mod);
void initialize_parameters_mixture(data *dat, model *mod);

static const char *outdir;
static FILE *man;
static int man_first = 1;

static void die(const char *msg) { fprintf(stderr, "ref_harness: %s\n", msg); exit(2); }

static FILE *xopen(const char *name)
{
	char path[4096];
	snprintf(path, sizeof path, "%s/%s", outdir, name);
	FILE *f = fopen(path, "wb");
	if (!f) die("cannot open output file");
	return f;
}

static void man_key(const char *k)
{
	fprintf(man, "%s\n  \"%s\": ", man_first ? "" : ",", k);
	man_first = 0;
}
static void man_int(const char *k, long v) { man_key(k); fprintf(man, "%ld", v); }
static void man_dbl(const char *k, double v)
{
	man_key(k);
	if (isnan(v)) fprintf(man, "\"nan\"");
	else if (isinf(v)) fprintf(man, v > 0 ? "\"inf\"" : "\"-inf\"");
	else fprintf(man, "%.17g", v);
}
static void man_hex(const char *k, double v) { man_key(k); fprintf(man, "\"%a\"", v); }

static int T_total(data *dat)
{
	int T = 0;
	for (int l = 0; l < dat->L; l++) T += dat->uniquealleles[l];
	return T;
}

static void dump_p(const char *name, data *dat, model *mod, int slot)
{
	FILE *f = xopen(name);
	for (int k = 0; k < mod->K; k++)
		for (int l = 0; l < dat->L; l++)
			fwrite(mod->vpklm[slot][k][l], sizeof(double), dat->uniquealleles[l], f);
	fclose(f);
}

static void dump_q(const char *name, options *opt, data *dat, model *mod, int slot)
{
	FILE *f = xopen(name);
	if (opt->admixture && !opt->eta_constrained)
		for (int i = 0; i < dat->I; i++)
			fwrite(mod->vetaik[slot][i], sizeof(double), mod->K, f);
	else
		fwrite(mod->vetak[slot], sizeof(double), mod->K, f);
	fclose(f);
}

/* sum_{l,m} diklm[i][k][l][m] in the reference M-step's order (em_alg.c:652-676); this is what the
 * writers consume (write_file.c:359-381,446-459,531-542). For the mixture model dump vik instead. */
static void dump_sik(const char *name, options *opt, data *dat, model *mod)
{
	FILE *f = xopen(name);
	for (int i = 0; i < dat->I; i++)
		for (int k = 0; k < mod->K; k++) {
			double e = 0;
			if (opt->admixture) {
				for (int l = 0; l < dat->L; l++)
					for (int m = 0; m < dat->uniquealleles[l]; m++)
						e += mod->diklm[i][k][l][m];
			} else {
				e = mod->vik[i][k];
			}
			fwrite(&e, sizeof e, 1, f);
		}
	fclose(f);
}

static void reset_model_state(options *opt, model *mod)
{
	/* what maximize_likelihood() does before each initialisation (multiclust.c:518-524) */
	mod->current_i = mod->current_l = mod->current_k = 0;
	mod->logL = 0.0;
	mod->converged = 0;
	mod->stopped = 0;
	mod->iter_stop = 0;
	mod->time_stop = 0;
	mod->accel_step = 0;
	mod->delta_index = 0;
	mod->pindex = mod->findex = mod->tindex = 0;
	mod->start = clock();
	srand(opt->seed);
}

int main(int argc, const char **argv)
{
	options *opt = NULL;
	data *dat = NULL;
	model *mod = NULL;
	int err, sep = -1;
	char name[256];

	if (argc < 7) die("usage: ref_harness <outdir> <n_em_steps> <snap,csv> <n_accel_cycles> -- <args>");
	outdir = argv[1];
	int n_em = atoi(argv[2]);
	const char *snapcsv = argv[3];
	int n_cycles = atoi(argv[4]);
	for (int i = 5; i < argc; i++)
		if (!strcmp(argv[i], "--")) { sep = i; break; }
	if (sep < 0) die("missing --");

	if ((err = make_options(&opt)) || (err = make_data(&dat)) || (err = make_model(&mod)))
		die("make_* failed");
	/* argv[sep] plays the role of argv[0] for parse_options */
	if ((err = parse_options(opt, dat, argc - sep, argv + sep))) die("parse_options failed");
	opt->write_files = 0;
	opt->verbosity = 1;		/* QUIET */
	if ((err = read_file(opt, dat))) die("read_file failed");
	if ((err = synchronize(opt, dat, mod))) die("synchronize failed");
	mod->K = opt->max_K;
	dat->max_M = dat->M > mod->K ? dat->M : mod->K;
	if ((err = allocate_model_for_k(opt, mod, dat))) die("allocate_model_for_k failed");

	/* The phantom trailing allele slot created by missing data is uninitialised memory in the
	 * reference (read_file.c:527-533 vs 581-585). Pin it to a code that can never match, after
	 * verifying its counts are zero, so the golden vectors do not depend on heap garbage. */
	int T = T_total(dat);
	int *n_real = malloc(dat->L * sizeof *n_real);
	for (int l = 0; l < dat->L; l++) {
		int has_missing = 0, M = dat->uniquealleles[l];
		for (int j = 0; j < dat->I * dat->ploidy; j++)
			if (dat->IL[j][l] == MISSING) has_missing = 1;
		n_real[l] = M - (has_missing && dat->L_alleles ? 1 : 0);
		if (has_missing && dat->L_alleles && M > 0) {
			for (int i = 0; i < dat->I; i++)
				if (dat->ILM[i][l][M - 1] != 0) die("phantom allele slot has counts (heap garbage matched an allele); rerun");
			dat->L_alleles[l][M - 1] = INT_MIN;
		}
	}

	snprintf(name, sizeof name, "%s/manifest.json", outdir);
	man = fopen(name, "w");
	if (!man) die("cannot open manifest");
	fprintf(man, "{");
	man_int("I", dat->I); man_int("L", dat->L); man_int("ploidy", dat->ploidy);
	man_int("K", mod->K); man_int("T", T); man_int("M", dat->M);
	man_int("admixture", opt->admixture); man_int("eta_constrained", opt->eta_constrained);
	man_int("accel_scheme", opt->accel_scheme); man_int("q", opt->q);
	man_int("do_projection", opt->do_projection);
	man_int("missing_data", dat->missing_data);
	man_int("seed", opt->seed);
	man_hex("lower_bound_hex", opt->lower_bound); man_dbl("lower_bound", opt->lower_bound);
	man_dbl("abs_error", opt->abs_error); man_dbl("rel_error", opt->rel_error);
	man_int("max_iter", opt->max_iter);
	man_int("adjust_step", opt->adjust_step);
	man_int("no_parameters", mod->no_parameters);

	/* ---- data ---- */
	FILE *f = xopen("uniquealleles.i32");
	fwrite(dat->uniquealleles, sizeof(int), dat->L, f); fclose(f);
	f = xopen("geno.u8");
	for (int i = 0; i < dat->I; i++)
		for (int l = 0; l < dat->L; l++)
			for (int a = 0; a < dat->ploidy; a++) {
				int v = dat->IL[dat->ploidy * i + a][l];
				uint8_t idx = 0xFF;
				if (v != MISSING) {
					int found = -1;
					for (int m = 0; m < n_real[l]; m++)
						if (dat->L_alleles[l][m] == v) { found = m; break; }
					if (found < 0 || found > 254) die("allele not found in L_alleles");
					idx = (uint8_t)found;
				}
				fwrite(&idx, 1, 1, f);
			}
	fclose(f);
	f = xopen("ilm.i32");
	for (int i = 0; i < dat->I; i++)
		for (int l = 0; l < dat->L; l++)
			fwrite(dat->ILM[i][l], sizeof(int), dat->uniquealleles[l], f);
	fclose(f);
	f = xopen("locale.i32");
	for (int i = 0; i < dat->I; i++) fwrite(&dat->idv[i].locale, sizeof(int), 1, f);
	fclose(f);
	man_int("numpops", dat->numpops);

	/* ---- libc rand() known answers for this seed (glibc TYPE_3) ---- */
	srand(opt->seed);
	man_key("rand_first"); fprintf(man, "[");
	for (int i = 0; i < 8; i++) fprintf(man, "%s%d", i ? ", " : "", rand());
	fprintf(man, "]");
	man_int("RAND_MAX", RAND_MAX);

	/* ---- section 1: initialisation + n plain EM steps (findex=tindex=0, in place) ---- */
	int accel_saved = opt->accel_scheme, q_saved = opt->q;
	opt->q = 1;		/* plain EM: options::q is 1 unless a QN scheme is selected (multiclust.c:950,820) */
	{
		int *snaps = calloc(n_em + 2, sizeof *snaps);
		char *csv = strdup(snapcsv), *tok;
		for (tok = strtok(csv, ","); tok; tok = strtok(NULL, ",")) {
			int s = atoi(tok);
			if (s >= 1 && s <= n_em) snaps[s] = 1;
		}
		opt->accel_scheme = 0;
		reset_model_state(opt, mod);
		if ((err = initialize_model(opt, dat, mod))) die("initialize_model failed");
		man_key("rand_after_init"); fprintf(man, "%d", rand());	/* stream position check */
		dump_q("q0.f64", opt, dat, mod, 0);
		dump_p("p0.f64", dat, mod, 0);
		if (opt->admixture) dump_sik("sik_init.f64", opt, dat, mod);	/* hard-partition counts */
		double saved_abs = opt->abs_error;
		opt->abs_error = 0; opt->rel_error = 0;	/* keep stepping: converged() then always "stops" but we ignore it */
		f = xopen("em_ll.f64");
		for (int s = 1; s <= n_em; s++) {
			em_step(opt, dat, mod);
			double ll = mod->logL;	/* stop() stored the E-step value (em_alg.c:140) */
			fwrite(&ll, sizeof ll, 1, f);
			if (snaps[s]) {
				snprintf(name, sizeof name, "q_step%d.f64", s); dump_q(name, opt, dat, mod, 0);
				snprintf(name, sizeof name, "p_step%d.f64", s); dump_p(name, dat, mod, 0);
				snprintf(name, sizeof name, "sik_step%d.f64", s); dump_sik(name, opt, dat, mod);
			}
		}
		fclose(f);
		opt->abs_error = saved_abs;
		man_int("n_em_steps", n_em);
		man_key("snapshots"); fprintf(man, "[");
		for (int s = 1, first = 1; s <= n_em; s++)
			if (snaps[s]) { fprintf(man, "%s%d", first ? "" : ", ", s); first = 0; }
		fprintf(man, "]");
		man_dbl("ll_after_em", log_likelihood(opt, dat, mod, 0));
		man_hex("ll_after_em_hex", log_likelihood(opt, dat, mod, 0));
		free(snaps); free(csv);
	}

	/* ---- section 2: full plain-EM run to convergence with the reference's own em() ---- */
	{
		opt->accel_scheme = 0;
		reset_model_state(opt, mod);
		initialize_model(opt, dat, mod);
		em(opt, dat, mod);
		man_int("em_run_n_iter", mod->n_iter);
		man_int("em_run_converged", mod->converged);
		man_dbl("em_run_logL", mod->logL); man_hex("em_run_logL_hex", mod->logL);
		dump_q("q_emrun.f64", opt, dat, mod, mod->pindex);
		dump_p("p_emrun.f64", dat, mod, mod->pindex);
		dump_sik("sik_emrun.f64", opt, dat, mod);
	}

	/* ---- section 3: accelerated run (only if -s given) ---- */
	opt->accel_scheme = accel_saved;
	opt->q = q_saved;
	if (opt->accel_scheme && mod->K > 1) {
		/* 3a: untraced reference em() */
		reset_model_state(opt, mod);
		initialize_model(opt, dat, mod);
		em(opt, dat, mod);
		int ref_n_iter = mod->n_iter, ref_conv = mod->converged, ref_pindex = mod->pindex;
		double ref_logL = mod->logL;
		man_int("accel_run_n_iter", ref_n_iter);
		man_int("accel_run_converged", ref_conv);
		man_int("accel_run_pindex", ref_pindex);
		man_dbl("accel_run_logL", ref_logL); man_hex("accel_run_logL_hex", ref_logL);
		dump_q("q_accelrun.f64", opt, dat, mod, mod->pindex);
		dump_p("p_accelrun.f64", dat, mod, mod->pindex);
		dump_sik("sik_accelrun.f64", opt, dat, mod);

		/* 3b: traced cycles: the call sequence of em() (em_alg.c:61-88) and accelerated_em_step()
		 * (accel_em.c:35-114) with recording. */
		reset_model_state(opt, mod);
		initialize_model(opt, dat, mod);
		int stop_flag = 0, cyc = 0;
		while (mod->n_iter < opt->n_init_iter && !stop_flag)
			stop_flag = em_step(opt, dat, mod);
		for (int i = 1; i < opt->q; i++) {
			em_2_steps(mod, dat, opt);
			mod->pindex = mod->findex;
		}
		f = xopen("accel_trace.f64");
		FILE *fu = NULL;
		/* REF_HARNESS_CYCLE_STATES=<stride>: the iterate every cycle starts from (slot pindex), for cycles c with c % stride
		 * <= 1 (stride 1: every cycle), so that each cycle of the whole run can be checked from the reference's own state */
		const int st_stride = getenv("REF_HARNESS_CYCLE_STATES") ? atoi(getenv("REF_HARNESS_CYCLE_STATES")) : 0;
		FILE *fs = st_stride > 0 ? xopen("accel_states.f64") : NULL;
		/* quasi-Newton with q > 1 carries the last q secant pairs from cycle to cycle: with every recorded state also
		 * delta_index and u_j, v_j (eta part, then p part; j = 0..q-1) -> accel_secants.f64 */
		FILE *fq = (fs && opt->q > 1) ? xopen("accel_secants.f64") : NULL;
		int n_states = 0;
		if (!mod->converged) do {
			double rec[8] = {0};
			if (fs && (st_stride == 1 || cyc % st_stride <= 1)) {
				double c = cyc;
				fwrite(&c, sizeof c, 1, fs);
				if (opt->admixture && !opt->eta_constrained)
					for (int i = 0; i < dat->I; i++) fwrite(mod->vetaik[mod->pindex][i], sizeof(double), mod->K, fs);
				else
					fwrite(mod->vetak[mod->pindex], sizeof(double), mod->K, fs);
				for (int k = 0; k < mod->K; k++)
					for (int l = 0; l < dat->L; l++)
						fwrite(mod->vpklm[mod->pindex][k][l], sizeof(double), dat->uniquealleles[l], fs);
				n_states++;
				if (fq) {
					double di = mod->delta_index;
					fwrite(&c, sizeof c, 1, fq);
					fwrite(&di, sizeof di, 1, fq);
					for (int j = 0; j < opt->q; j++)
						for (int w = 0; w < 2; w++) {
							if (opt->admixture && !opt->eta_constrained)
								for (int i = 0; i < dat->I; i++)
									fwrite((w ? mod->v_etaik : mod->u_etaik)[j][i], sizeof(double), mod->K, fq);
							else
								fwrite((w ? mod->v_etak : mod->u_etak)[j], sizeof(double), mod->K, fq);
							for (int k = 0; k < mod->K; k++)
								for (int l = 0; l < dat->L; l++)
									fwrite((w ? mod->v_pklm : mod->u_pklm)[j][k][l], sizeof(double), dat->uniquealleles[l], fq);
						}
				}
			}
			em_2_steps(mod, dat, opt);
			if (mod->stopped) { stop_flag = 1; break; }
			double emll = log_likelihood(opt, dat, mod, mod->findex);
			double s = 0, ll = 0;
			int accepted = 0, valid = 1;
			if (opt->accel_scheme <= QN) {
				s = step_size(opt, dat, mod);
				if (isnan(s) || isinf(s)) valid = 0;
			}
			if (valid) {
				int n_adjust = 0;
				do {
					if (opt->accel_scheme <= QN)
						ll = accelerated_update(opt, dat, mod, s);
					else
						ll = qn_accelerated_update(opt, dat, mod);
					if (opt->adjust_step && ll < emll)
						s = (s - 1) / 2;
				} while (n_adjust++ < opt->adjust_step && ll < emll && s < -1);
				if (ll > emll) {
					mod->pindex = mod->tindex;
					mod->accel_step = 1;
					accepted = 1;
				}
			}
			if (!accepted)
				mod->pindex = mod->findex;
			rec[0] = emll; rec[1] = s; rec[2] = ll; rec[3] = accepted; rec[4] = mod->n_iter;
			rec[5] = mod->logL; rec[6] = mod->pindex; rec[7] = valid;
			fwrite(rec, sizeof(double), 8, f);
			cyc++;
			if (cyc == 1) {	/* first cycle internals: u, v increments and the accepted iterate */
				fu = xopen("accel_u_p.f64");
				for (int k = 0; k < mod->K; k++) for (int l = 0; l < dat->L; l++)
					fwrite(mod->u_pklm[0][k][l], sizeof(double), dat->uniquealleles[l], fu);
				fclose(fu);
				fu = xopen("accel_v_p.f64");
				for (int k = 0; k < mod->K; k++) for (int l = 0; l < dat->L; l++)
					fwrite(mod->v_pklm[0][k][l], sizeof(double), dat->uniquealleles[l], fu);
				fclose(fu);
				dump_q("q_cycle1.f64", opt, dat, mod, mod->pindex);
				dump_p("p_cycle1.f64", dat, mod, mod->pindex);
			}
			if (n_cycles > 0 && cyc == n_cycles) {
				snprintf(name, sizeof name, "q_cycle%d.f64", cyc); dump_q(name, opt, dat, mod, mod->pindex);
				snprintf(name, sizeof name, "p_cycle%d.f64", cyc); dump_p(name, dat, mod, mod->pindex);
			}
		} while (!stop_flag);
		fclose(f);
		if (fq) fclose(fq);
		if (fs) {
			fclose(fs);
			man_int("accel_states", n_states);
			man_int("accel_states_stride", st_stride);
		}
		man_int("accel_trace_cycles", cyc);
		man_int("accel_snapshot_cycle", n_cycles);
		if (mod->n_iter != ref_n_iter || mod->logL != ref_logL || mod->converged != ref_conv
			|| mod->pindex != ref_pindex)
			die("traced accelerated run disagrees with the reference's own em()");
	}

	/* ---- section 4: michelot_project known answers ---- */
	{
		uint64_t st = 0x9E3779B97F4A7C15ull ^ (uint64_t)opt->seed;
		int ncase = 64;
		FILE *fi = xopen("proj_in.f64"), *fo = xopen("proj_out.f64"), *fl = xopen("proj_len.i32");
		for (int c = 0; c < ncase; c++) {
			int len = 2 + (c % 7);
			double x[16] = {0}, sum = 0, min = (c & 1) ? opt->lower_bound : 1e-3;
			for (int j = 0; j < len; j++) {
				st = st * 6364136223846793005ull + 1442695040888963407ull;
				x[j] = (double)(st >> 11) / 9007199254740992.0;
				if (c % 3 == 0 && j == 0) x[j] = 0.0;		/* forces a clamp */
				if (c % 5 == 0 && j == 1) x[j] = -0.2;		/* negative entry (accelerated update) */
				sum += x[j];
			}
			if (c % 2 == 0) for (int j = 0; j < len; j++) x[j] /= sum;	/* near-normalised */
			fwrite(&len, sizeof len, 1, fl);
			fwrite(x, sizeof(double), 16, fi);
			fwrite(&min, sizeof(double), 1, fi);
			michelot_project(x, len, 1.0, min);
			fwrite(x, sizeof(double), 16, fo);
		}
		fclose(fi); fclose(fo); fclose(fl);
		man_int("proj_cases", ncase);
	}

	/* ---- section 5: several initialisations from ONE continuing rand() stream (maximize_likelihood,
	 * multiclust.c:471-656): per-initialisation results from our own loop of the same calls, and the
	 * reference's own bookkeeping summary from maximize_likelihood() itself ---- */
	if (mod->K > 1 && opt->admixture) {
		const int n_units = 6;
		int saved_n_init = opt->n_init;
		opt->accel_scheme = accel_saved;
		f = xopen("multi_init.f64");
		srand(opt->seed);
		for (int u = 0; u < n_units; u++) {
			double rec[4];
			mod->current_i = mod->current_l = mod->current_k = 0;
			mod->logL = 0.0; mod->converged = 0; mod->stopped = 0; mod->iter_stop = 0;
			mod->delta_index = 0;
			initialize_model(opt, dat, mod);
			em(opt, dat, mod);
			rec[0] = mod->logL; rec[1] = mod->converged; rec[2] = mod->n_iter; rec[3] = mod->pindex;
			fwrite(rec, sizeof(double), 4, f);
		}
		fclose(f);
		man_key("rand_after_multi_init"); fprintf(man, "%d", rand());
		srand(opt->seed);
		opt->n_init = n_units;
		mod->max_logL = -INFINITY;
		mod->delta_index = 0;
		maximize_likelihood(opt, dat, mod, 0);
		opt->n_init = saved_n_init;
		man_int("mi_units", n_units);
		man_int("mi_n_init", mod->n_init);
		man_int("mi_n_total_iter", mod->n_total_iter);
		man_int("mi_n_max_iter", mod->n_max_iter);
		man_int("mi_n_maxll_times", mod->n_maxll_times);
		man_int("mi_n_maxll_init", mod->n_maxll_init);
		man_int("mi_ever_converged", mod->ever_converged);
		man_dbl("mi_max_logL", mod->max_logL); man_hex("mi_max_logL_hex", mod->max_logL);
		man_dbl("mi_first_max_logL", mod->first_max_logL);
		man_dbl("mi_aic", mod->aic); man_dbl("mi_bic", mod->bic);
	}

	/* ---- section 6: parametric_bootstrap (bootstrap.c:31-175) known answers: one simulated data set drawn from the
	 * parameters reached after three EM steps (standing in for the H0 MLEs), rand() stream seeded with seed + 7 ---- */
	if (getenv("REF_HARNESS_BOOTSTRAP")) {
		opt->accel_scheme = 0;
		opt->q = 1;
		reset_model_state(opt, mod);
		if ((err = initialize_model(opt, dat, mod))) die("initialize_model failed");
		for (int s = 0; s < 3; s++) em_step(opt, dat, mod);
		dump_q("q_bs.f64", opt, dat, mod, 0);
		dump_p("p_bs.f64", dat, mod, 0);
		mod->mle_pKLM = malloc(mod->K * sizeof *mod->mle_pKLM);
		for (int k = 0; k < mod->K; k++) mod->mle_pKLM[k] = mod->vpklm[0][k];	/* read-only aliases */
		if (opt->admixture && !opt->eta_constrained) mod->mle_etaik = mod->vetaik[0];
		else mod->mle_etak = mod->vetak[0];
		srand(opt->seed + 7);
		if ((err = parametric_bootstrap(opt, dat, mod))) die("parametric_bootstrap failed");
		man_key("rand_after_bootstrap"); fprintf(man, "%d", rand());
		f = xopen("bs_ilm.u8");
		for (int i = 0; i < dat->I; i++)
			for (int l = 0; l < dat->L; l++)
				for (int m = 0; m < dat->uniquealleles[l]; m++) {
					uint8_t c = (uint8_t)dat->ILM[i][l][m];
					if (dat->ILM[i][l][m] < 0 || dat->ILM[i][l][m] > 255) die("bootstrap count out of range");
					fwrite(&c, 1, 1, f);
				}
		fclose(f);
		/* an initialisation while the bootstrap data set is in place (what run_bootstrap's fits start from): the random
		 * partition reads dat->IL, which parametric_bootstrap does not replace (rnd_init.c:471) */
		{
			double ***alias = mod->mle_pKLM;
			double **eta_alias = mod->mle_etaik;
			double *etak_alias = mod->mle_etak;
			mod->mle_pKLM = NULL; mod->mle_etaik = NULL; mod->mle_etak = NULL;
			reset_model_state(opt, mod);		/* srand(opt->seed) */
			if ((err = initialize_model(opt, dat, mod))) die("initialize_model (bootstrap) failed");
			dump_q("q_bsinit.f64", opt, dat, mod, 0);
			dump_p("p_bsinit.f64", dat, mod, 0);
			mod->mle_pKLM = alias; mod->mle_etaik = eta_alias; mod->mle_etak = etak_alias;
		}
		/* 6b: what run_bootstrap's fits do next: EM steps on the simulated data set from that initialisation.  With
		 * --projection (do_projection = 0) an allele the replicate lacks keeps p = 0 for every k. */
		if (atoi(getenv("REF_HARNESS_BOOTSTRAP")) >= 2) {
			const int n_bs = 5;
			double saved_abs = opt->abs_error;
			int absent = 0;
			for (int l = 0; l < dat->L; l++)
				for (int m = 0; m < dat->uniquealleles[l]; m++) {
					int tot = 0;
					for (int i = 0; i < dat->I; i++) tot += dat->ILM[i][l][m];
					if (!tot) absent++;
				}
			man_int("bs_absent_columns", absent);
			opt->abs_error = 0; opt->rel_error = 0;
			f = xopen("bs_em_ll.f64");
			for (int s = 1; s <= n_bs; s++) {
				em_step(opt, dat, mod);
				double ll = mod->logL;
				fwrite(&ll, sizeof ll, 1, f);
			}
			fclose(f);
			opt->abs_error = saved_abs;
			dump_q("q_bsstep.f64", opt, dat, mod, 0);
			dump_p("p_bsstep.f64", dat, mod, 0);
			dump_sik("sik_bsstep.f64", opt, dat, mod);
			man_int("bs_em_steps", n_bs);
			man_dbl("bs_ll_after_em", log_likelihood(opt, dat, mod, 0));
		}
		cleanup_parametric_bootstrap(dat);
		free(mod->mle_pKLM);
		mod->mle_pKLM = NULL; mod->mle_etaik = NULL; mod->mle_etak = NULL;
		man_int("bootstrap_seed", (long)opt->seed + 7);
	}

	/* ---- section 7: Rand-EM initialisation (rnd_init.c:123-160, 412-444), unreachable from the reference's own command
	 * line (initialization_procedure is never set to RAND_EM): REF_HARNESS_RANDEM=<n candidates>.  7a repeats the calls of
	 * randem_initialize_*() with recording (per-candidate log likelihood, first candidate's parameters); 7b is the
	 * reference's own routine through initialize_model(), then em() from its result.  Both must agree. ---- */
	if (getenv("REF_HARNESS_RANDEM") && mod->K > 1) {
		const int n_cand = atoi(getenv("REF_HARNESS_RANDEM"));
		double max_ll = -INFINITY;
		int best = -1;
		opt->accel_scheme = 0;
		opt->q = 1;
		opt->n_rand_em_init = n_cand;
		reset_model_state(opt, mod);
		f = xopen("randem_ll.f64");
		for (int c = 0; c < n_cand; c++) {
			if (opt->admixture) {
				random_allele_center(dat, mod);
				initialize_parameters_admixture(opt, dat, mod);
			} else {
				if (opt->initialization_method == RANDOM_PARTITION) random_individual_partition(dat, mod);
				else random_individual_center(dat, mod);
				initialize_parameters_mixture(dat, mod);
			}
			if (c == 0) {
				dump_q("q_randem_c0.f64", opt, dat, mod, 0);
				dump_p("p_randem_c0.f64", dat, mod, 0);
			}
			double ll = em_e_step(opt, dat, mod);
			fwrite(&ll, sizeof ll, 1, f);
			if (ll > max_ll) { max_ll = ll; best = c; }
		}
		fclose(f);
		man_int("randem_candidates", n_cand);
		man_int("randem_best", best);
		man_key("rand_after_randem_trace"); fprintf(man, "%d", rand());
		opt->initialization_procedure = RAND_EM;
		reset_model_state(opt, mod);
		if ((err = initialize_model(opt, dat, mod))) die("initialize_model (Rand-EM) failed");
		man_key("rand_after_randem"); fprintf(man, "%d", rand());
		dump_q("q_randem.f64", opt, dat, mod, 0);
		dump_p("p_randem.f64", dat, mod, 0);
		{	/* the winner's parameters are those of a candidate: its log likelihood after one EM step must be the recorded maximum */
			double ll = em_e_step(opt, dat, mod);
			if (ll != max_ll) die("Rand-EM: traced candidates disagree with randem_initialize_*()");
		}
		reset_model_state(opt, mod);
		initialize_model(opt, dat, mod);
		em(opt, dat, mod);
		man_int("randem_run_n_iter", mod->n_iter);
		man_int("randem_run_converged", mod->converged);
		man_dbl("randem_run_logL", mod->logL); man_hex("randem_run_logL_hex", mod->logL);
		dump_q("q_randemrun.f64", opt, dat, mod, mod->pindex);
		dump_p("p_randemrun.f64", dat, mod, mod->pindex);
		opt->initialization_procedure = NOTHING;
	}

	fprintf(man, "\n}\n");
	fclose(man);
	return 0;
}
