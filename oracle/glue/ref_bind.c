/*
 * ref_bind.c -- the binding of INTEGRATION.md EXECUTED: the five EM-layer entry points the reference's driver calls
 * (`nm` of multiclust.o + cline.o + read_file.o + write_file.o + message.o + bootstrap.o leaves exactly these undefined:
 * initialize_model, em, converged, aic, bic), implemented on ref_glue.c's functions, i.e. on libmulticlust_host.so /
 * libmulticlust_hip.so.
 *
 * TEST INFRASTRUCTURE (oracle side).  oracle/Makefile links this file and ref_glue.c with the reference's OWN, unmodified
 * multiclust.c (its main(), option parser, maximize_likelihood, run_bootstrap), cline.c, read_file.c, write_file.c, message.c
 * and bootstrap.c where they lie under /root/reference -- and WITHOUT its em_alg.c, accel_em.c, log_likelihood.c, simplex.c
 * and rnd_init.c -- into oracle/_ref/multiclust_ref_hip: the reference program with its EM layer replaced by the MI355X
 * path, which is what a maintainer following INTEGRATION.md would build.  The binary travels to the GPU box (the sources do
 * not); tests/test_gpu_refbind.py runs it on the command lines of the reference's own goldens.  No GPU: every fit fails.
 *
 * The libc rand() stream stays the program's one random stream: the reference seeds it (srand in parse_options,
 * multiclust.c:1592-1596) and parametric_bootstrap() (bootstrap.c) keeps drawing from it; an initialisation borrows its
 * state (initstate/setstate expose glibc's TYPE_3 table: word 0 = 5 * rear + type, words 1..31 = the table), lets the device
 * draw the partition from it and hands the advanced state back.
 */
#define _DEFAULT_SOURCE		/* initstate / setstate under -std=c17 */
#include "multiclust.h"
#include "mc_host.h"
#include "ref_glue.h"

static struct {
	const data *dat;	/* data set the flat genotype was made from */
	int ***ilm;		/* its count array at that time: parametric_bootstrap() swaps dat->ILM (bootstrap.c:35-41) */
	uint8_t *flat;		/* observed genotype, allele indices */
	mc_options mo;
	mc_data md;
	uint8_t *sim;		/* the bootstrap replicate parametric_bootstrap() left in dat->ILM, as allele copies */
	uint64_t sim_hash;
	int on_sim;
	mc_model *mm;
	int K;
	double *sik;
} B;

static void bind_fatal(const char *what)
{
	fprintf(stderr, "multiclust_ref_hip: %s\n", what);
	exit(EXIT_FAILURE);
}

static void unbind(void)
{
	if (B.mm) mc_model_free(B.mm);
	B.mm = NULL;
}

/* counts -> copies in allele order (the EM path reads counts only; the random partition of the admixture model keeps reading the
 * observed copies, mc_data::init_geno, as rnd_init.c:471 keeps reading dat->IL) */
static uint64_t copies_from_counts(const data *dat, uint8_t *g)
{
	uint64_t h = 1469598103934665603ull;
	for (int i = 0; i < dat->I; i++)
		for (int l = 0; l < dat->L; l++) {
			int a = 0;
			for (int m = 0; m < dat->uniquealleles[l]; m++)
				for (int c = 0; c < dat->ILM[i][l][m] && a < dat->ploidy; c++) {
					g[((size_t)i * dat->L + l) * dat->ploidy + a++] = (uint8_t)m;
					h = (h ^ (uint64_t)(m + 1)) * 1099511628211ull;
				}
			for (; a < dat->ploidy; a++) g[((size_t)i * dat->L + l) * dat->ploidy + a] = MCHIP_MISSING;
			h = (h ^ 255u) * 1099511628211ull;
		}
	return h;
}

static void bind(options *opt, data *dat, model *mod)
{
	int fresh = 0;
	if (opt->pfile || opt->qfile) bind_fatal("starting values from files (-p/-q) are not bound");
	if (B.dat != dat) {
		mc_watchdog_from_env();		/* MC_WATCHDOG_S, as the drop-in command line: tests/procutil.py sets it */
		unbind();
		free(B.flat); free(B.sim);
		B.sim = NULL;
		B.on_sim = 0;
		B.flat = mcamd_flatten_genotypes(dat);
		if (!B.flat) bind_fatal("out of memory");
		B.dat = dat;
		B.ilm = dat->ILM;
		atexit(unbind);
		fresh = 1;
	}
	/* -b: parametric_bootstrap() (bootstrap.c, the reference's own generator, drawing from the program's rand() stream) swaps
	 * dat->ILM for its replicate array once and refills it per replicate (bootstrap.c:35-41): the replicate is uploaded like any
	 * data set (last row, second form) */
	const int on_sim = dat->ILM != B.ilm;
	const uint8_t *geno = B.flat;
	if (on_sim) {
		if (!B.sim && !(B.sim = malloc((size_t)dat->I * dat->L * dat->ploidy))) bind_fatal("out of memory");
		const uint64_t h = copies_from_counts(dat, B.sim);
		if (!B.on_sim || h != B.sim_hash) fresh = 1;	/* another replicate */
		B.sim_hash = h;
		geno = B.sim;
	} else if (B.on_sim) {
		fresh = 1;
	}
	B.on_sim = on_sim;
	/* after the reference's synchronize() (multiclust.c:807): opt->lower_bound is already MIN(bound, 0.5 / I / ploidy), and
	 * mc_synchronize applies the same rule again without changing it */
	if (mcamd_options(opt, dat, geno, opt->lower_bound, &B.mo, &B.md)) bind_fatal("options do not carry over");
	B.md.init_geno = on_sim ? B.flat : NULL;
	if (fresh || !B.mm || B.K != mod->K) {
		unbind();
		const char *dev = getenv("MCAMD_DEVICE");
		if (mcamd_allocate_model_for_k(&B.mo, &B.md, mod, dev ? atoi(dev) : 0, &B.mm))
			bind_fatal("no device context (there is no CPU fallback)");
		B.K = mod->K;
		free(B.sik);
		if (!(B.sik = malloc(sizeof(double) * (size_t)dat->I * mod->K))) bind_fatal("out of memory");
	}
}

/* parse_options() tests errno after every numeric argument without ever clearing it (multiclust.c:1538 and the like): whatever
 * the constructors of the GPU runtime's libraries left there must be gone before main() runs */
__attribute__((constructor)) static void clear_errno_before_main(void) { errno = 0; }

/* ---- the libc stream on loan ---- */
static int32_t *libc_words;

static void stream_take(mc_rng *g)
{
	static char parked[128];
	libc_words = (int32_t *)initstate(1, parked, sizeof parked);	/* switching away makes glibc record the rear index */
	if (!libc_words || libc_words[0] % 5 != 3) bind_fatal("libc's rand() is not glibc's TYPE_3 generator");
	memcpy(g->r, libc_words + 1, sizeof g->r);
	g->b = libc_words[0] / 5;
	g->f = (g->b + 3) % 31;
}

static void stream_give(const mc_rng *g)
{
	memcpy(libc_words + 1, g->r, sizeof g->r);
	libc_words[0] = 5 * g->b + 3;
	setstate((char *)libc_words);
}

/* ---- the five entry points (prototypes: multiclust.h:371-388) ---- */
int initialize_model(options *opt, data *dat, model *mod)		/* rnd_init.c:54 */
{
	mc_rng rng;
	const int errno_before = errno;
	/* the stream is taken BEFORE anything touches the GPU runtime: creating the first HIP context draws from (or reseeds) libc's
	 * generator -- found when this program first ran on a GPU box: same seed, different initialisations from run to run.
	 * While the state is on loan libc runs on a parked table, so such draws cannot disturb the program's stream */
	stream_take(&rng);
	bind(opt, dat, mod);
	const int err = mcamd_initialize_model(&B.mo, &B.md, mod, B.mm, &rng);
	stream_give(&rng);
	errno = errno_before;
	return err;
}

void em(options *opt, data *dat, model *mod)				/* em_alg.c:44 */
{
	mc_rng held;
	/* the reference reads errno as "did my last allocation fail" without clearing it first (write_file.c:417,424,...: the popq /
	 * indivq writers return silently when it is non-zero): what the GPU runtime's system calls leave there must not reach it */
	const int errno_before = errno;
	if (!B.mm || B.dat != dat || B.K != mod->K) bind_fatal("em() without initialize_model()");
	stream_take(&held);						/* parked for the duration of the GPU work, handed back as it was */
	mcamd_em(&B.mo, &B.md, mod, B.mm);
	if (mcamd_fetch_parameters(opt, dat, mod, B.mm) || mcamd_fetch_expected_counts(opt, dat, mod, B.mm, B.sik))
		bind_fatal("cannot read the fit back");
	stream_give(&held);
	errno = errno_before;
	if (opt->admixture) {
		/* the writers only ever sum diklm over (l, m) (write_file.c:366,455,540): the sums go into the first cell */
		int l0 = 0;
		while (l0 < dat->L && dat->uniquealleles[l0] < 1) l0++;
		if (l0 == dat->L) bind_fatal("no locus with an allele");
		for (int i = 0; i < dat->I; i++)
			for (int k = 0; k < mod->K; k++) {
				for (int l = 0; l < dat->L; l++)
					for (int m = 0; m < dat->uniquealleles[l]; m++) mod->diklm[i][k][l][m] = 0.0;
				mod->diklm[i][k][l0][0] = B.sik[(size_t)i * mod->K + k];
			}
	}
}

int converged(options *opt, model *mod, double loglik)		/* em_alg.c:163 */
{
	mc_options mo;
	mc_model t;
	mc_make_options(&mo);
	mo.abs_error = opt->abs_error;
	mo.rel_error = opt->rel_error;
	memset(&t, 0, sizeof t);
	t.logL = mod->logL;
	const int stop = mc_converged(&mo, &t, loglik);
	if (t.converged) mod->converged = 1;
	return stop;
}

double aic(model *mod) { return mc_aic(mod->max_logL, mod->no_parameters); }			/* log_likelihood.c:70 */
double bic(data *dat, model *mod) { return mc_bic(mod->max_logL, mod->no_parameters, dat->I); }	/* log_likelihood.c:82 */
