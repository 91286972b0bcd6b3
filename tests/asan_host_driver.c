/* CPU-only sanitizer driver for the host C code (reader, writers, rand() jump-ahead, threaded partition draw, bookkeeping).
 * Built with -fsanitize=address,undefined by tests/test_host_asan.py; the device entry points are stubbed because nothing
 * here reaches them.  (GPU AddressSanitizer is not available on the pool: sanitizers run on the CPU build only.) */
#include "mc_cli.h"
#include <string.h>
#include <stdlib.h>
/* stubs for the device calls the linked objects reference (never reached here) */
int mchip_create(mchip_context **c, int d){(void)c;(void)d;return 2;}
int mchip_destroy(mchip_context *c){(void)c;return 0;}
int mchip_init_from_allele_centers(mchip_context *c,const uint8_t*a,const uint64_t*b,const uint32_t*w,uint64_t n,int t){(void)c;(void)a;(void)b;(void)w;(void)n;(void)t;return 2;}
int mchip_copy_slot(mchip_context *c,int a,int b){(void)c;(void)a;(void)b;return 2;}
int mchip_copy_genotypes(mchip_context *c,const mchip_context *s){(void)c;(void)s;return 2;}
int mchip_empty_individuals(const mchip_context *c,int *f){(void)c;if(f)*f=-1;return 0;}
const char *mchip_last_error(const mchip_context *c){(void)c;return "";}
int mchip_set_genotypes(mchip_context *c,int a,int b,int d,const int32_t*u,const uint8_t*g){(void)c;(void)a;(void)b;(void)d;(void)u;(void)g;return 2;}
int mchip_set_model(mchip_context *c,int a,int b,int d,int e,double f,double g,int h){(void)c;(void)a;(void)b;(void)d;(void)e;(void)f;(void)g;(void)h;return 2;}
int mchip_set_p(mchip_context*c,int s,const double*p){(void)c;(void)s;(void)p;return 2;}
int mchip_get_p(mchip_context*c,int s,double*p){(void)c;(void)s;(void)p;return 2;}
int mchip_set_q(mchip_context*c,int s,const double*p){(void)c;(void)s;(void)p;return 2;}
int mchip_get_q(mchip_context*c,int s,double*p){(void)c;(void)s;(void)p;return 2;}
int mchip_get_expected_counts(mchip_context*c,double*p){(void)c;(void)p;return 2;}
int mchip_mstep_from_partition(mchip_context*c,const uint8_t*a,int t){(void)c;(void)a;(void)t;return 2;}
int mchip_mstep_from_rand_partition(mchip_context*c,const uint32_t*a,int t){(void)c;(void)a;(void)t;return 2;}
int mchip_set_init_genotypes(mchip_context*c,const uint8_t*g){(void)c;(void)g;return 2;}
int mchip_get_genotypes(mchip_context*c,uint8_t*g){(void)c;(void)g;return 2;}
int mchip_simulate_genotypes(mchip_context*c,int I,int L,int p,const int32_t*u,const uint32_t*w,int K,int e,const double*q,const double*pp){(void)c;(void)I;(void)L;(void)p;(void)u;(void)w;(void)K;(void)e;(void)q;(void)pp;return 2;}
int mchip_em_step(mchip_context*c,int a,int b,double*l){(void)c;(void)a;(void)b;(void)l;return 2;}
int mchip_em_run(mchip_context*c,int a,int b,mchip_run_state*l){(void)c;(void)a;(void)b;(void)l;return 2;}
int mchip_accel_run(mchip_context*c,int a,int s,int b,mchip_run_state*l){(void)c;(void)a;(void)s;(void)b;(void)l;return 2;}
int mchip_e_step(mchip_context*c,int a,double*l){(void)c;(void)a;(void)l;return 2;}
int mchip_loglik(mchip_context*c,int a,double*l){(void)c;(void)a;(void)l;return 2;}
int mchip_loglik_prefetch(mchip_context*c,int a,double*l){(void)c;(void)a;(void)l;return 2;}
int mchip_secant(mchip_context*c,int a,int b,int d,int e){(void)c;(void)a;(void)b;(void)d;(void)e;return 2;}
int mchip_step_dots(mchip_context*c,int a,double*l){(void)c;(void)a;(void)l;return 2;}
int mchip_secant_dots(mchip_context*c,int a,int b,double*l){(void)c;(void)a;(void)b;(void)l;return 2;}
int mchip_accel_update(mchip_context*c,int a,int b,int d,double s,int q){(void)c;(void)a;(void)b;(void)d;(void)s;(void)q;return 2;}
int mchip_multisecant_update(mchip_context*c,int a,int b,int d,int n,const int*v,const double*x,const double*y){(void)c;(void)a;(void)b;(void)d;(void)n;(void)v;(void)x;(void)y;return 2;}
static unsigned long long stub_events;
int mchip_progress_note(const char *w){(void)w;stub_events++;return 0;}
int mchip_progress_report(char *b,int n,unsigned long long *e){if(e)*e=stub_events;if(b&&n>0)snprintf(b,(size_t)n,"thread 0: stub\n");return 0;}
void mc_test_draw_partition(uint8_t *assign, size_t n, int K, mc_rng *rng);
int main(int argc, char **argv)
{
	for (int a = 1; a + 1 < argc; a += 2) {
		mc_cli_options o; memset(&o, 0, sizeof o);
		o.filename = argv[a]; o.ploidy = atoi(argv[a + 1]); o.missing_value = -9;
		mc_cli_data d;
		int rc = mc_read_structure(&o, &d);
		printf("%s: rc=%d I=%d L=%d T=%d\n", argv[a], rc, d.I, d.L, d.T);
		if (!rc) {
			/* writers + partition on fake fit */
			int K = 3; double *q = calloc((size_t)d.I*K,8), *p = calloc((size_t)K*d.T,8), *s = calloc((size_t)d.I*K,8); int cnt[3];
			for (int i = 0; i < d.I*K; i++) { q[i] = 1.0/K; s[i] = i % 7; }
			mc_fit_view fv = {K, 1, -1.0, 2.0, 3.0, q, p, s};
			o.path = "/tmp/asan"; o.filename_file = "x.stru"; o.em.admixture = 1;
			mc_partition(&d, &fv, NULL, cnt);
			mc_write_results(&o, &d, &fv, cnt);
			{	/* -A: a partition file of I labels, the adjusted Rand index against the MAP partition, the failure paths */
				int *I_K = calloc((size_t)d.I, sizeof *I_K), *lab = NULL, pK = 0;
				FILE *af = fopen("/tmp/asan/partition.txt", "w");
				for (int i = 0; i < d.I; i++) fprintf(af, "%d\n", 1 + i % 4);
				fclose(af);
				mc_partition(&d, &fv, I_K, cnt);
				if (mc_read_afile("/tmp/asan/partition.txt", d.I, &lab, &pK) || pK != 4) { printf("afile failed\n"); return 1; }
				const double ar = mc_adjusted_rand(d.I, pK, K, lab, I_K), self = mc_adjusted_rand(d.I, pK, pK, lab, lab);
				printf("adjusted rand %f, of a partition with itself %f\n", ar, self);
				if (!(self > 0.999999 && self < 1.000001)) return 1;
				free(lab);
				if (mc_read_afile("/tmp/asan/partition.txt", d.I + 1, &lab, &pK) != MC_EXIT_FILE_FORMAT_ERROR || lab) return 1;
				if (mc_read_afile("/tmp/asan/absent.txt", d.I, &lab, &pK) != MC_EXIT_FILE_OPEN_ERROR || lab) return 1;
				free(I_K);
				mc_watchdog_report(stdout, 0.0);	/* the report a stalled run would leave */
			}
			free(q); free(p); free(s);
			/* the host-side walks of Rand-EM (allele-count table, center draws, unmatched-copy counts) and of the mixture
			 * model's random centers: skipping initialisations touches no device */
			for (int admix = 0; admix < 2; admix++)
				for (int KK = 1; KK <= 5; KK += 2) {
					mc_options ro; mc_make_options(&ro);
					ro.admixture = admix; ro.initialization_procedure = MC_RAND_EM; ro.n_rand_em_init = 3;
					mc_data md = { d.I, d.L, d.ploidy, d.uniquealleles, d.geno, NULL };
					mc_model mm; memset(&mm, 0, sizeof mm); mm.K = KK;
					mc_rng rg; mc_srand(&rg, 11);
					if (mc_skip_initializations(&ro, &md, &mm, &rg, 2)) return 3;
					/* the starts of the units of a sharded run, walked once, are where skipping u units from the base ends
					 * (also for the random allele partition: a jump per unit); the skips of consecutive loci taken in one
					 * step leave the generator where the per-locus skips did (same draws afterwards) */
					for (int proc = 0; proc < 2; proc++) {
						ro.initialization_procedure = proc ? MC_RAND_EM : MC_INIT_NOTHING;
						mc_rng base, starts[5];
						mc_srand(&base, 77);
						if (mc_unit_starts(&ro, &md, &mm, &base, 4, starts)) return 4;
						for (int u = 0; u <= 4; u++) {
							mc_rng w = base;
							if (mc_skip_initializations(&ro, &md, &mm, &w, u)) return 5;
							mc_rng a = starts[u];
							for (int x = 0; x < 40; x++)
								if (mc_rand(&a) != mc_rand(&w)) { printf("unit start %d differs (admix %d K %d proc %d)\n", u, admix, KK, proc); return 6; }
						}
					}
					mc_init_cache_free(&mm);
				}
			mc_free_data(&d);
		}
	}
	mc_rng g; mc_srand(&g, 5); mc_rng_jump(&g, 123456789ull);
	uint8_t *as = malloc((1u<<24)+5); setenv("MC_INIT_THREADS","3",1);
	mc_test_draw_partition(as, (1u<<24)+5, 5, &g); free(as);
	mc_summary s; mc_summary_reset(&s); mc_options eo; mc_make_options(&eo);
	mc_unit_result r = {0, -10.0, 1, 5, 0, 0, 0, 0, 0.0}; mc_summary_add(&eo, &s, &r, 10, 20);
	printf("ok %d\n", s.n_init);
	return 0;
}
