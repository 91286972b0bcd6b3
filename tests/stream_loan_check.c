/* CPU check of ref_bind.c's loan of the libc rand() stream (build container only: needs the reference's multiclust.h):
 * srand(seed); a draws by rand(); take; b draws by mc_rand on the borrowed state; give; c draws by rand()
 * must equal a + b + c draws by rand() alone.  Prints "ok" or the first mismatch. */
#include "../oracle/glue/ref_bind.c"

int main(int argc, char **argv)
{
	const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1234567u;
	enum { A = 1000, Bn = 777, Cn = 1000, ROUNDS = 5 };
	static int want[ROUNDS * (A + Bn + Cn)];
	srand(seed);
	for (int i = 0; i < ROUNDS * (A + Bn + Cn); i++) want[i] = rand();
	srand(seed);
	int n = 0;
	for (int r = 0; r < ROUNDS; r++) {
		for (int i = 0; i < A; i++, n++) if (rand() != want[n]) { printf("mismatch at %d (libc, round %d)\n", n, r); return 1; }
		mc_rng g;
		stream_take(&g);
		for (int i = 0; i < Bn; i++, n++) if (mc_rand(&g) != want[n]) { printf("mismatch at %d (borrowed, round %d)\n", n, r); return 1; }
		stream_give(&g);
		for (int i = 0; i < Cn; i++, n++) if (rand() != want[n]) { printf("mismatch at %d (after give, round %d)\n", n, r); return 1; }
	}
	printf("ok\n");
	return 0;
}
