"""Reproducer of the K = 52 wrong result (round 2, commit 79fba51): the tetraploid reciprocal-per-copy S-side pass
k_individual_sparse<4, true, true, false> with ONE lane per individual (the shipped build splits K > 27 over lanes, so the
instance only exists in `make exp-k52` builds).  Run with MCHIP_LIB_PATH pointing at scripts/exp/libmulticlust_hip_k52agpr.so
(hipcc's default spilling) or ..._k52scratch.so (KFLAGS).  Prints, per case, the log likelihood of the stand-alone pass, of the
E step and of the oracle, and which individuals' expected counts are wrong."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params

CASES = [  # I, L, K, ploidy, max alleles, missing, projection, bound, seed
    (300, 129, 52, 4, 2, 0.0, 1, 1e-40, 1013700932),      # the fuzz case that failed
    (64, 40, 52, 4, 3, 0.02, 1, 1e-40, 5),
    (300, 129, 52, 4, 2, 0.0, 1, 1e-8, 5),                # shared reciprocals (SAFE = false): was never wrong
    (300, 129, 52, 2, 2, 0.0, 1, 1e-40, 5),
]
print("library:", mc.lib_path())
for (I, L, K, ploidy, maxal, missing, projection, bound, seed) in CASES:
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
    lb = ob.lib.mco_lower_bound(bound, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=max(lb, 1e-12))
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0, do_projection=projection)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    want = mod.loglik(0)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, do_projection=projection, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    ll, es = ctx.loglik(0), ctx.e_step(0)
    sik = ctx.expected_counts()
    line = "I=%d L=%d K=%d ploidy=%d bound=%g: loglik pass %.9f  E step %.9f  oracle %.9f" % (I, L, K, ploidy, bound, ll, es, want)
    bad_rows = np.where(~np.isfinite(sik).all(axis=1) | (np.abs(sik.sum(axis=1) - (geno[:, :, :] != 0xFF).sum(axis=(1, 2))) > 1e-6))[0]
    line += "  | individuals whose expected counts do not add up to their copies: %d" % len(bad_rows)
    if len(bad_rows):
        line += " %s..." % bad_rows[:16].tolist()
        i = bad_rows[0]
        line += "\n   row %d: %s" % (i, np.array2string(sik[i][:12], precision=4))
    print(line, flush=True)
    ctx.close()
