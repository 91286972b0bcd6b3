import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params
ctx = mc.Context(0)
bad = []
for ploidy in (2, 4, 3):
    for bound, proj in ((1e-40, 1), (1e-8, 0), (1e-8, 1)):
        for K in range(1, 65):
            I, L = 64, 40
            ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=3, seed=5, missing=0.02)
            lb = ob.lib.mco_lower_bound(bound, I, ploidy)
            q0, p0 = random_params(I, ua, K, seed=6, lower_bound=max(lb, 1e-12))
            ctx.set_genotypes(ua, geno)
            ctx.set_model(K, lower_bound=lb, do_projection=proj)
            ctx.set_q(0, q0); ctx.set_p(0, p0)
            a, b = ctx.loglik(0), ctx.e_step(0)
            if not (a == b):
                bad.append((ploidy, bound, proj, K, a, b))
print("mismatches:", bad)
