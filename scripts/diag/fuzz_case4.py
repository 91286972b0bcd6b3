import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params
ctx = mc.Context(0)
for K in (51, 52, 53):
    I, L, ploidy = 64, 40, 4
    ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=3, seed=5, missing=0.02)
    lb = ob.lib.mco_lower_bound(1e-40, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=6, lower_bound=1e-12)
    ctx.set_genotypes(ua, geno); ctx.set_model(K, lower_bound=lb); ctx.set_q(0, q0); ctx.set_p(0, p0)
    print(K, ctx.loglik(0), ctx.e_step(0))
