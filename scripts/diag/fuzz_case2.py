import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params
ctx = mc.Context(0)
for (I, L, K, ploidy, bound) in [(300,129,52,4,1e-40),(300,129,52,2,1e-40),(300,129,52,4,1e-8),(300,129,33,4,1e-40),(300,129,40,4,1e-40),(300,129,48,4,1e-40),(300,129,32,4,1e-40),(300,129,20,4,1e-40),(64,40,52,4,1e-40)]:
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=2, seed=5, missing=0.0)
    lb = ob.lib.mco_lower_bound(bound, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=6, lower_bound=max(lb, 1e-12))
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=lb)
    ctx.set_q(0, q0); ctx.set_p(0, p0)
    print((I, L, K, ploidy, bound), "loglik", ctx.loglik(0), "e_step", ctx.e_step(0), flush=True)
