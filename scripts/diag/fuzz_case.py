import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params
I, L, K, ploidy, maxal, missing, projection, bound, seed = 300, 129, 52, 4, 2, 0.0, 1, 1e-40, 1013700932
ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
lb = ob.lib.mco_lower_bound(bound, I, ploidy)
q0, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=max(lb, 1e-12))
print("lb", lb, "min q", q0.min(), "min p", p0.min())
opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0, do_projection=projection)
mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
mod.q(0)[...] = q0; mod.p(0)[...] = p0
print("oracle ll", mod.loglik(0))
ctx = mc.Context(0)
ctx.set_genotypes(ua, geno)
ctx.set_model(K, do_projection=projection, lower_bound=lb)
ctx.set_q(0, q0); ctx.set_p(0, p0)
print("gpu loglik", ctx.loglik(0), "e_step", ctx.e_step(0))
# which individuals? split
for lo, hi in ((0, 150), (150, 300), (0, 64), (64, 128)):
    c = mc.Context(0); c.set_genotypes(ua, geno[lo:hi]); c.set_model(K, do_projection=projection, lower_bound=lb); c.set_q(0, q0[lo:hi]); c.set_p(0, p0)
    print(lo, hi, c.loglik(0)); c.close()
