import sys, os, numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_bind as ob
from synth import make_dataset, random_params
I,L,K,ploidy = 70, 37, 3, 2
ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=4, seed=3, missing=0.02)
ua = ua.copy(); geno = geno.copy()
for l in (0, 7, 15, 36):          # first locus, last of a block of 8, last locus
    ua[l] = 0; geno[:, l, :] = 0xFF
lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
q0, p0 = random_params(I, ua, K, seed=4, lower_bound=lb)
print("T", int(ua.sum()), p0.shape)
opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0)
mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
mod.q(0)[...] = q0; mod.p(0)[...] = p0
for s in range(3): mod.em_step()
print("oracle", mod.logL, mod.loglik(0))
if len(sys.argv) > 1:
    import multiclust_amd as mc
    ctx = mc.Context(0); ctx.set_genotypes(ua, geno); ctx.set_model(K, lower_bound=lb); ctx.set_q(0, q0); ctx.set_p(0, p0)
    for s in range(3): ll = ctx.em_step(0, 0)
    print("gpu   ", ll, ctx.loglik(0), np.abs(ctx.get_q(0) - mod.q(0)).max(), np.abs(ctx.get_p(0) - mod.p(0)).max())
