import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset
I, L, ploidy, K, skip = 200, 300, 2, 47, 9
ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=4, seed=I + K, missing=0.01)
lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
ctx = mc.Context(0)
ctx.set_genotypes(ua, geno)
ctx.set_model(K, lower_bound=lb)
window, rng = ob.glibc_window(20250117, skip)
assign = ob.rand_mod(rng, I * L * ploidy, K)
ctx.mstep_from_partition(assign, 0)
ctx.mstep_from_rand_partition(window, 1)
ctx.mstep_from_partition(assign, 2)
p0, p1, p2 = ctx.get_p(0), ctx.get_p(1), ctx.get_p(2)
opt = ob.make_options(lower_bound=lb, fused=1)
mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
mod.init_from_partition(assign)
po = mod.p(0)
for name, a, b in (("p0 vs p1", p0, p1), ("p0 vs p2", p0, p2), ("p0 vs oracle", p0, po), ("p1 vs oracle", p1, po)):
    d = np.argwhere(a != b)
    print(name, len(d), d[:6].tolist(), [(a[tuple(x)], b[tuple(x)]) for x in d[:4]])
