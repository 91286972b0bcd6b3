"""Device-memory leak check: contexts created, used along every generator / fit path and destroyed in a loop; free device
memory afterwards must be where it started.  Prints the change per stage of use."""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests"))
import multiclust_amd as mc
import oracle_bind as ob
from multiclust_amd import hip
from synth import make_dataset, random_params


def free_mb():
    torch.cuda.synchronize()
    return torch.cuda.mem_get_info(0)[0] / 2**20


ua, geno = make_dataset(700, 3000, 8, ploidy=2, max_alleles=4, seed=3, missing=0.01)
q, p = random_params(700, ua, 8, seed=4, lower_bound=1e-8)
window, _ = ob.glibc_window(99)
torch.zeros(1, device="cuda")


def life(stage):
    c = mc.Context(0)
    c.set_genotypes(ua, geno)
    if stage >= 1:
        for K in (3, 8, 20, 40):
            c.set_model(K, lower_bound=1e-8)
            c.mstep_from_rand_partition(window, 0)
            c.em_step(0, 0)
            if stage >= 2:
                st = hip.RunState(logL=-np.inf, abs_error=1e-300, n_iter=0)
                c.lib.mchip_em_run(c.h, 0, 8, C.byref(st))
            if stage >= 3:
                st = hip.RunState(logL=-np.inf, abs_error=1e-300, n_iter=0)
                c.lib.mchip_accel_run(c.h, 0, 3, 3, C.byref(st))
    if stage >= 4:
        c.simulate_genotypes(700, 3000, 2, ua, window, 8, q, p)
        c.set_model(8, lower_bound=1e-8)
        c.mstep_from_rand_partition(window, 0)
        c.em_step(0, 0)
    if stage >= 5:
        d = mc.Context(0)
        d.copy_genotypes(c)
        d.set_model(7, lower_bound=1e-8)
        d.mstep_from_rand_partition(window, 0)
        d.close()
    c.close()


worst = 0.0
for stage in range(6):
    for _ in range(3):
        life(stage)
    a = free_mb()
    for _ in range(20):
        life(stage)
    b = free_mb()
    print("stage %d: %.1f MiB per context lifetime" % (stage, (a - b) / 20), flush=True)
    worst = max(worst, (a - b) / 20)
assert worst < 0.5, "device memory leaked"
print("leakcheck ok")
