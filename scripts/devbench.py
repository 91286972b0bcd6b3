"""Developer micro-benchmark (not the contract bench): times the EM passes at a given shape."""
import argparse
import sys
import time
import os

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import multiclust_amd as mc
from synth import random_params


def fast_geno(I, L, ploidy, maxal, seed):
    rng = np.random.default_rng(seed)
    ua = rng.integers(2, maxal + 1, size=L).astype(np.int32) if maxal > 2 else np.full(L, 2, np.int32)
    geno = (rng.integers(0, 1 << 30, size=(I, L, ploidy), dtype=np.int64) % ua[None, :, None]).astype(np.uint8)
    return ua, geno


ap = argparse.ArgumentParser()
ap.add_argument("--I", type=int, default=2000)
ap.add_argument("--L", type=int, default=20000)
ap.add_argument("--K", type=int, default=5)
ap.add_argument("--ploidy", type=int, default=2)
ap.add_argument("--maxal", type=int, default=2)
ap.add_argument("--steps", type=int, default=10)
a = ap.parse_args()
t0 = time.time()
ua, geno = fast_geno(a.I, a.L, a.ploidy, a.maxal, 1)
q0, p0 = random_params(a.I, ua, a.K, seed=2)
print("gen %.1fs" % (time.time() - t0), flush=True)
ctx = mc.Context(0)
print(ctx.device_info())
t0 = time.time()
ctx.set_genotypes(ua, geno)
ctx.set_model(a.K, lower_bound=1e-8)
ctx.set_q(0, q0); ctx.set_p(0, p0)
print("upload %.2fs" % (time.time() - t0), flush=True)
for _ in range(2):
    ctx.em_step(0, 0)
ctx.profile_begin()
for _ in range(a.steps):
    ctx.em_step(0, 0, sync=False)
ll = ctx.last_loglik()
total, km, kl = ctx.profile_end()
cells = a.I * int(ua.sum())
print("I=%d L=%d T=%d K=%d p=%d: %.3f ms/step (column pass %.3f ms, individual pass %.3f ms) logL=%.6f" % (
    a.I, a.L, int(ua.sum()), a.K, a.ploidy, total / a.steps, km[0] / max(kl[0], 1), km[1] / max(kl[1], 1), ll))
gb = a.I * a.L * a.ploidy / 1e9
print("  genotype %.3f GB; column pass %.1f GB/s, individual pass %.1f GB/s; %.2f ns/cell-step" % (
    gb, gb / (km[0] / kl[0] / 1e3), gb / (km[1] / kl[1] / 1e3), total / a.steps * 1e6 / cells))
ctx.profile_begin()
for _ in range(a.steps):
    ctx.loglik(0, sync=False)
ctx.synchronize()
total, km, kl = ctx.profile_end()
print("  loglik pass %.3f ms" % (km[2] / kl[2]))
