"""Time of one random initialisation (random_allele_partition + first M step, rnd_init.c:349-357,456-482) at a
given size: partition drawn on the device (mchip_mstep_from_rand_partition) against drawn on the host with the
threaded libc-compatible generator and uploaded (mchip_mstep_from_partition)."""
import argparse
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import multiclust_amd as mc
from multiclust_amd import host

ap = argparse.ArgumentParser()
ap.add_argument("--I", type=int, default=10000)
ap.add_argument("--L", type=int, default=100000)
ap.add_argument("--K", type=int, default=8)
ap.add_argument("--ploidy", type=int, default=2)
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()

rs = np.random.default_rng(1)
ua = rs.integers(2, 5, a.L).astype(np.int32)
geno = (rs.integers(0, 1 << 30, (a.I, a.L, a.ploidy)) % ua[None, :, None]).astype(np.uint8)
ctx = mc.Context(0)
ctx.set_genotypes(ua, geno)
ctx.set_model(a.K, lower_bound=1e-8)
hl = host.load()
rng = host.McRng()
hl.mc_srand(C.byref(rng), 1234567)
window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
n = a.I * a.L * a.ploidy
for rep in range(a.reps):
    t0 = time.perf_counter()
    ctx.mstep_from_rand_partition(window, 0)
    ctx.synchronize()
    t1 = time.perf_counter()
    print("device draw + first M step: %.1f ms" % ((t1 - t0) * 1e3), flush=True)
p_dev = ctx.get_p(0)
assign = np.empty(n, dtype=np.uint8)
hl.mc_test_draw_partition.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_void_p]
hl.mc_test_draw_partition.restype = None
for rep in range(2):
    hl.mc_srand(C.byref(rng), 1234567)
    t0 = time.perf_counter()
    hl.mc_test_draw_partition(assign.ctypes.data, n, a.K, C.byref(rng))
    t1 = time.perf_counter()
    ctx.mstep_from_partition(assign, 1)
    ctx.synchronize()
    t2 = time.perf_counter()
    print("host draw %.1f ms (%d threads) + upload and first M step %.1f ms" % ((t1 - t0) * 1e3, min(16, os.cpu_count()), (t2 - t1) * 1e3), flush=True)
print("identical:", bool(np.array_equal(p_dev, ctx.get_p(1))))

# ---- parametric-bootstrap replicate: generated on the device against drawn on the host (serial, as in the reference)
if os.environ.get("INITBENCH_BOOTSTRAP", "1") != "0":
    Kb = a.K
    T = int(ua.sum())
    q = rs.dirichlet(np.full(Kb, 0.3), a.I)
    p = np.empty((Kb, T))
    toff = np.concatenate([[0], np.cumsum(ua)])
    for M in (2, 3, 4):
        cols = toff[:-1][ua == M]
        d = rs.dirichlet(np.full(M, 0.5), (Kb, cols.size))
        for m in range(M):
            p[:, cols + m] = d[:, :, m]
    for rep in range(3):
        t0 = time.perf_counter()
        ctx.simulate_genotypes(a.I, a.L, a.ploidy, ua, window, Kb, q, p)
        ctx.synchronize()
        t1 = time.perf_counter()
        print("device bootstrap data set (generate + layouts + counts): %.1f ms" % ((t1 - t0) * 1e3), flush=True)
    sub = max(1, a.I // 50)
    opt = host.McOptions()
    hl.mc_make_options(C.byref(opt))
    opt.admixture = 1
    ua32 = np.ascontiguousarray(ua, dtype=np.int32)
    gsub = np.ascontiguousarray(geno[:sub])
    dat = host.McData(sub, a.L, a.ploidy, ua32.ctypes.data, gsub.ctypes.data)
    out = np.empty_like(gsub)
    qs = np.ascontiguousarray(q[:sub])
    hl.mc_srand(C.byref(rng), 1234567)
    t0 = time.perf_counter()
    hl.mc_bootstrap_genotypes(C.byref(opt), C.byref(dat), Kb, qs.ctypes.data, p.ctypes.data, C.byref(rng), out.ctypes.data)
    t1 = time.perf_counter()
    print("host bootstrap data set: %.2f s for %d of %d individuals -> %.1f s for all (one core), + upload" %
          (t1 - t0, sub, a.I, (t1 - t0) * a.I / sub), flush=True)
    print("first individuals identical:", bool(np.array_equal(ctx.get_genotypes()[:sub], out)))
