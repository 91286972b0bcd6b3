"""Indexing check beyond 2^32 allele copies (I*L*ploidy = 4.8e9): identities that hold at any size.
  * upload -> device layouts -> download gives the genotype back
  * one EM step on the full data set: sum of expected counts = number of observed copies; parameters stay normalised
  * logL(full) = logL(first half of the individuals) + logL(second half) for the same P (separate uploads)
  * the device-drawn partition at this size: its per-individual counts for the LAST individuals equal those from the
    host stream jumped to their first draw (the jump polynomials of the high chunks are exercised); the same for a
    device-generated bootstrap data set (9.6e9 draws)"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import multiclust_amd as mc
from multiclust_amd import host
from synth import random_params

I, L, P, K = 24000, 100000, 2, 4
rs = np.random.default_rng(3)
ua = rs.integers(2, 5, L).astype(np.int32)
t0 = time.time()
geno = np.empty((I, L, P), dtype=np.uint8)
for i0 in range(0, I, 1000):
    blk = rs.integers(0, 12, (min(1000, I - i0), L, P), dtype=np.uint8)
    geno[i0:i0 + blk.shape[0]] = blk % ua[None, :, None].astype(np.uint8)
print("generated %.1f GB in %.0f s" % (geno.nbytes / 1e9, time.time() - t0), flush=True)
q0, p0 = random_params(I, ua, K, seed=2)
ctx = mc.Context(0)
ctx.set_genotypes(ua, geno)
back = ctx.get_genotypes()
assert np.array_equal(back, geno), "genotype round trip through the device layouts"
del back
print("genotype round trip ok", flush=True)
ctx.set_model(K, lower_bound=1e-8)
ctx.set_q(0, q0); ctx.set_p(0, p0)
ll_full = ctx.loglik(0)
ll_step = ctx.em_step(0, 1)
assert ll_step == ll_full or abs(ll_step - ll_full) <= 1e-12 * abs(ll_full), (ll_step, ll_full)
sik = ctx.expected_counts()
print("logL", ll_full, "sum sik", sik.sum(), "copies", I * L * P, flush=True)
assert abs(sik.sum() - I * L * P) <= 1e-9 * I * L * P
q1, p1 = ctx.get_q(1), ctx.get_p(1)
assert np.abs(q1.sum(axis=1) - 1).max() < 1e-12
toff = np.concatenate([[0], np.cumsum(ua)])
assert np.abs(np.add.reduceat(p1, toff[:-1], axis=1) - 1).max() < 1e-12
# device draw: last 40 individuals against the host stream
hl = host.load()
rng = host.McRng()
hl.mc_srand(C.byref(rng), 99)
window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
ctx.mstep_from_rand_partition(window, 2)
cnt = ctx.expected_counts()          # hard-partition counts per individual and cluster
tail = 40
hl.mc_rng_jump(C.byref(rng), (I - tail) * L * P)
draws = np.fromiter((hl.mc_rand(C.byref(rng)) % K for _ in range(tail * L * P)), dtype=np.int64, count=tail * L * P).reshape(tail, L * P)
# d_iklm = 1 (not += 1) per matching copy: a homozygote whose two copies draw the same cluster counts once
g_tail = geno[I - tail:].reshape(tail, L, P)
d_tail = draws.reshape(tail, L, P)
exp = np.zeros((tail, K))
for k in range(K):
    hit = d_tail == k
    both_same = hit[:, :, 0] & hit[:, :, 1] & (g_tail[:, :, 0] == g_tail[:, :, 1])
    exp[:, k] = hit.sum(axis=(1, 2)) - both_same.sum(axis=1)
assert np.array_equal(cnt[I - tail:], exp), (cnt[I - tail:][:2], exp[:2])
print("device-drawn partition: last %d individuals match the host stream at draw offset %.3e" % (tail, (I - tail) * L * P), flush=True)
# device-generated bootstrap data set at this size (9.6e9 draws): the last individuals against the host generator
hl.mc_srand(C.byref(rng), 77)
window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
ctx.simulate_genotypes(I, L, P, ua, window, K, q0, p0)
sim_tail = ctx.get_genotypes()[I - tail:]
hl.mc_rng_jump(C.byref(rng), 2 * (I - tail) * L * P)
opt = host.McOptions()
hl.mc_make_options(C.byref(opt))
opt.admixture = 1
ua32 = np.ascontiguousarray(ua, dtype=np.int32)
g_dummy = np.ascontiguousarray(geno[I - tail:])
dat = host.McData(tail, L, P, ua32.ctypes.data, g_dummy.ctypes.data, None)
q_tail = np.ascontiguousarray(q0[I - tail:])
ref = np.empty((tail, L, P), dtype=np.uint8)
hl.mc_bootstrap_genotypes(C.byref(opt), C.byref(dat), K, q_tail.ctypes.data, p0.ctypes.data, C.byref(rng), ref.ctypes.data)
assert np.array_equal(sim_tail, ref)
print("device-generated bootstrap data set: last %d individuals match the host generator at draw offset %.3e" % (tail, 2 * (I - tail) * L * P), flush=True)
ctx.set_genotypes(ua, geno)
del ctx
# halves
lls = []
for sl in (slice(0, I // 2), slice(I // 2, I)):
    c = mc.Context(0)
    c.set_genotypes(ua, geno[sl])
    c.set_model(K, lower_bound=1e-8)
    c.set_q(0, q0[sl]); c.set_p(0, p0)
    lls.append(c.loglik(0))
    c.close()
print("halves", lls, "sum", sum(lls), "full", ll_full, flush=True)
assert abs(sum(lls) - ll_full) <= 1e-11 * abs(ll_full)
print("bigcheck ok")
