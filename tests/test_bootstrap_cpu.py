"""CPU test of the host-side parametric bootstrap generator (multiclust_amd/host/mc_em.c: mc_bootstrap_genotypes)
against data sets the reference's own parametric_bootstrap() produced from the same parameters and rand() seed
(tests/golden/*/bs_ilm.u8, written by section 6 of oracle/ref_harness.c)."""
import ctypes as C

import numpy as np
import pytest

from golden_util import Golden
from multiclust_amd import host

CASES = ["multi_admix_k4", "tetra_admix_k3", "missing_admix_k3", "multi_mix_k3", "multi_admix_c_k3"]


def counts_of(geno, ua):
    """allele-count form [I][T] (the reference's dat->ILM) of genotype bytes [I][L][ploidy]"""
    I, L, pl = geno.shape
    toff = np.concatenate([[0], np.cumsum(ua)])[:-1]
    out = np.zeros((I, int(np.sum(ua))), dtype=np.uint8)
    ii = np.repeat(np.arange(I), L * pl)
    cols = (toff[None, :, None] + geno.astype(np.int64)).ravel()
    ok = geno.ravel() != 255
    np.add.at(out, (ii[ok], cols[ok]), 1)
    return out


def golden_bootstrap(g):
    return np.fromfile(g.dir + "/bs_ilm.u8", dtype=np.uint8).reshape(g.I, g.T)


def host_options(lib, g):
    opt = host.McOptions()
    lib.mc_make_options(C.byref(opt))
    opt.admixture, opt.eta_constrained = g.m["admixture"], g.m["eta_constrained"]
    return opt


@pytest.mark.parametrize("name", CASES)
def test_host_bootstrap_equals_reference(name):
    g = Golden(name)
    lib = host.load()
    opt = host_options(lib, g)
    ua = np.ascontiguousarray(g.ua, dtype=np.int32)
    dat = host.McData(g.I, g.L, g.ploidy, ua.ctypes.data, g.geno.ctypes.data)
    q, p = np.ascontiguousarray(g.q("bs")), np.ascontiguousarray(g.p("bs"))
    rng = host.McRng()
    lib.mc_srand(C.byref(rng), g.m["bootstrap_seed"])
    sim = np.empty((g.I, g.L, g.ploidy), dtype=np.uint8)
    lib.mc_bootstrap_genotypes(C.byref(opt), C.byref(dat), g.K, q.ctypes.data, p.ctypes.data, C.byref(rng), sim.ctypes.data)
    assert np.array_equal(counts_of(sim, ua), golden_bootstrap(g))
    assert lib.mc_rand(C.byref(rng)) == g.m["rand_after_bootstrap"]     # the stream stands where the reference's does
    # and the draw count the jump-ahead relies on
    rng2 = host.McRng()
    lib.mc_srand(C.byref(rng2), g.m["bootstrap_seed"])
    lib.mc_rng_jump(C.byref(rng2), lib.mc_bootstrap_draws(C.byref(opt), C.byref(dat)))
    assert lib.mc_rand(C.byref(rng2)) == g.m["rand_after_bootstrap"]


def walk_threshold(c):
    """k_walk_tables' rule (multiclust_amd/csrc/mchip.hip: walk_threshold), restated: the smallest v in [0, 2^31] with
    float64(v) / RAND_MAX > c."""
    D = 2147483647.0
    if not (c < 2.0):
        return 1 << 31
    g = c * D - 2.0
    v = int(g) if g > 0.0 else 0
    v = min(v, 1 << 31)
    while v > 0 and np.float64(v - 1) / D > c:
        v -= 1
    while v < (1 << 31) and not (np.float64(v) / D > c):
        v += 1
    return v


def test_integer_thresholds_decide_what_the_inverse_cdf_walk_decides():
    """The device bootstrap generator replaces `while (k < K && r > sum) sum += eta[k++]; if (k) k--;` with r = rand() / RAND_MAX
    (bootstrap.c:97-119) by integer compares rand() >= V(c_k) against thresholds of the left-to-right partial sums c_k.  The two
    agree for every value rand() can take: checked on both sides of each threshold and on random draws, for partial sums that
    include 0, exact fractions, values at the lower bound and sums that round above 1."""
    rs = np.random.default_rng(8)
    D = 2147483647.0
    for trial in range(300):
        K = int(rs.integers(1, 9))
        eta = rs.dirichlet(np.full(K, 0.4))
        if trial % 3 == 0:
            eta[rs.integers(0, K)] = 1e-8
        if trial % 7 == 0:
            eta = np.full(K, 1.0 / K)
        cum, s = [], 0.0
        for k in range(K):
            cum.append(s)
            s += eta[k]
        thr = [walk_threshold(c) for c in cum]
        assert thr[0] == 1 and all(a <= b for a, b in zip(thr, thr[1:]))
        probes = set(int(x) for x in rs.integers(0, 1 << 31, 40))
        for t in thr:
            probes.update(v for v in (t - 2, t - 1, t, t + 1) if 0 <= v < (1 << 31))
        probes.update((0, 1, (1 << 31) - 1))
        for v in probes:
            r = np.float64(v) / D
            k, ssum = 0, 0.0
            while k < K and r > ssum:                       # the reference's walk
                ssum += eta[k]
                k += 1
            if k:
                k -= 1
            mine = max(sum(1 for t in thr if v >= t) - 1, 0)
            assert mine == k, (v, eta, thr)
