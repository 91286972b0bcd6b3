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
