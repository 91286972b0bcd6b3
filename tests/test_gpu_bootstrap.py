"""-m gpu: parametric-bootstrap data sets generated on the device (mchip_simulate_genotypes) against (a) the data
sets the reference's own parametric_bootstrap() drew from the same parameters and seed (tests/golden/*/bs_ilm.u8) and
(b) the host generator (pinned to the same golden files by tests/test_bootstrap_cpu.py) at sizes that span several
blocks of generator chunks."""
import ctypes as C

import numpy as np
import pytest

import multiclust_amd as mc
import oracle_bind as ob
from golden_util import Golden
from multiclust_amd import host
from synth import make_dataset, random_params
from test_bootstrap_cpu import counts_of, golden_bootstrap, host_options

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mc.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("name", ["multi_admix_k4", "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3"])
def test_device_bootstrap_equals_reference(ctx, name):
    g = Golden(name)
    window, _ = ob.glibc_window(g.m["bootstrap_seed"])
    ctx.simulate_genotypes(g.I, g.L, g.ploidy, g.ua, window, g.K, g.q("bs"), g.p("bs"),
                           eta_constrained=g.m["eta_constrained"])
    sim = ctx.get_genotypes()
    assert np.array_equal(counts_of(sim, g.ua), golden_bootstrap(g))


@pytest.mark.parametrize("I,L,ploidy,K,constrained,skip", [
    (300, 1000, 2, 8, 0, 0),         # 1.2e6 draws: two blocks of 256 chunks
    (257, 613, 4, 5, 0, 999),        # ragged sizes, stream already advanced, tetraploid
    (128, 2100, 2, 3, 1, 17),        # shared mixing proportions (-c)
    (40, 90, 3, 2, 0, 3),            # less than one chunk
])
def test_device_bootstrap_equals_host_generator(ctx, I, L, ploidy, K, constrained, skip):
    ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=4, seed=I + L, missing=0.02)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    q, p = random_params(I, ua, K, seed=5, lower_bound=lb)
    q[::3, 0] = lb                                    # entries at the lower bound, as fitted models have
    q /= q.sum(axis=1, keepdims=True)
    if constrained:
        q = np.ascontiguousarray(q[0])
    lib = host.load()
    opt = host.McOptions()
    lib.mc_make_options(C.byref(opt))
    opt.admixture, opt.eta_constrained = 1, constrained
    ua32 = np.ascontiguousarray(ua, dtype=np.int32)
    geno = np.ascontiguousarray(geno)
    dat = host.McData(I, L, ploidy, ua32.ctypes.data, geno.ctypes.data)
    rng = host.McRng()
    lib.mc_srand(C.byref(rng), 4242)
    for _ in range(skip):
        lib.mc_rand(C.byref(rng))
    gen = host.McSimulation()
    rng_dev = host.McRng.from_buffer_copy(rng)
    lib.mc_simulation_begin(C.byref(gen), C.byref(opt), C.byref(dat), K, q.ctypes.data, p.ctypes.data, C.byref(rng_dev))
    ref = np.empty((I, L, ploidy), dtype=np.uint8)
    lib.mc_bootstrap_genotypes(C.byref(opt), C.byref(dat), K, q.ctypes.data, p.ctypes.data, C.byref(rng), ref.ctypes.data)
    assert lib.mc_rand(C.byref(rng)) == lib.mc_rand(C.byref(rng_dev))     # both streams stand behind the data set
    ctx.simulate_genotypes(I, L, ploidy, ua, np.array(gen.window, dtype=np.uint32), K, q, p, eta_constrained=constrained)
    assert np.array_equal(ctx.get_genotypes(), ref)
    # the generated data set is a working data set: same log likelihood as the same bytes uploaded
    ctx.set_model(K, eta_constrained=constrained, lower_bound=lb)
    ctx.set_q(0, q)
    ctx.set_p(0, p)
    ll_generated = ctx.loglik(0)
    ctx.set_genotypes(ua, ref)
    ctx.set_model(K, eta_constrained=constrained, lower_bound=lb)
    ctx.set_q(0, q)
    ctx.set_p(0, p)
    assert ctx.loglik(0) == ll_generated


def test_get_genotypes_roundtrip(ctx):
    ua, geno = make_dataset(77, 131, 3, ploidy=2, max_alleles=5, seed=1, missing=0.05)
    ctx.set_genotypes(ua, geno)
    assert np.array_equal(ctx.get_genotypes(), geno)


def geno_from_counts(counts, ua, ploidy):
    """genotype bytes [I][L][ploidy] with the given allele counts [I][T] (the order of copies inside a locus is free)"""
    I, L = counts.shape[0], len(ua)
    toff = np.concatenate([[0], np.cumsum(ua)])
    out = np.full((I, L, ploidy), 255, dtype=np.uint8)
    for l in range(L):
        c = counts[:, toff[l]:toff[l + 1]]
        fill = np.zeros(I, dtype=np.int64)
        for m in range(c.shape[1]):
            for rep in range(int(c[:, m].max(initial=0))):
                rows = np.nonzero(c[:, m] > rep)[0]
                out[rows, l, fill[rows]] = m
                fill[rows] += 1
    return out


@pytest.mark.parametrize("name", ["multi_admix_k4", "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3"])
def test_bootstrap_fit_initialises_like_the_reference(ctx, name):
    """While bootstrapping, the reference's random_allele_partition reads the observed haplotypes (rnd_init.c:471), so
    the initial parameters of a fit to the simulated data equal those of a fit to the observed data for the same seed
    (q_bsinit/p_bsinit, dumped by the harness with the bootstrap data set in place)."""
    g = Golden(name)
    window, _ = ob.glibc_window(g.m["bootstrap_seed"])
    ctx.simulate_genotypes(g.I, g.L, g.ploidy, g.ua, window, g.K, g.q("bs"), g.p("bs"),
                           eta_constrained=g.m["eta_constrained"])
    ctx.set_init_genotypes(g.geno)
    ctx.set_model(g.K, eta_constrained=g.m["eta_constrained"], lower_bound=g.lower_bound)
    w0, _ = ob.glibc_window(g.m["seed"])
    ctx.mstep_from_rand_partition(w0, 0)
    np.testing.assert_allclose(ctx.get_q(0), g.q("bsinit"), rtol=1e-15, atol=1e-18)
    np.testing.assert_allclose(ctx.get_p(0), g.p("bsinit"), rtol=1e-15, atol=1e-18)
    # without the observed haplotypes the partition counts come from the simulated alleles: different parameters
    ctx.set_init_genotypes(None)
    ctx.mstep_from_rand_partition(w0, 1)
    assert not np.array_equal(ctx.get_p(1), ctx.get_p(0))


def test_bootstrap_mixture_fit_initialises_like_the_reference():
    """The mixture model's initialisation reads the allele counts, i.e. the simulated data (rnd_init.c:192-339)."""
    g = Golden("multi_mix_k3")
    sim = geno_from_counts(golden_bootstrap(g), g.ua, g.ploidy)
    fit = host.Fit(g.ua, sim, g.K, admixture=0, verbosity=1)
    fit.initialize(g.m["seed"])
    np.testing.assert_allclose(fit.get_q(0).ravel(), g.q("bsinit").ravel(), rtol=1e-15, atol=1e-18)
    np.testing.assert_allclose(fit.get_p(0), g.p("bsinit"), rtol=1e-15, atol=1e-18)
    fit.close()


@pytest.mark.parametrize("name", ["multi_admix_k4", "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3"])
def test_tiled_and_chunked_generators_agree_on_the_reference_fixtures(ctx, name, monkeypatch):
    """Both forms of both device generators on the reference's own bootstrap fixtures: the tiled ones (thread = individual x
    locus tile, integer thresholds in LDS / counts taken as the partition is drawn) and the chunked ones (thread = 3 968
    consecutive draws, stream-order intermediate + layout kernel) return the reference's data set and initial parameters."""
    g = Golden(name)
    window, _ = ob.glibc_window(g.m["bootstrap_seed"])
    w0, _ = ob.glibc_window(g.m["seed"])
    out = []
    for knobs in ({}, {"MCHIP_SIM_NO_TILE": "1", "MCHIP_PART_NO_TILE": "1"}):
        for k in ("MCHIP_SIM_NO_TILE", "MCHIP_PART_NO_TILE"):
            monkeypatch.delenv(k, raising=False)
        for k, v in knobs.items():
            monkeypatch.setenv(k, v)
        ctx.simulate_genotypes(g.I, g.L, g.ploidy, g.ua, window, g.K, g.q("bs"), g.p("bs"), eta_constrained=g.m["eta_constrained"])
        sim = ctx.get_genotypes()
        assert np.array_equal(counts_of(sim, g.ua), golden_bootstrap(g))
        ctx.set_init_genotypes(g.geno)
        ctx.set_model(g.K, eta_constrained=g.m["eta_constrained"], lower_bound=g.lower_bound)
        ctx.mstep_from_rand_partition(w0, 0)
        np.testing.assert_allclose(ctx.get_q(0), g.q("bsinit"), rtol=1e-15, atol=1e-18)
        np.testing.assert_allclose(ctx.get_p(0), g.p("bsinit"), rtol=1e-15, atol=1e-18)
        ll = ctx.em_step(0, 1)
        out.append((sim, ctx.get_q(0), ctx.get_p(0), ll, ctx.get_p(1)))
        ctx.set_init_genotypes(None)
    for a, b in zip(out[0], out[1]):
        assert np.array_equal(a, b)
