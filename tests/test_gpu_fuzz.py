"""-m gpu: seeded random shapes through the C-ABI against the CPU oracle (fused order): sizes around the tile and block
edges (8, 64, 128, 256), every ploidy 1..6, K 1..12, 2..7 alleles, with and without missing data, both eta forms, plain EM
and one SQUAREM-3 cycle's log likelihoods."""
import os

import numpy as np
import pytest

import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params

pytestmark = pytest.mark.gpu
# MC_FUZZ_SEED / MC_FUZZ_SCALE: other seeds and more cases for occasional soak runs (defaults are what the suite runs)
SEED_SHIFT = int(os.environ.get("MC_FUZZ_SEED", "0"))
SCALE = int(os.environ.get("MC_FUZZ_SCALE", "1"))


def cases(n, seed):
    rs = np.random.default_rng(seed)
    edges = [1, 2, 7, 8, 9, 15, 16, 17, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 513, 1025]
    out = []
    for _ in range(n):
        I = int(rs.choice(edges[3:]))
        L = int(rs.choice(edges[3:]))
        out.append((I, L, int(rs.integers(1, 13)), int(rs.integers(1, 7)), int(rs.integers(2, 8)),
                    float(rs.choice([0.0, 0.0, 0.03, 0.2])), int(rs.integers(0, 2)), int(rs.integers(0, 1 << 30))))
    return out


@pytest.fixture(scope="module")
def ctx():
    c = mc.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("I,L,K,ploidy,maxal,missing,constrained,seed", cases(120 * SCALE, 20250117 + SEED_SHIFT))
def test_random_shapes_vs_oracle(ctx, I, L, K, ploidy, maxal, missing, constrained, seed):
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=lb)
    if constrained:
        q0 = q0[0].copy()
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0, eta_constrained=constrained)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, eta_constrained=constrained, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    for s in (1, 2):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= max(1e-8, 1e-12 * abs(mod.logL)), (s, ll, mod.logL)
        rtol = 1e-11 if s == 1 else 1e-9
        np.testing.assert_allclose(ctx.get_q(0), mod.q(0), rtol=rtol, atol=1e-15)
        np.testing.assert_allclose(ctx.get_p(0), mod.p(0), rtol=rtol, atol=1e-15)
        np.testing.assert_allclose(ctx.expected_counts(), mod.sik(), rtol=rtol, atol=1e-12)
    ll_o = mod.loglik(0)
    assert abs(ctx.loglik(0) - ll_o) <= max(1e-8, 1e-12 * abs(ll_o))
    assert ctx.loglik_prefetch(0) == ctx.loglik(0)


def mixture_cases(n, seed):
    rs = np.random.default_rng(seed)
    sizes = [9, 17, 64, 65, 129, 257, 300]
    return [(int(rs.choice(sizes)), int(rs.choice(sizes)), int(rs.integers(1, 9)), int(rs.integers(1, 5)), int(rs.integers(2, 7)),
             float(rs.choice([0.0, 0.03])), int(rs.integers(0, 1 << 30))) for _ in range(n)]


@pytest.mark.parametrize("I,L,K,ploidy,maxal,missing,seed", mixture_cases(24 * SCALE, 7 + SEED_SHIFT))
def test_random_shapes_mixture_vs_oracle(ctx, I, L, K, ploidy, maxal, missing, seed):
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    _, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=lb)
    eta0 = np.full(K, 1.0 / K)
    opt = ob.make_options(admixture=0, lower_bound=lb, abs_error=0.0)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    mod.q(0)[...] = eta0
    mod.p(0)[...] = p0
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, admixture=0, lower_bound=lb)
    ctx.set_q(0, eta0)
    ctx.set_p(0, p0)
    for s in (1, 2):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= max(1e-8, 1e-12 * abs(mod.logL)), (s, ll, mod.logL)
        np.testing.assert_allclose(ctx.get_q(0), mod.q(0), rtol=1e-7, atol=1e-13)
        np.testing.assert_allclose(ctx.get_p(0), mod.p(0), rtol=1e-7, atol=1e-13)
        np.testing.assert_allclose(ctx.expected_counts(), mod.sik(), rtol=1e-7, atol=1e-12)
    # the log likelihood of the parameters after the second M step.  Mixture model: v_ik = log eta_k + sum n log p is a sum of
    # ploidy L terms of size 1-18 (|v| ~ 400 here), rounded differently by the chunked device sums (~1e-12 absolute); exp() turns
    # that ABSOLUTE error of v into a RELATIVE error of vik, the M step into one of P, and away from a stationary point
    # d logL ~ |logL| x that: a soak run (MC_FUZZ_SEED=301) found 2.4e-12 |logL| on 65 triploid individuals x 129 loci
    assert abs(ctx.loglik(0) - mod.loglik(0)) <= max(1e-8, 1e-11 * abs(mod.logL))


def draw_cases(n, seed):
    rs = np.random.default_rng(seed)
    return [(int(rs.integers(3, 700)), int(rs.integers(3, 900)), int(rs.choice([1, 2, 2, 3, 4, 4, 5, 6, 7, 8, 9])), int(rs.integers(1, 65)),
             int(rs.choice([2, 3, 4, 7, 12])), float(rs.choice([0.0, 0.0, 0.05])), int(rs.integers(0, 2)),
             int(rs.integers(0, 5000)), int(rs.integers(1, 1 << 31))) for _ in range(n)]


@pytest.mark.parametrize("I,L,ploidy,K,maxal,missing,constrained,skip,seed", draw_cases(40 * SCALE, 11 + SEED_SHIFT))
def test_random_device_draws_vs_host_stream(ctx, I, L, ploidy, K, maxal, missing, constrained, skip, seed):
    """random_allele_partition drawn on the device = drawn from the oracle's glibc stream on the host, for random sizes,
    seeds, stream offsets and every K up to 64 (the multiply-shift remainder).  The device form draws and counts tile by tile
    without storing an assignment (ploidy <= 8); the uploaded partition goes through the layout and counting kernels: two
    independent routes to the same counts, with missing copies, repeated (allele, cluster) pairs and both eta forms."""
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed % 1000, missing=missing)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=1e-8, eta_constrained=constrained)
    window, rng = ob.glibc_window(seed, skip)
    assign = ob.rand_mod(rng, I * L * ploidy, K)
    ctx.mstep_from_partition(assign, 0)
    ctx.mstep_from_rand_partition(window, 1)
    # equal_nan: with few haploid individuals and many clusters a (cluster, locus) pair can receive no copy at all, and the
    # first M step then divides 0 by 0 there, in the reference as here
    assert np.array_equal(ctx.get_q(0), ctx.get_q(1), equal_nan=True) and np.array_equal(ctx.get_p(0), ctx.get_p(1), equal_nan=True)
    assert np.array_equal(np.isnan(ctx.get_p(0)), np.isnan(ctx.get_p(1)))


def accel_cases(n, seed):
    rs = np.random.default_rng(seed)
    return [(int(rs.integers(40, 200)), int(rs.integers(60, 400)), int(rs.integers(2, 6)), int(rs.choice([1, 2, 2, 2, 4])),
             int(rs.integers(1, 5)), float(rs.choice([0.0, 0.02])), int(rs.choice([0, 0, 7, 30])), int(rs.integers(1, 1 << 20)))
            for _ in range(n)]


@pytest.mark.parametrize("I,L,K,ploidy,scheme,missing,max_iter,seed", accel_cases(16 * SCALE, 5 + SEED_SHIFT))
def test_random_accelerated_fits_batched_vs_cycle_by_cycle(I, L, K, ploidy, scheme, missing, max_iter, seed, monkeypatch):
    """Whole accelerated fits (-s 1..4) from the same random initialisation: device-side batches of cycles against the
    cycle-by-cycle host loop, bit for bit, with and without an iteration cap that can fire inside a cycle."""
    from multiclust_amd import host
    ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=4, seed=seed, missing=missing)
    out = []
    for batched in (True, False):
        if batched:
            monkeypatch.delenv("MC_NO_BATCH", raising=False)
        else:
            monkeypatch.setenv("MC_NO_BATCH", "1")
        fit = host.Fit(ua, geno, K, admixture=1, accel_scheme=scheme, verbosity=1, max_iter=max_iter)
        fit.initialize(seed)
        fit.em()
        m = fit.mod
        out.append((m.n_iter, m.converged, m.iter_stop, m.fatal, m.logL, fit.get_q(m.pindex), fit.get_p(m.pindex)))
        fit.close()
    a, b = out
    assert a[:5] == b[:5], (a[:5], b[:5])
    assert np.array_equal(a[5], b[5], equal_nan=True) and np.array_equal(a[6], b[6], equal_nan=True)


def variant_cases(n, seed):
    """round-2 kernel variants: K up to 64, projection off, lower bounds far below the default (reciprocal-per-cell kernels),
    all-biallelic data (scalar-row individual-side passes from K = 6 / 10)"""
    rs = np.random.default_rng(seed)
    sizes = [9, 33, 64, 65, 129, 257, 300, 513]
    out = []
    for _ in range(n):
        K = int(rs.choice([int(rs.integers(1, 13)), int(rs.integers(13, 33)), int(rs.integers(33, 65))], p=[0.5, 0.3, 0.2]))
        out.append((int(rs.choice(sizes)), int(rs.choice(sizes)), K, int(rs.choice([1, 2, 2, 2, 3, 4])), int(rs.choice([2, 2, 3, 5])),
                    float(rs.choice([0.0, 0.0, 0.04])), int(rs.choice([1, 1, 0])), float(rs.choice([1e-8, 1e-8, 1e-40, 1e-120])),
                    int(rs.integers(0, 1 << 30))))
    return out


@pytest.mark.parametrize("I,L,K,ploidy,maxal,missing,projection,bound,seed", variant_cases(60 * SCALE, 2 + SEED_SHIFT))
def test_random_kernel_variants_vs_oracle(ctx, I, L, K, ploidy, maxal, missing, projection, bound, seed):
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
    lb = ob.lib.mco_lower_bound(bound, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=max(lb, 1e-12))
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0, do_projection=projection)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, do_projection=projection, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    for s in (1, 2):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= max(1e-8, 1e-12 * abs(mod.logL)), (s, ll, mod.logL)
        rtol = 1e-11 if s == 1 else 1e-9
        np.testing.assert_allclose(ctx.get_q(0), mod.q(0), rtol=rtol, atol=1e-15)
        np.testing.assert_allclose(ctx.get_p(0), mod.p(0), rtol=rtol, atol=1e-15)
        np.testing.assert_allclose(ctx.expected_counts(), mod.sik(), rtol=rtol, atol=1e-12)
    ll_o = mod.loglik(0)
    assert abs(ctx.loglik(0) - ll_o) <= max(1e-8, 1e-12 * abs(ll_o))
    assert ctx.loglik_prefetch(0) == ctx.loglik(0)


def sim_cases(n, seed):
    rs = np.random.default_rng(seed)
    return [(int(rs.integers(5, 600)), int(rs.integers(5, 700)), int(rs.choice([1, 2, 2, 3, 4, 4, 5, 6, 8, 9])), int(rs.choice([1, 2, 3, 5, 7, 8, 9, 13, 33, 64])),
             int(rs.choice([2, 3, 4, 4, 5, 9])), int(rs.integers(0, 2)), int(rs.integers(0, 3000)), int(rs.integers(1, 1 << 30))) for _ in range(n)]


@pytest.mark.parametrize("I,L,ploidy,K,maxal,constrained,skip,seed", sim_cases(16 * SCALE, 3 + SEED_SHIFT))
def test_random_bootstrap_data_sets_device_vs_host(ctx, I, L, ploidy, K, maxal, constrained, skip, seed):
    """parametric_bootstrap_admixture (bootstrap.c:84-124) generated on the device from the stream's window against the host
    generator (pinned to the reference's own data sets by tests/test_bootstrap_cpu.py), byte for byte, for random shapes,
    ploidies, K up to 64, both eta forms and stream offsets; at most 4 alleles per locus with K <= 8 and ploidy <= 8 takes the
    tiled generator (thresholds in LDS, written straight into the device layouts), everything else the chunked one."""
    import ctypes as C
    from multiclust_amd import host
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed % 9973, missing=0.02)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    q, p = random_params(I, ua, K, seed=seed % 7919, lower_bound=lb)
    q[::3, 0] = lb
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
    lib.mc_srand(C.byref(rng), seed)
    for _ in range(skip):
        lib.mc_rand(C.byref(rng))
    gen = host.McSimulation()
    rng_dev = host.McRng.from_buffer_copy(rng)
    lib.mc_simulation_begin(C.byref(gen), C.byref(opt), C.byref(dat), K, q.ctypes.data, p.ctypes.data, C.byref(rng_dev))
    ref = np.empty((I, L, ploidy), dtype=np.uint8)
    lib.mc_bootstrap_genotypes(C.byref(opt), C.byref(dat), K, q.ctypes.data, p.ctypes.data, C.byref(rng), ref.ctypes.data)
    assert lib.mc_rand(C.byref(rng)) == lib.mc_rand(C.byref(rng_dev))
    ctx.simulate_genotypes(I, L, ploidy, ua, np.array(gen.window, dtype=np.uint32), K, q, p, eta_constrained=constrained)
    assert np.array_equal(ctx.get_genotypes(), ref)


def randem_cases(n, seed):
    rs = np.random.default_rng(seed)
    return [(int(rs.integers(20, 260)), int(rs.integers(20, 500)), int(rs.choice([1, 2, 2, 3, 4])), int(rs.choice([1, 2, 3, 4, 6, 9, 17, 40])),
             int(rs.choice([2, 3, 5, 12, 30])), float(rs.choice([0.0, 0.03])), int(rs.integers(0, 2)), int(rs.integers(1, 1 << 30))) for _ in range(n)]


@pytest.mark.parametrize("I,L,ploidy,K,maxal,missing,constrained,seed", randem_cases(12 * SCALE, 4 + SEED_SHIFT))
def test_random_randem_initialisations_vs_oracle(I, L, ploidy, K, maxal, missing, constrained, seed):
    """Rand-EM (host walk of the loci + device assignment, counts and scoring) against the oracle's serial restatement: same
    winner and parameters (ratios of small integers: identical), same stream position, for random shapes, K and allele counts
    on both sides of K (loci with fewer alleles than clusters draw nothing, the others draw centers with retries)."""
    from multiclust_amd import host
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed % 9973, missing=missing)
    fit = host.Fit(ua, geno, K, admixture=1, eta_constrained=constrained, verbosity=1, initialization_procedure=1, n_rand_em_init=2)
    rng = fit.initialize(seed)
    opt = ob.make_options(eta_constrained=constrained, lower_bound=fit.opt.lower_bound, fused=1)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    org, ll = mod.init_randem(seed, 2)
    assert fit.lib.mc_rand(rng) == ob.lib.mco_rand(org)
    if abs(ll[0] - ll[1]) > 1e-6 * abs(ll[0]) or K == 1:      # (a near-tie between the two candidates could go either way)
        np.testing.assert_allclose(fit.get_q(0), mod.q(0), rtol=1e-15, atol=0)
        np.testing.assert_allclose(fit.get_p(0), mod.p(0), rtol=1e-15, atol=0)
    fit.close()
