"""GPU parity tests proper (-m gpu): the HIP path, called through the C-ABI, against
(a) the golden vectors dumped from the reference itself and (b) the CPU oracle on seeded inputs.

Tolerances (north_star): Q/P within 1e-6 relative, logL within 1e-8 absolute at config-1 scale
(|logL| ~ 5e4); at larger |logL| the reference's own double summation error exceeds 1e-8
(SURVEY.md App. D), so the bound there is 1e-12 * |logL|.  Tests state tighter bounds where observed.
"""
import numpy as np
import pytest

import multiclust_amd as mc
import oracle_bind as ob
from golden_util import Golden
from synth import make_dataset, random_params

pytestmark = pytest.mark.gpu

ADMIX = ["c1_admix_k3", "multi_admix_k4", "tetra_admix_k3", "missing_admix_k3", "multi_admix_k1",
         "allmiss_admix_k2",       # loci without any allele column (every individual missing): first / last of a block of 8, last locus
         "mono_admix_k3",          # monomorphic loci, with and without missing copies
         "haploid_admix_k2", "triploid_admix_k3", "hexaploid_admix_k2",
         "manyallele_admix_k2"]    # loci with up to 35 alleles: the dense fallback kernels against the reference itself
ADMIX_C = ["multi_admix_c_k3", "missing_admix_c_k2"]      # shared mixing proportions (-c), the second on data with missing copies
# --projection (unobserved allele columns -- the phantom slot of loci with missing data, alleles with no carrier in a cluster --
# keep p = 0 for every k) and --bound 1e-120: the reciprocal-per-cell kernel variants (mchip_set_model)
UNPROJECTED = ["missing_admix_k3_noproj", "multi_admix_k4_noproj", "rare_admix_k3_noproj", "tetra_admix_k3_noproj",
               "multi_admix_k4_tinybound"]


@pytest.fixture(scope="module")
def ctx():
    c = mc.Context(0)
    yield c
    c.close()


def setup_case(ctx, g, n_secants=1):
    ctx.set_genotypes(g.ua, g.geno)
    ctx.set_model(g.K, admixture=g.m["admixture"], eta_constrained=g.m["eta_constrained"],
                  do_projection=g.m["do_projection"], lower_bound=g.lower_bound, n_secants=n_secants)
    ctx.set_q(0, g.q("q0"))
    ctx.set_p(0, g.p("p0"))


def close(a, b, rtol, atol):
    np.testing.assert_allclose(a, b, rtol=rtol, atol=atol)


@pytest.mark.parametrize("name", ADMIX + ADMIX_C)
def test_param_roundtrip(ctx, name):
    g = Golden(name)
    setup_case(ctx, g)
    assert np.array_equal(ctx.get_p(0), g.p("p0"))
    assert np.array_equal(ctx.get_q(0), g.q("q0"))


@pytest.mark.parametrize("name", ADMIX + ADMIX_C + UNPROJECTED)
def test_em_steps_vs_reference_golden(ctx, name):
    g = Golden(name)
    setup_case(ctx, g)
    ll_ref = g.f64("em_ll.f64")
    snaps = set(g.m["snapshots"])
    for s in range(1, g.m["n_em_steps"] + 1):
        ll = ctx.em_step(0, 0)
        assert abs(ll - ll_ref[s - 1]) <= 1e-8, (s, ll, ll_ref[s - 1])
        if s in snaps:
            # one step: rounding only; later: EM dynamics amplify last-bit differences (bound 10x inside 1e-6)
            rtol, atol = (1e-12, 1e-15) if s == 1 else (1e-7, 1e-12)
            close(ctx.get_q(0), g.q("step%d" % s), rtol, atol)
            close(ctx.get_p(0), g.p("step%d" % s), rtol, atol)
            close(ctx.expected_counts(), g.sik("step%d" % s), rtol, max(atol, 1e-13))
    assert abs(ctx.loglik(0) - g.m["ll_after_em"]) <= 1e-8


@pytest.mark.parametrize("name", ADMIX)
def test_first_mstep_from_partition_same_seed(ctx, name):
    """random_initialize_admixture with the libc-compatible stream drawn on the host: identical Q0/P0."""
    g = Golden(name)
    ctx.set_genotypes(g.ua, g.geno)
    ctx.set_model(g.K, lower_bound=g.lower_bound)
    draws = np.array(ob.glibc_rand(g.m["seed"], g.I * g.L * g.ploidy), dtype=np.int64)
    assign = (draws % g.K).astype(np.uint8)
    ctx.mstep_from_partition(assign, 0)
    assert np.array_equal(ctx.expected_counts(), g.sik("init"))
    close(ctx.get_q(0), g.q("q0"), 1e-15, 1e-18)
    close(ctx.get_p(0), g.p("p0"), 1e-15, 1e-18)


@pytest.mark.parametrize("name", ADMIX)
def test_first_mstep_partition_drawn_on_device(ctx, name):
    """The same with the rand() stream generated on the device (mchip_mstep_from_rand_partition): the reference's
    own initial parameters for its seed."""
    g = Golden(name)
    ctx.set_genotypes(g.ua, g.geno)
    ctx.set_model(g.K, lower_bound=g.lower_bound)
    window, _ = ob.glibc_window(g.m["seed"])
    ctx.mstep_from_rand_partition(window, 0)
    assert np.array_equal(ctx.expected_counts(), g.sik("init"))
    close(ctx.get_q(0), g.q("q0"), 1e-15, 1e-18)
    close(ctx.get_p(0), g.p("p0"), 1e-15, 1e-18)


@pytest.mark.parametrize("I,L,ploidy,K,skip", [
    (700, 1000, 2, 8, 0),        # 1.4e6 draws: more than one block of 256 chunks (both jump tables used)
    (301, 997, 4, 3, 1234),      # ragged tail chunk, stream already advanced, K not a power of two
    (513, 640, 2, 32, 77),       # K = 32
    (1500, 64, 2, 47, 9),        # K > 32: the partition's cluster flags are 64 bits wide (enough copies that no cluster is empty at a locus)
    (129, 400, 3, 7, 5),         # odd ploidy
    (64, 100, 2, 1, 0),          # K = 1: rand() % 1
    (900, 900, 2, 2, 31),
])
def test_device_draw_equals_host_draw(ctx, I, L, ploidy, K, skip):
    """Bit-identical initial Q, P and per-individual counts whether the partition is drawn on the host from the
    libc-compatible stream and uploaded, or generated on the device from the stream's 31-word window."""
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=4, seed=I + K, missing=0.01)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=lb)
    window, rng = ob.glibc_window(20250117, skip)
    assign = ob.rand_mod(rng, I * L * ploidy, K)
    ctx.mstep_from_partition(assign, 0)
    ctx.mstep_from_rand_partition(window, 1)
    sik_dev = ctx.expected_counts()
    ctx.mstep_from_partition(assign, 2)
    assert np.array_equal(ctx.expected_counts(), sik_dev)
    assert np.array_equal(ctx.get_q(0), ctx.get_q(1))
    assert np.array_equal(ctx.get_p(0), ctx.get_p(1))


@pytest.mark.parametrize("I,L,K,ploidy,maxal,missing", [
    (300, 1000, 8, 2, 4, 0.0),       # config-3 shape, small
    (257, 999, 5, 2, 2, 0.02),       # ragged sizes (not multiples of 8 / 64 / 256), missing data
    (64, 200, 16, 4, 6, 0.0),        # tetraploid generic path, max K
    (33, 77, 2, 3, 3, 0.05),         # odd ploidy
    (2000, 300, 3, 2, 3, 0.0),       # many individuals, few loci
])
def test_em_steps_vs_oracle_synthetic(ctx, I, L, K, ploidy, maxal, missing):
    ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=maxal, seed=I + L, missing=missing)
    lb = ob.lib.mco_lower_bound(1e-8, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=3, lower_bound=lb)
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0)
    data = ob.Data(I, L, ploidy, ua, geno)
    mod = ob.Model(data, opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    for s in range(1, 6):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= max(1e-8, 1e-12 * abs(mod.logL)), (s, ll, mod.logL)
        rtol, atol = (1e-11, 1e-15) if s == 1 else (1e-8, 1e-13)
        close(ctx.get_q(0), mod.q(0), rtol, atol)
        close(ctx.get_p(0), mod.p(0), rtol, atol)
        close(ctx.expected_counts(), mod.sik(), rtol, 1e-12)
    assert abs(ctx.loglik(0) - mod.loglik(0)) <= max(1e-8, 1e-12 * abs(mod.logL))
    assert abs(ctx.e_step(0) - mod.loglik(0)) <= max(1e-8, 1e-12 * abs(mod.logL))


def test_run_to_run_bitwise_reproducible(ctx):
    g = Golden("multi_admix_k4")
    outs = []
    for _ in range(2):
        setup_case(ctx, g)
        lls = [ctx.em_step(0, 0) for _ in range(5)]
        outs.append((lls, ctx.get_q(0), ctx.get_p(0)))
    assert outs[0][0] == outs[1][0]
    assert np.array_equal(outs[0][1], outs[1][1]) and np.array_equal(outs[0][2], outs[1][2])


def test_invalid_inputs_are_rejected(ctx):
    g = Golden("multi_admix_k4")
    bad = g.geno.copy()
    bad[0, 0, 0] = 200                      # allele index >= uniquealleles[0]
    with pytest.raises(mc.HipError):
        ctx.set_genotypes(g.ua, bad)
    ctx.set_genotypes(g.ua, g.geno)
    with pytest.raises(mc.HipError):
        ctx.set_model(65)                   # K > MCHIP_MAX_K
    with pytest.raises(mc.HipError):
        ctx.set_model(0)


# mixlong: 1 800 loci, exp(max_k v_ik) underflows, so the reference's logL_mixture takes its scaling branch
# (log_likelihood.c:209-224) for ll_after_em
MIX = ["multi_mix_k3", "missing_mix_k2", "mixlong_mix_k3", "mixslow_mix_k3_s3", "allmiss_mix_k2", "hexaploid_mix_k2"]


@pytest.mark.parametrize("name", MIX)
def test_mixture_em_steps_vs_reference_golden(ctx, name):
    """e_step_mixture / m_step_mixture / logL_mixture (em_alg.c:763-1011, log_likelihood.c:157-232)."""
    g = Golden(name)
    setup_case(ctx, g)
    ll_ref = g.f64("em_ll.f64")
    snaps = set(g.m["snapshots"])
    for s in range(1, g.m["n_em_steps"] + 1):
        ll = ctx.em_step(0, 0)
        assert abs(ll - ll_ref[s - 1]) <= 1e-8, (s, ll, ll_ref[s - 1])
        if s in snaps:
            rtol, atol = (1e-11, 1e-14) if s == 1 else (1e-7, 1e-12)
            close(ctx.get_q(0), g.q("step%d" % s), rtol, atol)
            close(ctx.get_p(0), g.p("step%d" % s), rtol, atol)
            close(ctx.expected_counts(), g.sik("step%d" % s), 1e-7, 1e-12)      # vik
    assert abs(ctx.loglik(0) - g.m["ll_after_em"]) <= 1e-8


@pytest.mark.parametrize("I,L,K", [(300, 700, 6), (120, 300, 40)])
def test_mixture_vs_oracle_synthetic(ctx, I, L, K):
    ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=5, seed=77, missing=0.01)
    lb = ob.lib.mco_lower_bound(1e-8, I, 2)
    _, p0 = random_params(I, ua, K, seed=3, lower_bound=lb)
    eta0 = np.full(K, 1.0 / K)
    opt = ob.make_options(admixture=0, lower_bound=lb, abs_error=0.0)
    mod = ob.Model(ob.Data(I, L, 2, ua, geno), opt, K)
    mod.q(0)[...] = eta0
    mod.p(0)[...] = p0
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, admixture=0, lower_bound=lb)
    ctx.set_q(0, eta0)
    ctx.set_p(0, p0)
    for s in range(1, 5):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= 1e-12 * abs(mod.logL) + 1e-8
        # posteriors close to 0/1 amplify last-bit differences of the log-sums; 10x inside north_star's 1e-6
        close(ctx.get_q(0), mod.q(0), 1e-7, 1e-13)
        close(ctx.get_p(0), mod.p(0), 1e-7, 1e-13)
        close(ctx.expected_counts(), mod.sik(), 1e-7, 1e-12)
    assert abs(ctx.loglik(0) - mod.loglik(0)) <= 1e-12 * abs(mod.logL) + 1e-8


@pytest.mark.parametrize("I,L,K,ploidy,maxal,missing,opts", [
    (150000, 9, 4, 2, 3, 0.0, {}),                                  # extreme aspect ratios: nine loci, one block of 8 and a remainder ...
    (5, 300000, 3, 2, 4, 0.01, {}),                                 # ... and five individuals: one partly filled workgroup per locus chunk
    (70, 150, 3, 2, 40, 0.0, {}),                                   # > 32 alleles at a locus: dense fallback kernels
    (300, 40, 3, 2, 120, 0.0, {}),                                  # > 64 alleles: projection with byte flags instead of a 64-bit mask
    (90, 210, 4, 1, 3, 0.02, {}),                                   # haploid
    (50, 120, 3, 3, 4, 0.03, {}),                                   # triploid: generic copy loop, 2-bit counts
    (40, 100, 2, 6, 5, 0.0, {}),                                    # hexaploid: generic copy loop, 4-bit counts
    (130, 260, 5, 2, 3, 0.01, {"do_projection": 0}),                # projection off: per-copy log-product checks
    (130, 260, 5, 2, 3, 0.0, {"lower_bound": 1e-40}),               # tiny lower bound: per-copy log-product checks
    (100, 300, 16, 2, 4, 0.0, {}),                                  # K = 16 (largest tuned K)
    (100, 300, 1, 2, 4, 0.0, {}),                                   # K = 1
    (80, 200, 23, 2, 4, 0.01, {}),                                  # K > 16: untuned but correct
    (64, 128, 32, 4, 5, 0.0, {}),                                   # K = 32, tetraploid
    (90, 160, 33, 2, 4, 0.01, {}),                                  # K > 32: the projection's fixed-entry mask is 64 bits wide
    (70, 140, 48, 2, 3, 0.0, {"eta_constrained": 1}),
    (66, 130, 64, 4, 4, 0.02, {}),                                  # K = MCHIP_MAX_K, tetraploid, missing data
    (120, 240, 4, 2, 3, 0.02, {"eta_constrained": 1}),              # shared eta (-c)
    (120, 240, 4, 4, 4, 0.02, {"eta_constrained": 1}),
    (513, 64, 7, 2, 2, 0.0, {}),                                    # more individuals than a 512-wide tile, few loci
    (9, 11, 2, 2, 2, 0.0, {}),                                      # tiny: fewer individuals / loci than any tile
    (200, 500, 8, 2, 2, 0.0, {}),                                   # every locus biallelic: scalar-row log-likelihood pass (K >= 6)
    (150, 401, 12, 2, 2, 0.0, {}),                                  # ... and scalar-row S-side pass (K >= 10)
    (150, 401, 12, 2, 2, 0.02, {}),                                 # ... with missing copies (no phantom slot in this generator)
])
def test_em_steps_vs_oracle_paths(ctx, I, L, K, ploidy, maxal, missing, opts):
    """Every kernel variant (dense fallback, generic ploidy, safe log-product path, shared eta, K extremes)
    against the oracle's fused-order restatement."""
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=7 * I + L, missing=missing)
    user_lb = opts.get("lower_bound", 1e-8)
    lb = ob.lib.mco_lower_bound(user_lb, I, ploidy)
    q0, p0 = random_params(I, ua, K, seed=3, lower_bound=max(lb, 1e-12))
    con = opts.get("eta_constrained", 0)
    proj = opts.get("do_projection", 1)
    if con:
        q0 = q0[0].copy()
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0, eta_constrained=con, do_projection=proj)
    mod = ob.Model(ob.Data(I, L, ploidy, ua, geno), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, eta_constrained=con, do_projection=proj, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    for s in range(1, 4):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= max(1e-8, 1e-12 * abs(mod.logL)), (s, ll, mod.logL)
        rtol, atol = (1e-11, 1e-15) if s == 1 else (1e-8, 1e-13)
        close(ctx.get_q(0), mod.q(0), rtol, atol)
        close(ctx.get_p(0), mod.p(0), rtol, atol)
        close(ctx.expected_counts(), mod.sik(), rtol, 1e-12)
    assert abs(ctx.loglik(0) - mod.loglik(0)) <= max(1e-8, 1e-12 * abs(mod.logL))


def test_accel_vector_ops_vs_numpy(ctx):
    """mchip_secant / step_dots / secant_dots / accel_update against numpy on the fetched parameters."""
    g = Golden("multi_admix_k4")
    setup_case(ctx, g, n_secants=2)
    ctx.em_step(0, 1)
    ctx.em_step(1, 2)
    x0q, x0p, x1q, x1p, x2q, x2p = ctx.get_q(0), ctx.get_p(0), ctx.get_q(1), ctx.get_p(1), ctx.get_q(2), ctx.get_p(2)
    ctx.secant(0, 0, 1, 0)
    ctx.secant(1, 0, 2, 1)
    ctx.secant(0, 1, 2, 1)
    ctx.secant(1, 1, 1, 0)
    u = np.concatenate([(x1q - x0q).ravel(), (x1p - x0p).ravel()])
    v = np.concatenate([(x2q - x1q).ravel(), (x2p - x1p).ravel()])
    d = ctx.step_dots(0)
    np.testing.assert_allclose(d, [u @ u, u @ (v - u), (v - u) @ (v - u)], rtol=1e-12)
    d2 = ctx.secant_dots(0, 1)                     # u_0 . u_1 and u_0 . v_1, with u_1 = v, v_1 = u
    np.testing.assert_allclose(d2, [u @ v, u @ u], rtol=1e-12)
    s = -2.5
    ctx.set_model  # (no-op reference to keep flake quiet)
    lb = g.lower_bound
    ctx.accel_update(2, 0, 0, s, 0)
    want_q = x0q - 2 * s * (x1q - x0q) + s * s * ((x2q - x1q) - (x1q - x0q))
    got_q = ctx.get_q(2)
    for i in range(g.I):
        np.testing.assert_allclose(got_q[i], ob.michelot(want_q[i], lb), rtol=1e-13, atol=1e-16)
    want_p = x0p - 2 * s * (x1p - x0p) + s * s * ((x2p - x1p) - (x1p - x0p))
    got_p = ctx.get_p(2)
    off = 0
    for M in g.ua:
        for k in range(g.K):
            np.testing.assert_allclose(got_p[k, off:off + M], ob.michelot(want_p[k, off:off + M], lb), rtol=1e-13, atol=1e-16)
        off += M


@pytest.mark.parametrize("name", ["rare_admix_k3_noproj", "rare_admix_k3_bs", "multi_admix_k4_noproj"])
def test_em_on_bootstrap_replicate_lacking_alleles(ctx, name):
    """What run_bootstrap's fits do (multiclust.c:675-708): a data set simulated from fitted parameters (on the device,
    byte-identical to the reference's), initialised from the OBSERVED haplotypes (rnd_init.c:471), then EM steps on the
    simulated counts.  The replicate lacks some alleles (bs_absent_columns): with --projection their columns are 0 for
    every k, with projection they sit at the lower bound."""
    g = Golden(name)
    assert g.m["bs_absent_columns"] > 0
    window, _ = ob.glibc_window(g.m["bootstrap_seed"])
    ctx.simulate_genotypes(g.I, g.L, g.ploidy, g.ua, window, g.K, g.q("bs"), g.p("bs"))
    sim = ctx.get_genotypes()
    from test_bootstrap_cpu import counts_of, golden_bootstrap
    assert np.array_equal(counts_of(sim, g.ua), golden_bootstrap(g))
    ctx.set_init_genotypes(g.geno)
    ctx.set_model(g.K, do_projection=g.m["do_projection"], lower_bound=g.lower_bound)
    w0, _ = ob.glibc_window(g.m["seed"])
    ctx.mstep_from_rand_partition(w0, 0)
    close(ctx.get_q(0), g.q("bsinit"), 1e-15, 1e-18)
    close(ctx.get_p(0), g.p("bsinit"), 1e-15, 1e-18)
    ll_ref = g.f64("bs_em_ll.f64")
    for s in range(g.m["bs_em_steps"]):
        ll = ctx.em_step(0, 0)
        assert np.isfinite(ll) and abs(ll - ll_ref[s]) <= 1e-8, (s, ll, ll_ref[s])
    close(ctx.get_q(0), g.q("bsstep"), 1e-9, 1e-13)
    close(ctx.get_p(0), g.p("bsstep"), 1e-9, 1e-13)
    close(ctx.expected_counts(), g.sik("bsstep"), 1e-9, 1e-12)
    assert abs(ctx.loglik(0) - g.m["bs_ll_after_em"]) <= 1e-8
    if not g.m["do_projection"]:
        absent = counts_of(sim, g.ua).sum(axis=0) == 0
        assert np.all(ctx.get_p(0)[:, absent] == 0.0)


@pytest.mark.parametrize("ploidy", [2, 3, 4])
@pytest.mark.parametrize("bound,projection", [(1e-40, 1), (1e-8, 0), (1e-8, 1)])
def test_every_K_both_individual_side_variants_agree(ctx, ploidy, bound, projection):
    """K = 1..64 in every individual-side kernel family (general / reciprocal-per-cell, diploid / tetraploid / generic ploidy):
    the accumulating pass (E step) and the stand-alone log-likelihood pass are different template instances and must return
    the same log likelihood.  A fuzz soak found one instance (tetraploid, reciprocal-per-cell, K = 52) that hipcc 7.2
    miscompiled when it spilled vector registers to accumulation registers (-inf / NaN); the kernel objects are built with
    -amdgpu-spill-vgpr-to-agpr=false since."""
    for K in range(1, 65):
        I, L = 64, 40
        ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=3, seed=5, missing=0.02)
        lb = ob.lib.mco_lower_bound(bound, I, ploidy)
        q0, p0 = random_params(I, ua, K, seed=6, lower_bound=max(lb, 1e-12))
        ctx.set_genotypes(ua, geno)
        ctx.set_model(K, lower_bound=lb, do_projection=projection)
        ctx.set_q(0, q0)
        ctx.set_p(0, p0)
        a, b = ctx.loglik(0), ctx.e_step(0)
        assert np.isfinite(a) and a == b, (K, a, b)


@pytest.mark.parametrize("I,L,K,pl,maxal,miss", [(300, 257, 49, 2, 4, 0.0), (129, 300, 64, 2, 3, 0.03), (257, 129, 56, 4, 4, 0.0),
                                                 (200, 200, 33, 3, 5, 0.02), (300, 300, 28, 2, 4, 0.0), (150, 400, 27, 2, 4, 0.01)])
def test_large_k_mixture_and_shared_eta_models(ctx, I, L, K, pl, maxal, miss):
    """K on both sides of the lane split of the sparse individual pass (one lane per individual up to K = 27, two up to 48,
    four and 256-lane workgroups above) for the models that share its kernels and workgroup size: the mixture model
    (k_mix_gather / k_mix_finalize) and the admixture model with shared mixing proportions (-c), two EM steps against the oracle."""
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=pl, max_alleles=maxal, seed=K, missing=miss)
    lb = ob.lib.mco_lower_bound(1e-8, I, pl)
    _, p0 = random_params(I, ua, K, seed=K + 1, lower_bound=lb)
    eta0 = np.full(K, 1.0 / K)
    opt = ob.make_options(admixture=0, lower_bound=lb, abs_error=0.0)
    mod = ob.Model(ob.Data(I, L, pl, ua, geno), opt, K)
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
    q0, p0 = random_params(I, ua, K, seed=K + 2, lower_bound=lb)
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0, eta_constrained=1)
    mod = ob.Model(ob.Data(I, L, pl, ua, geno), opt, K)
    mod.q(0)[...] = q0[0]
    mod.p(0)[...] = p0
    ctx.set_model(K, eta_constrained=1, lower_bound=lb)
    ctx.set_q(0, np.ascontiguousarray(q0[0]))
    ctx.set_p(0, p0)
    for s in (1, 2):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        assert abs(ll - mod.logL) <= max(1e-8, 1e-12 * abs(mod.logL)), ("c", s, ll, mod.logL)
        np.testing.assert_allclose(ctx.get_q(0), mod.q(0), rtol=1e-8, atol=1e-14)
        np.testing.assert_allclose(ctx.get_p(0), mod.p(0), rtol=1e-8, atol=1e-14)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("poison", ["nan", "-inf", "+inf"])
def test_mixture_loglik_without_an_ordinary_term_is_nan_not_a_stall(ctx, poison):
    """logL_mixture halves a scale until exp() stops overflowing (log_likelihood.c:212-221); for an individual whose every
    log-sum is NaN or -inf (or one +inf) its maximum is infinite, inf / 2 = inf, and the reference never leaves that loop -- a
    stuck process on the CPU, a wave the stream never gets back from on the device (k_mix_finalize).  No reference answer
    exists: the individual's term is NaN here, so the log likelihood is NaN and stop() ends the run on "nan"
    (em_alg.c:106-110).  The call returns; finite rows beside it and the next call are unaffected."""
    I, L, K = 70, 90, 3
    ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=3, seed=5)
    _, p0 = random_params(I, ua, K, seed=9, lower_bound=1e-8)
    eta = np.full(K, 1.0 / K)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, admixture=0, do_projection=0, lower_bound=1e-8)
    ctx.set_q(0, eta)
    ctx.set_p(0, p0)
    good = ctx.loglik(0)
    assert np.isfinite(good)
    bad = p0.copy()
    col = int(geno[3, 0, 0])                                      # an allele individual 3 carries at the first locus
    # a negative p (an extrapolated point with the projection off) has log = NaN; p = 0 has log = -inf in every cluster;
    # p = +inf cannot come out of any step, but the loop's other way out of the ordinary numbers is closed too
    bad[:, col] = {"nan": -0.25, "-inf": 0.0, "+inf": np.inf}[poison]
    ctx.set_p(0, bad)
    ll = ctx.loglik(0)                                            # log_likelihood(): mode 1 of k_mix_finalize
    assert np.isnan(ll)
    ctx.set_p(0, p0)
    assert ctx.loglik(0) == good                                  # the stream came back, and with the same bits


def test_rows_of_individuals_without_copies_are_reported_per_slot_and_survive_a_round_trip(ctx):
    """An individual without a single observed copy has mixing proportions 0 / 0 = NaN in the reference from its first M step on
    (em_alg.c:685-690); the device holds a finite row and mchip_get_q reports NaN for the SLOTS that stand for it -- the slot an M
    step wrote, not the one it read -- and mchip_set_q takes such a row back (warm start, checkpoint): the NaN is stored as the
    finite row, the slot goes on reporting NaN, and nothing it is multiplied into turns NaN."""
    I, L, K = 60, 80, 3
    ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=3, seed=9)
    geno[7] = 0xFF
    lb = ob.lib.mco_lower_bound(1e-8, I, 2)
    q0, p0 = random_params(I, ua, K, seed=4, lower_bound=lb)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    ll = ctx.em_step(0, 1)                                  # E step on slot 0, M step into slot 1
    assert np.isfinite(ll)
    q1 = ctx.get_q(1)
    assert np.isnan(q1[7]).all() and np.isfinite(np.delete(q1, 7, axis=0)).all()
    assert np.array_equal(ctx.get_q(0), q0)                 # the slot the step read still holds what the caller put there
    ctx.set_q(2, q1)                                        # the round trip: a NaN row goes back in
    ctx.set_p(2, ctx.get_p(1))
    assert np.isnan(ctx.get_q(2)[7]).all()
    ll2 = ctx.em_step(2, 2)
    p2, q2 = ctx.get_p(2), ctx.get_q(2)
    assert np.isfinite(ll2) and np.isfinite(p2).all() and np.isnan(q2[7]).all() and np.isfinite(np.delete(q2, 7, axis=0)).all()
    ll1 = ctx.em_step(1, 1)                                 # the same step without the round trip: same bits
    assert ll1 == ll2 and np.array_equal(ctx.get_p(1), p2)
    finite = q1.copy()
    finite[7] = 1.0 / K
    ctx.set_q(2, finite)                                    # a finite row is stored and reported as given
    assert np.array_equal(ctx.get_q(2), finite)
