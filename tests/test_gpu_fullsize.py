"""-m gpu: parity at BASELINE.json's full sizes through size-independent properties (and the oracle where it finishes
in seconds): config 2 (2 000 x 20 000, K = 5) against the oracle on the whole problem, config 3 (10 000 x 100 000,
M_l <= 4, K = 8) against exact identities of the E/M steps and against the oracle on a slice of individuals."""
import os

import numpy as np
import pytest

import multiclust_amd as mc
import oracle_bind as ob
from synth import make_dataset, random_params

pytestmark = pytest.mark.gpu


def fast_geno(I, L, ploidy, maxal, seed, chunk=500):
    rng = np.random.default_rng(seed)
    ua = rng.integers(2, maxal + 1, size=L).astype(np.int32) if maxal > 2 else np.full(L, 2, np.int32)
    geno = np.empty((I, L, ploidy), dtype=np.uint8)
    for i0 in range(0, I, chunk):
        i1 = min(I, i0 + chunk)
        geno[i0:i1] = (rng.integers(0, 1 << 20, size=(i1 - i0, L, ploidy), dtype=np.int32) % ua[None, :, None]).astype(np.uint8)
    return ua, geno


REF_TIME = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref", "ref_time")


@pytest.mark.skipif(not os.access(REF_TIME, os.X_OK), reason="oracle/_ref/ref_time not built")
@pytest.mark.parametrize("scheme,iters", [(0, 3), (3, 4)])
def test_config2_full_size_against_the_reference_em(scheme, iters, tmp_path):
    """BASELINE.json's configs[1] whole -- 2 000 diploid individuals x 20 000 biallelic loci, K = 5 -- through the reference's own
    em() (oracle/_ref/ref_time: its unmodified sources, 3.2 GB of diklm, a few seconds per iteration on one host core) and through
    mc_em from the same parameters: plain EM (the configuration's own -s 0) and two SQUAREM-3 cycles.  north_star's bounds are
    1e-6 relative on Q / P and 1e-8 absolute on the log likelihood at config-1 scale; here |logL| = 5e7, where the reference's
    own running sum of 8e7 terms is good to about 1e-12 relative: 5e-12 is asked, and 1e-9 on the parameters."""
    import json
    import subprocess
    from multiclust_amd import host
    I, L, K = 2000, 20000, 5
    ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=2, seed=20250119, chunk=128)
    lb = min(1e-8, 0.5 / (I * 2))
    q0, p0 = random_params(I, ua, K, seed=4, lower_bound=lb)
    d = str(tmp_path)
    np.ascontiguousarray(ua, dtype=np.int32).tofile(d + "/ua.i32")
    geno.tofile(d + "/geno.u8")
    q0.tofile(d + "/q0.f64")
    p0.tofile(d + "/p0.f64")
    res = subprocess.run([REF_TIME, d, str(I), str(L), "2", str(K), str(iters - 1), "--", "-f", "x", "-a", "-k", str(K)] +
                         (["-s", str(scheme)] if scheme else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-1000:]
    ref = json.loads(res.stdout)
    fit = host.Fit(ua, geno, K, admixture=1, accel_scheme=scheme, verbosity=1, abs_error=1e-300, rel_error=0.0, max_iter=iters - 1)
    fit.set_params(q0, p0)
    fit.em()
    assert fit.mod.n_iter == ref["n_iter"] == iters
    assert abs(fit.mod.logL - ref["logL"]) <= 5e-12 * abs(ref["logL"]), (fit.mod.logL, ref["logL"])
    gq, gp = fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex)
    fit.close()
    np.testing.assert_allclose(gq, np.fromfile(d + "/q_ref.f64").reshape(I, K), rtol=1e-9 if not scheme else 1e-7, atol=1e-13)
    np.testing.assert_allclose(gp, np.fromfile(d + "/p_ref.f64").reshape(K, -1), rtol=1e-9 if not scheme else 1e-7, atol=1e-13)


@pytest.mark.skipif(not os.access(REF_TIME, os.X_OK), reason="oracle/_ref/ref_time not built")
@pytest.mark.parametrize("I,L,ploidy,K,scheme,iters,what", [
    (10000, 2500, 2, 8, 3, 4, "config 3 / 4: every individual, the first 2 500 of its 100 000 loci, two SQUAREM-3 cycles"),
    (5000, 4000, 4, 8, 0, 3, "config 5, alternative model: every individual, the first 4 000 of its 50 000 loci, plain EM"),
    (5000, 4000, 4, 7, 0, 3, "config 5, null model"),
])
def test_locus_slices_of_configs_3_and_5_against_the_reference_em(I, L, ploidy, K, scheme, iters, what, tmp_path):
    """the reference cannot hold configs 3-5 (0.5 TB of diklm at config 3): every individual and as many loci as fit a few GB go
    through its em() and through mc_em from the same parameters.  Bounds as in the config-2 test."""
    import json
    import subprocess
    from multiclust_amd import host
    ua, geno = fast_geno(I, L, ploidy, 4, seed=I + L + K)
    lb = min(1e-8, 0.5 / (I * ploidy))
    q0, p0 = random_params(I, ua, K, seed=6, lower_bound=lb)
    d = str(tmp_path)
    np.ascontiguousarray(ua, dtype=np.int32).tofile(d + "/ua.i32")
    geno.tofile(d + "/geno.u8")
    q0.tofile(d + "/q0.f64")
    p0.tofile(d + "/p0.f64")
    res = subprocess.run([REF_TIME, d, str(I), str(L), str(ploidy), str(K), str(iters - 1), "--", "-f", "x", "-a", "-k", str(K)] +
                         (["-s", str(scheme)] if scheme else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900)
    assert res.returncode == 0, res.stderr[-1000:]
    ref = json.loads(res.stdout)
    fit = host.Fit(ua, geno, K, admixture=1, accel_scheme=scheme, verbosity=1, abs_error=1e-300, rel_error=0.0, max_iter=iters - 1)
    fit.set_params(q0, p0)
    fit.em()
    assert fit.mod.n_iter == ref["n_iter"] == iters
    assert abs(fit.mod.logL - ref["logL"]) <= 5e-12 * abs(ref["logL"]), (fit.mod.logL, ref["logL"])
    gq, gp = fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex)
    fit.close()
    np.testing.assert_allclose(gq, np.fromfile(d + "/q_ref.f64").reshape(I, K), rtol=1e-9 if not scheme else 1e-7, atol=1e-13)
    np.testing.assert_allclose(gp, np.fromfile(d + "/p_ref.f64").reshape(K, -1), rtol=1e-9 if not scheme else 1e-7, atol=1e-13)


def check_simplex(q, p, ua, lb):
    assert np.all(q >= lb * (1 - 1e-12)) and np.all(p >= lb * (1 - 1e-12))
    np.testing.assert_allclose(q.sum(axis=1), 1.0, rtol=0, atol=1e-12)
    off = np.concatenate([[0], np.cumsum(ua)])
    sums = np.add.reduceat(p, off[:-1], axis=1)
    np.testing.assert_allclose(sums, 1.0, rtol=0, atol=1e-12)


def test_config2_full_size_against_oracle():
    I, L, K = 2000, 20000, 5
    ua, geno = make_dataset(I, L, K, ploidy=2, max_alleles=2, seed=20250119, chunk=128)
    lb = ob.lib.mco_lower_bound(1e-8, I, 2)
    q0, p0 = random_params(I, ua, K, seed=4, lower_bound=lb)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=lb)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0)
    mod = ob.Model(ob.Data(I, L, 2, ua, geno), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    prev = -np.inf
    for s in range(3):
        mod.em_step()
        ll = ctx.em_step(0, 0)
        # |logL| ~ 5e7: the reference's own sequential double sum is only good to ~1e-13 |logL| there
        assert abs(ll - mod.logL) <= 1e-12 * abs(mod.logL), (s, ll, mod.logL)
        assert ll > prev
        prev = ll
        q, p = ctx.get_q(0), ctx.get_p(0)
        np.testing.assert_allclose(q, mod.q(0), rtol=1e-9, atol=1e-13)
        np.testing.assert_allclose(p, mod.p(0), rtol=1e-9, atol=1e-13)
        # every allele copy is shared out over the K clusters: sum_k S_ik = number of observed copies of individual i
        np.testing.assert_allclose(ctx.expected_counts().sum(axis=1), (geno != 255).sum(axis=(1, 2)), rtol=1e-13)
        check_simplex(q, p, ua, lb)
    ctx.close()


def test_config3_full_size_properties_and_slice_against_oracle():
    I, L, K = 10000, 100000, 8
    ua, geno = fast_geno(I, L, 2, 4, seed=20250120)
    T = int(ua.sum())
    lb = ob.lib.mco_lower_bound(1e-8, I, 2)
    q0, p0 = random_params(I, ua, K, seed=5, lower_bound=lb)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=lb, n_secants=1)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    ll0 = ctx.em_step(0, 1)                                    # E(x0), M -> slot 1
    sik = ctx.expected_counts()
    np.testing.assert_allclose(sik.sum(axis=1), 2.0 * L, rtol=1e-13)       # no missing data: 2 L copies per individual
    assert abs(sik.sum() - 2.0 * I * L) <= 1e-12 * 2.0 * I * L
    # slice of individuals against the oracle: S_ik and q' of individual i depend on its own genotype, q_i and P only
    sel = np.array([0, 1, 17, 4095, 4096, 5000, 9998, 9999])
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0)
    mod = ob.Model(ob.Data(len(sel), L, 2, ua, geno[sel]), opt, K)
    mod.q(0)[...] = q0[sel]
    mod.p(0)[...] = p0
    mod.em_step()
    np.testing.assert_allclose(sik[sel], mod.sik(), rtol=1e-11, atol=1e-12)
    q1, p1 = ctx.get_q(1), ctx.get_p(1)
    np.testing.assert_allclose(q1[sel], mod.q(0), rtol=1e-10, atol=1e-14)
    check_simplex(q1, p1, ua, lb)
    # log likelihood: stand-alone pass = E-step value (same parameters), and EM increases it
    assert ctx.loglik(0) == ll0
    ll1 = ctx.em_step(1, 1)
    assert ll1 > ll0
    assert abs(ctx.loglik_prefetch(1) - ctx.loglik(1)) == 0.0
    # linearity of the secant / dot kernels at full size
    ctx.secant(0, 0, 1, 0)
    ctx.secant(1, 0, 1, 0)
    d = ctx.step_dots(0)                                       # v == u: u.(v-u) = 0, |v-u|^2 = 0
    assert d[0] > 0 and d[1] == 0.0 and d[2] == 0.0
    q2, p2 = ctx.get_q(1), ctx.get_p(1)                        # slot 1 now holds the second iterate
    u2 = float(((q2 - q0) ** 2).sum() + ((p2 - p0) ** 2).sum())
    assert abs(d[0] - u2) <= 1e-9 * u2
    ctx.close()


def columns_of(ua, loci):
    off = np.concatenate([[0], np.cumsum(ua)])
    return np.concatenate([np.arange(off[l], off[l + 1]) for l in loci])


def check_step_and_cycle_on_slices(ctx, ua, geno, K, lb, q0, p0):
    """One EM step and one SQUAREM-3 cycle at full size, checked where the oracle finishes in seconds:
      * S_ik and q'_i of an individual depend on its own genotype, q_i and P: oracle on a slice of individuals;
      * p'_kl. of a locus depends on every individual's genotype at that locus, Q and p_.l.: oracle on a slice of loci
        with ALL individuals (the N-side sums of em_alg.c:706-752);
      * the cycle (accel_em.c:35-114): u, v, the three dot products and the step from numpy on the downloaded iterates, the
        extrapolated point x0 - 2 s u + s^2 (v - u) elementwise, its projection by the oracle's michelot_project on the
        same slices; log likelihoods by the halves identity."""
    I, L, pl = geno.shape
    rs = np.random.default_rng(11)
    sel_i = np.unique(np.concatenate([[0, 1, I // 2 - 1, I // 2, I - 2, I - 1], rs.integers(0, I, 10)]))
    sel_l = np.unique(np.concatenate([[0, 1, 7, 8, L // 2, L - 9, L - 8, L - 1], rs.integers(0, L, 24)]))
    cols = columns_of(ua, sel_l)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    ll0 = ctx.em_step(0, 1)
    sik = ctx.expected_counts()
    np.testing.assert_allclose(sik.sum(axis=1), float(pl) * L, rtol=1e-13)
    q1, p1 = ctx.get_q(1), ctx.get_p(1)
    opt = ob.make_options(lower_bound=lb, fused=1, abs_error=0.0)
    mod_i = ob.Model(ob.Data(len(sel_i), L, pl, ua, geno[sel_i]), opt, K)
    mod_i.q(0)[...] = q0[sel_i]
    mod_i.p(0)[...] = p0
    mod_i.em_step()
    np.testing.assert_allclose(sik[sel_i], mod_i.sik(), rtol=1e-11, atol=1e-12)
    np.testing.assert_allclose(q1[sel_i], mod_i.q(0), rtol=1e-10, atol=1e-14)
    mod_l = ob.Model(ob.Data(I, len(sel_l), pl, ua[sel_l], np.ascontiguousarray(geno[:, sel_l])), opt, K)
    mod_l.q(0)[...] = q0
    mod_l.p(0)[...] = p0[:, cols]
    mod_l.em_step()
    np.testing.assert_allclose(p1[:, cols], mod_l.p(0), rtol=1e-10, atol=1e-14)
    check_simplex(q1, p1, ua, lb)
    assert ctx.loglik(0) == ll0
    # second EM step, secants, step size
    ll1 = ctx.em_step(1, 2)
    assert ll1 > ll0
    q2, p2 = ctx.get_q(2), ctx.get_p(2)
    mod_i.q(0)[...] = q1[sel_i]                                       # the slice's own M step saw only its individuals: the
    mod_i.p(0)[...] = p1                                              # second step starts from the full p' (ours, checked above)
    mod_i.em_step()
    np.testing.assert_allclose(q2[sel_i], mod_i.q(0), rtol=1e-10, atol=1e-14)
    mod_l.q(0)[...] = q1
    mod_l.p(0)[...] = p1[:, cols]
    mod_l.em_step()
    np.testing.assert_allclose(p2[:, cols], mod_l.p(0), rtol=1e-10, atol=1e-14)
    ctx.secant(0, 0, 1, 0)
    ctx.secant(1, 0, 2, 1)
    d = ctx.step_dots(0)
    uq, up, vq, vp = q1 - q0, p1 - p0, q2 - q1, p2 - p1
    utu = float((uq * uq).sum() + (up * up).sum())
    utvu = float((uq * (vq - uq)).sum() + (up * (vp - up)).sum())
    vutvu = float(((vq - uq) ** 2).sum() + ((vp - up) ** 2).sum())
    np.testing.assert_allclose(d, [utu, utvu, vutvu], rtol=1e-9)
    s = -np.sqrt(d[0] / d[2])                                          # SQUAREM S3 (accel_em.c:227-229), clamp 236-237
    if s > -1:
        s = -1.0
    ctx.accel_update(1, 0, 0, s, 0)                                    # slot 1 := projected extrapolation
    qx, px = ctx.get_q(1), ctx.get_p(1)
    want_q = q0 - 2 * s * uq + s * s * (vq - uq)
    want_p = p0 - 2 * s * up + s * s * (vp - up)
    for i in sel_i:
        np.testing.assert_allclose(qx[i], ob.michelot(want_q[i], lb), rtol=1e-12, atol=1e-16)
    off = np.concatenate([[0], np.cumsum(ua)])
    for l in sel_l:
        for k in range(K):
            np.testing.assert_allclose(px[k, off[l]:off[l + 1]], ob.michelot(want_p[k, off[l]:off[l + 1]], lb), rtol=1e-12, atol=1e-16)
    check_simplex(qx, px, ua, lb)
    llx = ctx.loglik_prefetch(1)
    assert llx == ctx.loglik(1)
    # the same cycle as ONE device-side batch (mchip_accel_run: stop rule, step size and accept test on the device; for diploid
    # data the dual individual pass takes log L of the second iterate together with the extrapolated point's E step): slot 0
    # ends on the extrapolated point if its log likelihood beats the second iterate's, else on the second iterate -- the very
    # arrays the call-by-call sequence above produced
    import ctypes as C
    from multiclust_amd import hip
    ll2 = ctx.loglik(2)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    st = hip.RunState(logL=-np.inf, abs_error=1e-300, n_iter=0)
    rc = ctx.lib.mchip_accel_run(ctx.h, 0, 3, 1, C.byref(st))
    assert rc == 0 and st.fatal == 0 and st.n_iter == 2 and st.logL == ll1
    want = (qx, px) if llx > ll2 else (q2, p2)
    assert np.array_equal(ctx.get_q(0), want[0]) and np.array_equal(ctx.get_p(0), want[1])
    return qx, px, llx


def test_config3_pside_and_squarem_cycle_on_slices():
    I, L, K = 10000, 100000, 8
    ua, geno = fast_geno(I, L, 2, 4, seed=20250121)
    lb = ob.lib.mco_lower_bound(1e-8, I, 2)
    q0, p0 = random_params(I, ua, K, seed=6, lower_bound=lb)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    assert ctx.data_counts()[1] == I * L * 2
    ctx.set_model(K, lower_bound=lb, n_secants=1)
    qx, px, llx = check_step_and_cycle_on_slices(ctx, ua, geno, K, lb, q0, p0)
    ctx.close()
    halves = []
    for sl in (slice(0, I // 2), slice(I // 2, I)):                    # log likelihood of the extrapolated point: halves identity
        c = mc.Context(0)
        c.set_genotypes(ua, geno[sl])
        c.set_model(K, lower_bound=lb)
        c.set_q(0, qx[sl])
        c.set_p(0, px)
        halves.append(c.loglik(0))
        c.close()
    assert abs(sum(halves) - llx) <= 1e-11 * abs(llx)


@pytest.mark.parametrize("K", [7, 8])
def test_config5_full_size_tetraploid(K):
    """BASELINE.json configs[4]: 5 000 tetraploid x 50 000 loci, K = 7 and 8 (the two models of -b 200 -k 8): the 4-bit
    packed column pass and the tetraploid sparse pass at size; then one parametric-bootstrap data set of that size
    generated on the device from the fitted point, byte for byte against the host generator (bootstrap.c:84-124) on the
    first and the last individuals."""
    import ctypes as C
    from multiclust_amd import host
    I, L, pl = 5000, 50000, 4
    ua, geno = fast_geno(I, L, pl, 4, seed=20250122 + K)
    lb = ob.lib.mco_lower_bound(1e-8, I, pl)
    q0, p0 = random_params(I, ua, K, seed=7, lower_bound=lb)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    cells, copies = ctx.data_counts()
    assert copies == I * L * pl and I * L <= cells <= copies
    ctx.set_model(K, lower_bound=lb, n_secants=1)
    qx, px, llx = check_step_and_cycle_on_slices(ctx, ua, geno, K, lb, q0, p0)
    # bootstrap data set from (qx, px)
    hl = host.load()
    opt = host.McOptions()
    hl.mc_make_options(C.byref(opt))
    opt.admixture = 1
    rng = host.McRng()
    hl.mc_srand(C.byref(rng), 777)
    window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
    ctx.simulate_genotypes(I, L, pl, ua, window, K, qx, px)
    sim = ctx.get_genotypes()
    assert sim.shape == (I, L, pl) and np.all(sim < ua[None, :, None])
    ua32 = np.ascontiguousarray(ua, dtype=np.int32)
    n = 3
    for first in (0, I - n):
        r2 = host.McRng.from_buffer_copy(rng)
        hl.mc_rng_jump(C.byref(r2), 2 * first * L * pl)                # two draws per allele copy (bootstrap.c:95-120)
        part = np.empty((n, L, pl), dtype=np.uint8)
        qs = np.ascontiguousarray(qx[first:first + n])
        dat = host.McData(n, L, pl, ua32.ctypes.data, part.ctypes.data)
        hl.mc_bootstrap_genotypes(C.byref(opt), C.byref(dat), K, qs.ctypes.data, px.ctypes.data, C.byref(r2), part.ctypes.data)
        assert np.array_equal(sim[first:first + n], part), first
    # the generated data set is a working data set
    ctx.set_model(K, lower_bound=lb)
    ctx.set_q(0, qx)
    ctx.set_p(0, px)
    ll = ctx.em_step(0, 0)
    assert np.isfinite(ll) and ctx.em_step(0, 0) > ll
    np.testing.assert_allclose(ctx.expected_counts().sum(axis=1), float(pl) * L, rtol=1e-13)
    ctx.close()


def test_beyond_2_pow_32_allele_copies():
    """22 000 x 100 000 diploid = 4.4e9 allele copies: more bytes per genotype layout than a launch has work-items.
    Identities that hold at any size: the genotype comes back from the device layouts; logL(all individuals) =
    logL(first half) + logL(second half) for the same P; the device-drawn partition of the last individuals equals the
    oracle's glibc stream advanced to their first draw (4.4e9 draws in: the high jump polynomials)."""
    import ctypes as C
    from multiclust_amd import host
    I, L, P, K = 22000, 100000, 2, 3
    rs = np.random.default_rng(3)
    ua = rs.integers(2, 5, L).astype(np.int32)
    base = rs.integers(0, 12, (64, L, P), dtype=np.uint8)              # 64 random individuals, tiled with a per-row shift
    geno = np.empty((I, L, P), dtype=np.uint8)
    for i0 in range(0, I, 64):
        n = min(64, I - i0)
        geno[i0:i0 + n] = (base[:n] + np.uint8(i0 // 64 % 12)) % ua[None, :, None].astype(np.uint8)
    q0, p0 = random_params(I, ua, K, seed=2)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    assert np.array_equal(ctx.get_genotypes(), geno)
    ctx.set_model(K, lower_bound=1e-8)
    ctx.set_q(0, q0)
    ctx.set_p(0, p0)
    ll_full = ctx.loglik(0)
    ctx.em_step(0, 1)
    assert abs(ctx.expected_counts().sum() - I * L * P) <= 1e-9 * I * L * P
    # device-drawn partition, last individuals (all loci heterozygous or not: d_iklm is an indicator, so a homozygote whose
    # two copies draw the same cluster counts once)
    hl = host.load()
    rng = host.McRng()
    hl.mc_srand(C.byref(rng), 99)
    window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
    ctx.mstep_from_rand_partition(window, 2)
    cnt = ctx.expected_counts()
    tail = 3
    hl.mc_rng_jump(C.byref(rng), (I - tail) * L * P)
    d = np.fromiter((hl.mc_rand(C.byref(rng)) % K for _ in range(tail * L * P)), dtype=np.int64, count=tail * L * P).reshape(tail, L, P)
    g = geno[I - tail:]
    exp = np.zeros((tail, K))
    for k in range(K):
        hit = d == k
        exp[:, k] = hit.sum(axis=(1, 2)) - (hit[:, :, 0] & hit[:, :, 1] & (g[:, :, 0] == g[:, :, 1])).sum(axis=1)
    assert np.array_equal(cnt[I - tail:], exp)
    ctx.close()
    lls = []
    for sl in (slice(0, I // 2), slice(I // 2, I)):
        c = mc.Context(0)
        c.set_genotypes(ua, geno[sl])
        c.set_model(K, lower_bound=1e-8)
        c.set_q(0, q0[sl])
        c.set_p(0, p0)
        lls.append(c.loglik(0))
        c.close()
    assert abs(sum(lls) - ll_full) <= 1e-11 * abs(ll_full), (lls, ll_full)


def test_config4_units_are_independent_of_who_fits_them():
    """BASELINE.json configs[3] at size: units of one data set sharded u -> rank u mod N.  Unit u must be the serial program's
    initialisation u whoever fits it and whatever that context fitted before: unit 3 (its partition starts 6e9 draws into the
    rand() stream, past 2^32) fitted after units 0 and 1 by one context, and alone by a fresh context, give the same bits;
    its device-drawn partition equals the host generator jumped to its first draw on the last individuals."""
    import ctypes as C
    from multiclust_amd import host
    I, L, K = 10000, 100000, 8
    ua, geno = fast_geno(I, L, 2, 4, seed=20250123)
    opts = dict(admixture=1, accel_scheme=3, verbosity=1, abs_error=1e-300, max_iter=5)      # -T 5: 3 SQUAREM cycles
    a = host.Fit(ua, geno, K, **opts)
    res_a = {u: a.fit_unit(1234567, u) for u in (0, 1, 3)}
    qa, pa = a.get_q(a.mod.pindex), a.get_p(a.mod.pindex)
    a.close()
    b = host.Fit(ua, geno, K, **opts)
    r3 = b.fit_unit(1234567, 3)
    assert (r3.logL, r3.n_iter, r3.pindex) == (res_a[3].logL, res_a[3].n_iter, res_a[3].pindex) and r3.n_iter == 6
    assert np.array_equal(b.get_q(b.mod.pindex), qa) and np.array_equal(b.get_p(b.mod.pindex), pa)
    assert len({res_a[u].logL for u in res_a}) == 3                                     # three different initialisations
    # the partition unit 3 starts from: expected counts of the hard partition on the last individuals
    rng = host.McRng()
    b.lib.mc_srand(C.byref(rng), 1234567)
    b.lib.mc_rng_jump(C.byref(rng), 3 * I * L * 2)
    window = np.array([rng.r[(rng.f + t) % 31] for t in range(31)], dtype=np.int64).astype(np.uint32)
    ctx = mc.Context(0)
    ctx.set_genotypes(ua, geno)
    ctx.set_model(K, lower_bound=1e-8)
    ctx.mstep_from_rand_partition(window, 0)
    cnt = ctx.expected_counts()
    tail = 2
    b.lib.mc_rng_jump(C.byref(rng), (I - tail) * L * 2)
    d = np.fromiter((b.lib.mc_rand(C.byref(rng)) % K for _ in range(tail * L * 2)), dtype=np.int64, count=tail * L * 2).reshape(tail, L, 2)
    g = geno[I - tail:]
    exp = np.zeros((tail, K))
    for k in range(K):
        hit = d == k
        exp[:, k] = hit.sum(axis=(1, 2)) - (hit[:, :, 0] & hit[:, :, 1] & (g[:, :, 0] == g[:, :, 1])).sum(axis=1)
    assert np.array_equal(cnt[I - tail:], exp)
    ctx.close()
    b.close()
