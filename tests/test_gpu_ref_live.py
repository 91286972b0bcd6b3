"""-m gpu: the hot path against the reference's OWN em(), live, at sizes the committed goldens do not reach.
oracle/_ref/ref_time (oracle/ref_time.c linked with the reference's unmodified sources in the build container; the binary
travels with the tree) runs em() from given parameters on flat arrays; the same data, parameters, scheme and -T go through
mc_em on the GPU.  Drawn cases: 200-3 000 individuals, 300-6 000 loci, ploidy 1-6, up to 12 alleles per locus, K 2-12 and, in
every fourth case, 13-64 (the lane-split kernels),
admixture / -c / mixture, plain EM and every acceleration scheme, 3-8 iterations, a quarter of the cases with 3 % and a quarter
with 20 % missing copies.  Bounds: log likelihood 1e-8 absolute at the
scale of config 1 and 5e-12 relative beyond (the reference's own running sum of 1e7 terms is good to about 1e-12 at
|logL| = 7e7: the largest difference seen in 136 cases was 1.3e-12); Q and P entries above 1e-6: 1e-9 relative after plain EM
iterations -- three orders inside north_star's 1e-6 -- and north_star's 1e-6 itself after accelerated cycles, whose
extrapolation x - 2su + s^2(v - u) multiplies last-bit differences of u and v by the step size squared (largest seen: 8.5e-8).
Skipped where the binary is absent."""
import json
import os
import subprocess

import numpy as np
import pytest

from multiclust_amd import host
from synth import make_dataset, random_params

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_TIME = os.path.join(ROOT, "oracle", "_ref", "ref_time")


def draw_cases(n, seed):
    rs = np.random.default_rng(seed)
    out = []
    for c in range(n):
        big = c % 8 == 7
        I = int(rs.integers(1500, 3000)) if big else int(rs.integers(200, 900))
        L = int(rs.integers(3000, 6000)) if big else int(rs.integers(300, 2000))
        ploidy = int(rs.choice([1, 2, 2, 2, 3, 4, 4, 6]))
        maxal = int(rs.choice([2, 3, 4, 4, 6, 12]))
        K = int(rs.choice([2, 3, 4, 5, 7, 8, 8, 12])) if c % 4 else int(rs.choice([13, 16, 21, 24, 29, 33, 40, 47, 52, 57, 64]))
        if K > 12:                                            # the lane-split kernel families (two or four lanes per individual,
            I, L = min(I, 400), min(L, 600)                   # two lanes per allele column from K = 37)
        model = str(rs.choice(["admix", "admix", "admix", "admix_c", "mix"]))
        scheme = int(rs.choice([0, 0, 1, 2, 3, 3, 4, 5, 6]))
        iters = int(rs.integers(3, 9))
        out.append((c, I, L, ploidy, maxal, K, model, scheme, iters, int(rs.integers(1, 10 ** 6))))
    return out


@pytest.mark.skipif(not os.access(REF_TIME, os.X_OK), reason="oracle/_ref/ref_time not built (needs /root/reference at build time)")
@pytest.mark.parametrize("c,I,L,ploidy,maxal,K,model,scheme,iters,seed",
                         draw_cases(int(os.environ.get("MC_LIVE_CASES", "16")), 20250117 + int(os.environ.get("MC_LIVE_SEED", "0"))))
def test_mc_em_against_the_reference_em_on_drawn_shapes(c, I, L, ploidy, maxal, K, model, scheme, iters, seed, tmp_path):
    missing = [0.0, 0.0, 0.03, 0.2][seed % 4]
    ua, geno = make_dataset(I, L, max(K, 2), ploidy=ploidy, max_alleles=maxal, seed=seed, missing=missing)
    if missing and seed % 8 < 6:      # as the reference's reader shapes it: one more allele slot, which nothing matches, at loci with missing copies
        ua = (ua + (geno == 0xFF).any(axis=(0, 2)).astype(np.int32)).astype(np.int32)
    admixture, constrained = int(model != "mix"), int(model == "admix_c")
    lb = min(1e-8, 0.5 / (I * ploidy))                                     # multiclust.c:812-815
    q0, p0 = random_params(I, ua, K, seed=seed + 1, lower_bound=lb)
    if constrained or not admixture:
        q0 = np.ascontiguousarray(q0.mean(axis=0) / q0.mean(axis=0).sum())
    d = str(tmp_path)
    np.ascontiguousarray(ua, dtype=np.int32).tofile(d + "/ua.i32")
    geno.tofile(d + "/geno.u8")
    q0.tofile(d + "/q0.f64")
    p0.tofile(d + "/p0.f64")
    flags = (["-a"] if admixture else []) + (["-c"] if constrained else []) + ["-k", str(K)] + (["-s", str(scheme)] if scheme else [])
    res = subprocess.run([REF_TIME, d, str(I), str(L), str(ploidy), str(K), str(iters - 1), "--", "-f", "x"] + flags,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    if res.returncode == 0 and not res.stdout.strip():
        # the reference left through one of its exit(0)s: a log likelihood lower by an ulp at a fixed point reached to the last bit
        # (one row of eta), or "nan"
        pytest.skip("the reference stopped itself: %s" % res.stderr.strip()[-120:])
    assert res.returncode == 0, res.stderr[-1000:]
    ref = json.loads(res.stdout)
    q_ref = np.fromfile(d + "/q_ref.f64").reshape(q0.shape)
    p_ref = np.fromfile(d + "/p_ref.f64").reshape(K, -1)
    fit = host.Fit(ua, geno, K, admixture=admixture, eta_constrained=constrained, accel_scheme=scheme, verbosity=1,
                   abs_error=1e-300, rel_error=0.0, max_iter=iters - 1)
    assert fit.opt.lower_bound == ref["lower_bound"]
    fit.set_params(q0, p0)
    fit.em()
    if fit.mod.fatal:
        # "never converged" (abs_error 1e-300) on a fit that is at its fixed point to the last bit: the next step's log likelihood
        # is lower by an ulp in one summation order and not in the other, and stop() ends the run there (em_alg.c:113-120).  Seen
        # with the mixture model, whichever program it hits (the reference's exit is handled above)
        assert abs(fit.mod.logL - ref["logL"]) <= 5e-12 * abs(ref["logL"])
        fit.close()
        pytest.skip("this build's run ended on a log likelihood one ulp lower, at the fixed point")
    if fit.mod.converged and fit.mod.n_iter < ref["n_iter"]:
        # the other face of the same coin (round 4 soak, mixture model, K = 3, quasi-Newton: 7 iterations against 8): at the fixed
        # point two successive log likelihoods are the SAME double in this summation order -- |difference| = 0 <= 1e-300, so
        # converged() says yes -- and one ulp apart in the reference's, which therefore runs on to its -T cap
        assert abs(fit.mod.logL - ref["logL"]) <= 5e-12 * abs(ref["logL"]), (fit.mod.logL, ref["logL"])
        early = ref["n_iter"] - fit.mod.n_iter
        fit.close()
        pytest.skip("this build's run converged exactly (two equal log likelihoods) at the fixed point, %d iteration(s) before the cap" % early)
    assert fit.mod.n_iter == ref["n_iter"], (fit.mod.n_iter, ref["n_iter"])
    assert abs(fit.mod.logL - ref["logL"]) <= max(1e-8, 5e-12 * abs(ref["logL"])), (fit.mod.logL, ref["logL"])
    gq, gp = fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex)
    fit.close()
    for got, want in ((gq, q_ref), (gp, p_ref)):
        big = want > 1e-6
        assert np.max(np.abs(got - want)[big] / want[big]) <= (1e-6 if scheme else 1e-9)
        assert np.max(np.abs(got - want)[~big], initial=0.0) <= (1e-10 if scheme else 1e-12)


@pytest.mark.skipif(not os.access(REF_TIME, os.X_OK), reason="oracle/_ref/ref_time not built (needs /root/reference at build time)")
@pytest.mark.parametrize("scheme", [0, 3, 1, 4, 5])
def test_an_individual_without_a_single_observed_copy(scheme, tmp_path):
    """two individuals of thirty have every copy missing.  The reference gives them mixing proportions 0 / 0 = NaN in its first
    M step (em_alg.c:685-690) and carries on: they add to no sum, zero-count cells being skipped one by one.  With NaN in the
    secants its step size is NaN and every accelerated cycle falls back to its EM iterate (accel_em.c:58-62; with q > 1 the
    quasi-Newton point is NaN and refused).  Here the device keeps a finite row for such an individual, mchip_get_q reports NaN
    as the reference has it, and mc_step_size / mc_qn_accelerated_update supply the NaN: same rows NaN, same iteration count,
    same log likelihood, same parameters for everybody else -- under individual and shared (-c) mixing proportions and the
    mixture model.  (Found by the live comparison: the run used to end in "nan" at its third iteration.)"""
    I, L, ploidy, K = 30, 40, 2, 3
    ua, geno = make_dataset(I, L, K, ploidy=ploidy, max_alleles=4, seed=5, missing=0.05)
    geno[4] = 0xFF
    geno[20] = 0xFF
    ua = (ua + (geno == 0xFF).any(axis=(0, 2)).astype(np.int32)).astype(np.int32)
    lb = min(1e-8, 0.5 / (I * ploidy))
    q0, p0 = random_params(I, ua, K, seed=6, lower_bound=lb)
    d = str(tmp_path)
    ua.tofile(d + "/ua.i32")
    geno.tofile(d + "/geno.u8")
    p0.tofile(d + "/p0.f64")
    for flags, q_start in ((["-a"], q0), (["-a", "-c"], q0[0] / q0[0].sum()), ([], q0[0] / q0[0].sum())):
        np.ascontiguousarray(q_start).tofile(d + "/q0.f64")
        res = subprocess.run([REF_TIME, d, str(I), str(L), str(ploidy), str(K), "7", "--", "-f", "x"] + flags + ["-k", str(K)] +
                             (["-s", str(scheme)] if scheme else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        assert res.returncode == 0 and res.stdout.strip(), res.stderr[-500:]
        ref = json.loads(res.stdout)
        q_ref, p_ref = np.fromfile(d + "/q_ref.f64").reshape(q_start.shape), np.fromfile(d + "/p_ref.f64").reshape(K, -1)
        fit = host.Fit(ua, geno, K, admixture=int("-a" in flags), eta_constrained=int("-c" in flags), accel_scheme=scheme, verbosity=1,
                       abs_error=1e-300, rel_error=0.0, max_iter=7)
        fit.set_params(np.ascontiguousarray(q_start), p0)
        fit.em()
        if fit.mod.fatal:           # (one row of eta: the fixed point reached to the last bit, see above)
            fit.close()
            continue
        assert fit.mod.n_iter == ref["n_iter"] and abs(fit.mod.logL - ref["logL"]) <= 1e-8, (flags, fit.mod.n_iter, ref["n_iter"])
        gq, gp = fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex)
        fit.close()
        if flags == ["-a"]:
            assert np.isnan(q_ref[[4, 20]]).all() and np.isnan(gq[[4, 20]]).all() and np.signbit(gq[[4, 20]]).all()
            keep = np.delete(np.arange(I), [4, 20])
            gq, q_ref = gq[keep], q_ref[keep]
        assert not np.isnan(q_ref).any() and not np.isnan(p_ref).any()
        np.testing.assert_allclose(gq, q_ref, rtol=1e-9 if not scheme else 1e-6, atol=1e-13)
        np.testing.assert_allclose(gp, p_ref, rtol=1e-9 if not scheme else 1e-6, atol=1e-13)
