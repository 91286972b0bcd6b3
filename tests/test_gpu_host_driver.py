"""-m gpu: the plain-C host side (multiclust_amd/host/mc_em.c: mc_em, mc_stop, mc_accelerated_em_step ...)
driving the HIP path, against the reference's own full runs (golden) -- iteration counts, convergence
flags, ring index, log likelihood, final iterate, expected counts."""
import numpy as np
import pytest

from golden_util import Golden
from multiclust_amd import host

pytestmark = pytest.mark.gpu


def make_fit(g, accel=0, **kw):
    if g.name.endswith("tinybound"):
        kw.setdefault("lower_bound", 1e-120)       # --bound on the fixture's command line
    fit = host.Fit(g.ua, g.geno, g.K, admixture=g.m["admixture"], eta_constrained=g.m["eta_constrained"],
                   do_projection=g.m["do_projection"], accel_scheme=accel, verbosity=1, **kw)
    assert fit.opt.lower_bound == g.lower_bound
    fit.set_params(g.q("q0"), g.p("p0"))
    return fit


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "tetra_admix_k3", "missing_admix_k3",
                                  "multi_admix_k1", "multi_admix_c_k3", "multi_mix_k3", "missing_mix_k2"])
def test_em_to_convergence(name):
    g = Golden(name)
    fit = make_fit(g)
    fit.em()
    m = fit.mod
    assert m.fatal == 0
    # the stopping iteration may shift by one when |delta logL| hovers at abs_error
    assert abs(m.n_iter - g.m["em_run_n_iter"]) <= 1, (m.n_iter, g.m["em_run_n_iter"])
    assert m.converged == g.m["em_run_converged"]
    assert abs(m.logL - g.m["em_run_logL"]) <= 2e-4       # both within abs_error=1e-4 of the fixed point
    if m.n_iter == g.m["em_run_n_iter"]:
        assert abs(m.logL - g.m["em_run_logL"]) <= 1e-8
        np.testing.assert_allclose(fit.get_q(m.pindex), g.q("emrun"), rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(fit.get_p(m.pindex), g.p("emrun"), rtol=1e-6, atol=1e-10)
        np.testing.assert_allclose(fit.expected_counts(), g.sik("emrun"), rtol=1e-6, atol=1e-9)
    fit.close()


@pytest.mark.parametrize("name", ["c1_admix_k3_tight", "multi_admix_k4_tight"])
def test_long_em_run_stays_on_the_reference_trajectory(name):
    """-E 1e-10: the reference iterates 11 469 / 19 212 times.  At that tolerance the stopping iteration is decided by
    the last bits of the log likelihood (the reference stops when two successive sequential sums happen to agree to
    1e-10; with tree-ordered sums the same tail runs a few thousand iterations longer and ends 4e-6 higher), so the
    comparison is made at the reference's own iteration count: after that many EM iterations the log likelihood agrees
    within 1e-8 absolute and Q/P within 1e-6 relative (north_star's tolerances)."""
    g = Golden(name)
    n = g.m["em_run_n_iter"]
    fit = make_fit(g, abs_error=g.m["abs_error"], max_iter=n - 1)     # -T n runs n + 1 iterations (em_alg.c:150)
    fit.em()
    m = fit.mod
    assert m.fatal == 0 and m.n_iter == n and m.iter_stop == 1
    assert abs(m.logL - g.m["em_run_logL"]) <= 1e-8, (m.logL, g.m["em_run_logL"])
    np.testing.assert_allclose(fit.get_q(m.pindex), g.q("emrun"), rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(fit.get_p(m.pindex), g.p("emrun"), rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(fit.expected_counts(), g.sik("emrun"), rtol=1e-6, atol=1e-9)
    fit.close()


# mixture model (accel_em.c:444-541 over vetak): mixslow converges slowly enough for 5-7 cycles; mixlong's log likelihoods
# come from logL_mixture's scaling branch (log_likelihood.c:209-224)
MIX_ACCEL = ["mixslow_mix_k3_s1", "mixslow_mix_k3_s2", "mixslow_mix_k3_s3", "mixslow_mix_k3_qn1", "mixlong_mix_k3",
             "mixlong_mix_k3_s1", "multi_mix_k3"]


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "multi_admix_k4_s1", "multi_admix_k4_s2",
                                  "multi_admix_k3_qn1", "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3",
                                  "multi_admix_k4_tinybound"] + MIX_ACCEL)
def test_accelerated_cycles_trace(name):
    """First cycles of SQUAREM / QN1 against the reference's recorded emll, step size, ll, accept, ring index."""
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"])
    trace = g.f64("accel_trace.f64").reshape(-1, 8)
    ring_in_sync = True
    for c in range(min(5, len(trace))):
        stop = fit.accelerated_em_step()
        m = fit.mod
        assert not stop and m.fatal == 0
        assert abs(m.last_emll - trace[c, 0]) <= 1e-8, c
        # When the step size is clamped to s = -1 the extrapolated point IS the EM iterate (x0 + 2u + (v-u)), so
        # the reference's accept test ll > emll (accel_em.c:90) is decided by the last bit of two equal sums:
        # a genuine tie.  Values must still agree; which ring slot holds them may not.
        tie = abs(trace[c, 2] - trace[c, 0]) <= 1e-9 * abs(trace[c, 0])
        if trace[c, 7]:
            assert abs(m.last_step - trace[c, 1]) <= 1e-7 * abs(trace[c, 1]), (c, m.last_step, trace[c, 1])
            assert abs(m.last_ll - trace[c, 2]) <= 1e-7 * max(1.0, abs(trace[c, 2]) * 1e-3), c
            if not tie:
                assert m.last_accepted == trace[c, 3], c
            elif m.last_accepted != trace[c, 3]:
                ring_in_sync = False
        assert m.n_iter == trace[c, 4]
        if ring_in_sync:
            assert m.pindex == trace[c, 6], c
        assert abs(m.logL - trace[c, 5]) <= 1e-8
        if c == 0:
            np.testing.assert_allclose(fit.get_q(m.pindex), g.q("cycle1"), rtol=1e-7, atol=1e-12)
            np.testing.assert_allclose(fit.get_p(m.pindex), g.p("cycle1"), rtol=1e-7, atol=1e-12)
    fit.close()


@pytest.mark.parametrize("name", ["multi_admix_k3_qn2", "multi_admix_k3_qn3", "mixslow_mix_k3_qn2", "mixslow_mix_k3_qn3"])
def test_quasi_newton_q2_q3_first_cycle(name):
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"])
    assert fit.opt.q == g.m["q"]
    # em_alg.c:69-72: q-1 secant-collecting double steps, then the first accelerated cycle
    for _ in range(1, fit.opt.q):
        fit.lib.mc_em_2_steps(fit.mp, fit.dat, fit.opt)
        fit.mod.pindex = fit.mod.findex
    trace = g.f64("accel_trace.f64").reshape(-1, 8)
    stop = fit.accelerated_em_step()
    m = fit.mod
    assert not stop and m.fatal == 0
    assert abs(m.last_emll - trace[0, 0]) <= 1e-8
    assert abs(m.last_ll - trace[0, 2]) <= 1e-6 * max(1.0, abs(trace[0, 2]))
    assert m.last_accepted == trace[0, 3]
    assert m.n_iter == trace[0, 4] and m.pindex == trace[0, 6]
    np.testing.assert_allclose(fit.get_p(m.pindex), g.p("cycle1"), rtol=1e-6, atol=1e-10)
    fit.close()


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "multi_admix_k4_s1", "multi_admix_k4_s2", "multi_admix_k3_qn1",
                                  "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3", "multi_admix_k4_tinybound",
                                  "mixslow_mix_k3_s3", "mixslow_mix_k3_qn1"])
def test_whole_accelerated_run_on_the_reference_path(name):
    """The whole -s run, cycle by cycle with the calls of accelerated_em_step (accel_em.c:35-114), against the reference's
    recorded trace (emll, step, ll, accept, n_iter per cycle) and its final iterate.  The accept test ll > emll is a
    last-bit tie whenever the step was clamped to -1 (the extrapolated point then IS the second EM iterate); only on such
    cycles (|ll - emll| <= 1e-9 |emll| in the reference's own record) the recorded flag is followed instead of our own
    comparison, everywhere else our own decision must equal the reference's.  With the path pinned like this the run ends
    at the reference's iteration with north_star's tolerances: logL 1e-8 absolute, Q/P 1e-6 relative."""
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"], abs_error=g.m["abs_error"])
    trace = g.f64("accel_trace.f64").reshape(-1, 8)
    m = fit.mod
    lib = fit.lib
    for c in range(len(trace) + 1):
        lib.mc_em_2_steps(fit.mp, fit.dat, fit.opt)
        assert m.fatal == 0
        if m.stopped:
            break
        assert c < len(trace), "the reference had stopped by now"
        emll = lib.mc_log_likelihood(*fit._a(), m.findex)
        assert abs(emll - trace[c, 0]) <= 1e-8, (c, emll, trace[c, 0])
        s = lib.mc_step_size(*fit._a())
        valid = not (np.isnan(s) or np.isinf(s))
        assert valid == bool(trace[c, 7]), c
        accept = False
        if valid:
            assert abs(s - trace[c, 1]) <= 1e-6 * abs(trace[c, 1]), (c, s, trace[c, 1])
            ll = lib.mc_accelerated_update(*fit._a(), s)
            assert abs(ll - trace[c, 2]) <= 1e-7 * max(1.0, abs(trace[c, 2]) * 1e-3), (c, ll, trace[c, 2])
            tie = abs(trace[c, 2] - trace[c, 0]) <= 1e-9 * abs(trace[c, 0])
            accept = ll > emll
            if tie:
                accept = bool(trace[c, 3])
            else:
                assert accept == bool(trace[c, 3]), (c, ll, emll, trace[c])
        if accept:
            m.pindex = m.tindex
            m.accel_step = 1
        else:
            m.pindex = m.findex
        assert m.n_iter == trace[c, 4] and m.pindex == trace[c, 6], c
        assert abs(m.logL - trace[c, 5]) <= 1e-8, c
    assert m.n_iter == g.m["accel_run_n_iter"] and m.converged == g.m["accel_run_converged"]
    assert m.pindex == g.m["accel_run_pindex"]
    assert abs(m.logL - g.m["accel_run_logL"]) <= 1e-8, (m.logL, g.m["accel_run_logL"])
    np.testing.assert_allclose(fit.get_q(m.pindex), g.q("accelrun"), rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(fit.get_p(m.pindex), g.p("accelrun"), rtol=1e-6, atol=1e-10)
    np.testing.assert_allclose(fit.expected_counts(), g.sik("accelrun"), rtol=1e-6, atol=1e-9)
    fit.close()


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "tetra_admix_k3", "missing_admix_k3"])
def test_accelerated_run_to_convergence(name):
    """Whole SQUAREM-3 fit.  The accept test ll > emll can flip on near-ties under re-associated sums
    (SURVEY.md section 7), so the converged fit is compared within the convergence tolerance."""
    g = Golden(name)
    fit = make_fit(g, accel=3)
    fit.em()
    m = fit.mod
    assert m.fatal == 0 and m.converged == 1
    # "converged" = |delta logL| <= 1e-4 per step on a slowly converging tail: the reference's own plain-EM and
    # SQUAREM fits of this data stop 8e-3 apart (-50284.1359 vs -50284.1278), so the stopping point is only
    # defined to that order
    assert abs(m.logL - g.m["accel_run_logL"]) <= 5e-2, (m.logL, g.m["accel_run_logL"], m.n_iter, g.m["accel_run_n_iter"])
    # tie-driven accept flips change the path, not the destination: iteration counts agree loosely
    assert abs(m.n_iter - g.m["accel_run_n_iter"]) <= max(8, g.m["accel_run_n_iter"] // 4)
    fit.close()


@pytest.mark.parametrize("name", ["c1_admix_k3", "missing_admix_k3", "tetra_admix_k3"])
def test_initialize_model_same_seed(name):
    g = Golden(name)
    fit = host.Fit(g.ua, g.geno, g.K, admixture=1, verbosity=1)
    rng = fit.initialize(g.m["seed"])
    assert fit.lib.mc_rand(rng) == g.m["rand_after_init"]
    np.testing.assert_allclose(fit.get_q(0), g.q("q0"), rtol=1e-15, atol=0)
    np.testing.assert_allclose(fit.get_p(0), g.p("p0"), rtol=1e-15, atol=0)
    assert fit.mod.logL == -np.inf and fit.mod.n_iter == 0
    fit.close()


def test_max_iter_runs_n_plus_one_steps():
    """-T n executes n+1 iterations (n_iter > max_iter, em_alg.c:150)."""
    g = Golden("multi_admix_k4")
    fit = make_fit(g, max_iter=7)
    fit.em()
    assert fit.mod.n_iter == 8 and fit.mod.iter_stop == 1 and fit.mod.converged == 0
    assert abs(fit.mod.logL - g.f64("em_ll.f64")[7]) <= 1e-8
    fit.close()


def test_em_e_step_returns_post_step_loglik():
    g = Golden("multi_admix_k4")
    fit = make_fit(g)
    ll = fit.em_e_step()
    assert abs(ll - g.f64("em_ll.f64")[1]) <= 1e-8          # logL entering step 2 = logL after step 1
    np.testing.assert_allclose(fit.expected_counts().sum(), g.I * g.L * g.ploidy, rtol=1e-12)
    fit.close()


@pytest.mark.parametrize("name", ["multi_admix_k4", "missing_admix_k3"])
def test_fit_units_with_jump_ahead_vs_serial_reference(name):
    """mc_fit_unit: unit u starts from the serial program's stream position (jump-ahead), is initialised and fitted on
    the GPU; per-unit results and the replayed bookkeeping against the reference's maximize_likelihood()."""
    from multiclust_amd import shard
    g = Golden(name)
    ref = g.f64("multi_init.f64").reshape(-1, 4)
    fit = host.Fit(g.ua, g.geno, g.K, admixture=1, accel_scheme=g.m["accel_scheme"], verbosity=1)
    results = []
    for u in reversed(range(g.m["mi_units"])):          # any order: units are independent
        r = fit.fit_unit(g.m["seed"], u)
        assert r.fatal == 0 and r.converged == 1
        # same starting point as the reference; the SQUAREM path may differ through accept ties (see above)
        # SQUAREM stops where one cycle gains < 1e-4; on a slowly converging ridge two paths that differ through
        # accept ties (see above) stop at visibly different points (observed 0.11 on unit 1 of multi_admix_k4)
        assert abs(r.logL - ref[u, 0]) <= 0.25, (u, r.logL, ref[u, 0])
        results.append(r)
    results.sort(key=lambda r: r.unit)
    s = shard.replay(results, fit.opt, fit.no_parameters(), g.I)
    assert s.n_init == g.m["mi_n_init"] and s.ever_converged == 1
    assert abs(s.max_logL - g.m["mi_max_logL"]) <= 0.25
    assert fit.no_parameters() == g.m["no_parameters"]
    fit.close()


@pytest.mark.parametrize("name", ["multi_admix_k4", "missing_admix_k3", "tetra_admix_k3", "multi_admix_c_k3", "multi_mix_k3",
                                  "missing_mix_k2"])
def test_batched_em_equals_step_by_step(name, monkeypatch):
    """mc_em's batched loop (stopping rule on the device, mchip_em_run) against the step-by-step loop: same stopping
    iteration, bitwise the same log likelihood and parameters."""
    g = Golden(name)
    res = []
    for nobatch in (False, True):
        if nobatch:
            monkeypatch.setenv("MC_NO_BATCH", "1")
        fit = make_fit(g)
        fit.em()
        m = fit.mod
        res.append((m.n_iter, m.converged, m.logL, fit.get_q(m.pindex), fit.get_p(m.pindex), fit.expected_counts()))
        fit.close()
    a, b = res
    assert a[0] == b[0] and a[1] == b[1] and a[2] == b[2]
    assert np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])


def test_batched_em_iteration_cap_and_decrease_detection():
    g = Golden("multi_admix_k4")
    fit = make_fit(g, max_iter=40)                     # cap inside the second batch of 32
    fit.em()
    assert fit.mod.n_iter == 41 and fit.mod.iter_stop == 1 and fit.mod.converged == 0
    fit.close()
    fit = make_fit(g)
    fit.mod.logL = 0.0                                 # a "previous" log likelihood above anything reachable
    fit.lib.mc_em(*fit._a())
    assert fit.mod.fatal == 2 and fit.mod.n_iter == 1  # the reference would exit(0) here (em_alg.c:115-120)
    fit.close()


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "multi_admix_k4_s1", "multi_admix_k4_s2",
                                  "multi_admix_k3_qn1", "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3",
                                  "multi_admix_k4_tinybound"] + MIX_ACCEL)
def test_batched_accelerated_run_equals_cycle_by_cycle(name, monkeypatch):
    """mc_em with an acceleration scheme runs its cycles in device-side batches (mchip_accel_run: stop rule, step size and
    accept test decided by one-thread kernels, one captured graph per cycle).  Same arithmetic as the cycle-by-cycle host
    loop: iteration count, flags, log likelihood and the reported iterate are identical bit for bit."""
    g = Golden(name)
    out = []
    for batched in (True, False):
        if batched:
            monkeypatch.delenv("MC_NO_BATCH", raising=False)
        else:
            monkeypatch.setenv("MC_NO_BATCH", "1")
        fit = make_fit(g, accel=g.m["accel_scheme"])
        fit.em()
        m = fit.mod
        assert m.fatal == 0
        out.append((m.n_iter, m.converged, m.iter_stop, m.logL, fit.get_q(m.pindex), fit.get_p(m.pindex), fit.expected_counts()))
        fit.close()
    a, b = out
    assert a[:4] == b[:4], (a[:4], b[:4])
    assert np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    assert np.array_equal(a[6], b[6])


def test_batched_accelerated_run_honours_the_iteration_cap():
    """-T n with -s 3: stop() counts EM steps, so the cap can fire inside em_2_steps; the reported iterate is then the
    cycle's starting point (accel_em.c:44-45)."""
    g = Golden("multi_admix_k4")
    res = []
    for env in ({}, {"MC_NO_BATCH": "1"}):
        import os
        old = os.environ.pop("MC_NO_BATCH", None)
        os.environ.update(env)
        try:
            fit = make_fit(g, accel=3, max_iter=6)
            fit.em()
            m = fit.mod
            res.append((m.n_iter, m.iter_stop, m.converged, m.logL, fit.get_p(m.pindex)))
            fit.close()
        finally:
            os.environ.pop("MC_NO_BATCH", None)
            if old is not None:
                os.environ["MC_NO_BATCH"] = old
    assert res[0][:4] == res[1][:4] and res[0][0] == 7 and res[0][1] == 1
    assert np.array_equal(res[0][4], res[1][4])
