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
    if g.m.get("adjust_step"):
        kw.setdefault("adjust_step", g.m["adjust_step"])      # -g: step back-tracking (accel_em.c:67-82)
    fit = host.Fit(g.ua, g.geno, g.K, admixture=g.m["admixture"], eta_constrained=g.m["eta_constrained"],
                   do_projection=g.m["do_projection"], accel_scheme=accel, verbosity=1, **kw)
    assert fit.opt.lower_bound == g.lower_bound
    fit.set_params(g.q("q0"), g.p("p0"))
    return fit


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "tetra_admix_k3", "missing_admix_k3",
                                  "multi_admix_k1", "multi_admix_c_k3", "multi_mix_k3", "missing_mix_k2",
                                  "allmiss_admix_k2", "allmiss_mix_k2", "mono_admix_k3", "haploid_admix_k2",
                                  "triploid_admix_k3", "hexaploid_admix_k2", "hexaploid_mix_k2",
                                  "manyallele_admix_k2", "missing_admix_c_k2"])
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


QN_MULTI = ["multi_admix_k3_qn2", "multi_admix_k3_qn3", "mixslow_mix_k3_qn2", "mixslow_mix_k3_qn3"]
BACKTRACK = ["multi_admix_k4_g3", "multi_admix_k4_s1_g2"]


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "multi_admix_k4_s1", "multi_admix_k4_s2",
                                  "multi_admix_k3_qn1", "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3",
                                  "multi_admix_k4_tinybound", "allmiss_admix_k2", "mono_admix_k3", "haploid_admix_k2",
                                  "triploid_admix_k3", "hexaploid_admix_k2", "manyallele_admix_k2", "missing_admix_c_k2"] + MIX_ACCEL + QN_MULTI + BACKTRACK)
def test_accelerated_cycles_trace(name):
    """First cycles of SQUAREM / QN with q = 1, 2, 3 secant pairs / SQUAREM with step back-tracking (-g), one after the other from
    the fixture's initial parameters, against the reference's recorded emll, step size, ll, accept, ring index.  With q > 1 the
    cycles use the pairs the previous cycles left behind, in the slots em_alg.c:1171 rotates through."""
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"])
    assert fit.opt.q == g.m["q"] and fit.opt.adjust_step == g.m.get("adjust_step", 0)
    # em_alg.c:69-72: q - 1 secant-collecting double steps before the first accelerated cycle
    for _ in range(1, fit.opt.q):
        fit.lib.mc_em_2_steps(fit.mp, fit.dat, fit.opt)
        fit.mod.pindex = fit.mod.findex
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
            if trace[c, 2] > trace[c, 0] - 1.0:
                assert abs(m.last_ll - trace[c, 2]) <= 2e-7, c
            else:       # rejected by a mile: the value is dominated by entries the projection clamped to the lower bound
                assert m.last_ll < m.last_emll and abs(m.last_ll - trace[c, 2]) <= 2e-2 * abs(trace[c, 2]), c
            if not tie:
                assert m.last_accepted == trace[c, 3], c
            elif m.last_accepted != trace[c, 3]:
                ring_in_sync = False
        assert m.n_iter == trace[c, 4]
        if ring_in_sync:
            assert m.pindex == trace[c, 6], c
        # (cycles run one after the other here: from the third on the two runs stand on iterates a few ulp apart, and the
        # extrapolation amplifies that; the strict comparison of every cycle is the from-the-reference-state test below)
        assert abs(m.logL - trace[c, 5]) <= (1e-8 if c < 2 else 1e-7), c
        if c == 0:
            np.testing.assert_allclose(fit.get_q(m.pindex), g.q("cycle1"), rtol=1e-7, atol=1e-12)
            np.testing.assert_allclose(fit.get_p(m.pindex), g.p("cycle1"), rtol=1e-7, atol=1e-12)
        if not ring_in_sync:
            break               # a last-bit tie went the other way: the runs are on different iterates from here on
    fit.close()


def set_secants(fit, delta_index, parts):
    """the state a quasi-Newton run with q > 1 carries from cycle to cycle: the last q secant pairs (mchip_set_secant) and
    model::delta_index"""
    import ctypes as C
    from multiclust_amd import hip
    lib, ctx = hip.load(), C.c_void_p(fit.mod.dev)
    for j, (uq, up, vq, vp) in enumerate(parts):
        for which, (qq, pp) in enumerate(((uq, up), (vq, vp))):
            qq, pp = np.ascontiguousarray(qq), np.ascontiguousarray(pp)
            assert lib.mchip_set_secant(ctx, which, j, pp.ctypes.data, qq.ctypes.data) == 0
    fit.mod.delta_index = delta_index


def get_secants(fit, n):
    import ctypes as C
    from multiclust_amd import hip
    lib, ctx = hip.load(), C.c_void_p(fit.mod.dev)
    out = []
    for j in range(n):
        four = []
        for which in range(2):
            qq, pp = np.empty(fit.I * fit.K if fit.indiv_q else fit.K), np.empty((fit.K, fit.T))
            assert lib.mchip_get_secant(ctx, which, j, pp.ctypes.data, qq.ctypes.data) == 0
            four += [qq, pp]
        out.append(tuple(four))
    return out


@pytest.mark.parametrize("name", ["multi_admix_k3_qn3", "mixslow_mix_k3_qn2", "multi_admix_c_k3"])
def test_secant_buffers_are_readable_and_writable_state(name):
    """mchip_set_secant / mchip_get_secant: what is written comes back bit for bit in the boundary's flat order (P part [K][T], Q
    part [I][K] or [K]); a pair collected by mc_em_2_steps reads back as x[to] - x[from]; indices out of range are refused."""
    import ctypes as C
    from multiclust_amd import hip
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"])
    nsec = fit.opt.q
    rng = np.random.default_rng(3)
    nq = g.I * g.K if g.indiv_q else g.K
    parts = [(rng.standard_normal(nq), rng.standard_normal((g.K, g.T)), rng.standard_normal(nq), rng.standard_normal((g.K, g.T)))
             for _ in range(nsec)]
    set_secants(fit, 0, parts)
    for got, want in zip(get_secants(fit, nsec), parts):
        for a, b in zip(got, want):
            assert np.array_equal(a, b.reshape(a.shape))
    lib, ctx = hip.load(), C.c_void_p(fit.mod.dev)
    buf_p, buf_q = np.zeros((g.K, g.T)), np.zeros(nq)
    assert lib.mchip_set_secant(ctx, 0, nsec, buf_p.ctypes.data, buf_q.ctypes.data) == 1        # MCHIP_ERR_INVALID
    assert lib.mchip_get_secant(ctx, 2, 0, buf_p.ctypes.data, buf_q.ctypes.data) == 1
    assert lib.mchip_set_secant(ctx, 0, 0, None, buf_q.ctypes.data) == 1
    # one em_2_steps: u = x1 - x0, v = x2 - x1 of the ring (em_alg.c:1104-1161), in slot delta_index = 0
    q0, p0 = fit.get_q(0), fit.get_p(0)
    fit.lib.mc_em_2_steps(fit.mp, fit.dat, fit.opt)
    x1q, x1p, x2q, x2p = fit.get_q(1), fit.get_p(1), fit.get_q(2), fit.get_p(2)
    uq, up, vq, vp = get_secants(fit, nsec)[0]
    assert np.array_equal(up, x1p - p0) and np.array_equal(vp, x2p - x1p)
    assert np.array_equal(uq, (x1q - q0).ravel()) and np.array_equal(vq, (x2q - x1q).ravel())
    assert fit.mod.delta_index == 1 % nsec
    fit.close()


@pytest.mark.parametrize("name", ["c1_admix_k3", "multi_admix_k4", "multi_admix_k4_s1", "multi_admix_k4_s2", "multi_admix_k3_qn1",
                                  "tetra_admix_k3", "missing_admix_k3", "multi_admix_c_k3", "multi_admix_k4_tinybound",
                                  "mixslow_mix_k3_s3", "mixslow_mix_k3_qn1"] + QN_MULTI + BACKTRACK)
def test_every_cycle_of_the_accelerated_run_from_the_reference_state(name):
    """The WHOLE -s run of the reference, cycle by cycle.  SQUAREM's path is not reproducible to a tolerance: the step
    -sqrt(u'u / (v-u)'(v-u)) amplifies last-bit differences until two runs that differ only in summation order are whole
    units of log likelihood apart mid-run (the CPU oracle in fused order shows the same against the reference,
    tests/test_oracle_golden.py::test_squarem_path_depends_on_summation_order).  So every cycle is restarted from the
    reference's own iterate (recorded by the harness for each cycle, every sixth pair for config 1) and ONE
    accelerated_em_step (accel_em.c:35-114) is compared with what the reference did from there: emll and the E-step log
    likelihood to 1e-8 absolute, the step to 1e-8 relative, the log likelihood of the extrapolated point to 2e-7 when it is
    competitive (a point the reference rejected by more than 1 is only required to be rejected here too: its value is
    dominated by entries clamped to the lower bound), the accept decision (except last-bit ties: step clamped to -1, the
    extrapolated point IS the EM iterate), and the resulting iterate against the reference's next state to 1e-6 relative.
    Quasi-Newton with q = 2, 3 (qn_accelerated_update, accel_em.c:262-419) restarts with the secant pairs and delta_index the
    reference held at that point (accel_secants.f64), and what the cycle leaves in the secant slots is compared as well; the
    -g fixtures run the back-tracking loop (accel_em.c:67-82) wherever the reference's run did."""
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"], abs_error=g.m["abs_error"])
    trace = g.f64("accel_trace.f64").reshape(-1, 8)
    states, secants = g.cycle_states(), g.cycle_secants()
    assert (g.m["q"] > 1) == bool(secants)
    m = fit.mod
    checked = compared_next = ties = 0
    for c in sorted(states):
        if c >= len(trace):
            continue                                   # the state the final, stopping em_2_steps started from
        fit.reset()
        fit.set_params(*states[c])
        if secants:
            set_secants(fit, *secants[c])
        m.n_iter = int(trace[c - 1, 4]) if c else 2 * (g.m["q"] - 1)
        assert not fit.accelerated_em_step() and m.fatal == 0
        emll, s, ll, accepted = trace[c, 0], trace[c, 1], trace[c, 2], trace[c, 3]
        assert abs(m.last_emll - emll) <= 1e-8, (c, m.last_emll, emll)
        assert abs(m.logL - trace[c, 5]) <= 1e-8 and m.n_iter == trace[c, 4], c
        if not trace[c, 7]:
            continue
        assert abs(m.last_step - s) <= 1e-8 * abs(s), (c, m.last_step, s)
        tie = abs(ll - emll) <= 1e-9 * abs(emll)
        if ll > emll - 1.0:
            assert abs(m.last_ll - ll) <= 2e-7, (c, m.last_ll, ll)
        else:
            assert m.last_ll < m.last_emll and not m.last_accepted, c
        if not tie:
            assert m.last_accepted == accepted, (c, m.last_ll, m.last_emll, ll, emll)
        ties += tie
        checked += 1
        if c + 1 in states and m.last_accepted == accepted:
            q2, p2 = states[c + 1]
            np.testing.assert_allclose(fit.get_q(m.pindex), q2, rtol=1e-6, atol=1e-12, err_msg="cycle %d" % c)
            np.testing.assert_allclose(fit.get_p(m.pindex), p2, rtol=1e-6, atol=1e-12, err_msg="cycle %d" % c)
            compared_next += 1
        if secants and c + 1 in secants:
            di2, parts2 = secants[c + 1]
            assert m.delta_index == di2, c
            for got, want in zip(get_secants(fit, g.m["q"]), parts2):
                for a, b in zip(got, want):
                    np.testing.assert_allclose(a, b, rtol=0, atol=1e-11, err_msg="secants after cycle %d" % c)
    # every recorded cycle was checked; its outcome was compared with the next recorded state unless a tie went the other way
    assert checked >= len([c for c in states if c < len(trace) and trace[c, 7]])
    assert compared_next >= len([c for c in states if c + 1 in states and c < len(trace) and trace[c, 7]]) - ties
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
    # how far the two converged fits are from each other in the parameters: both stopped where a cycle gains less than 1e-4 in
    # log likelihood, on a likelihood surface that is flat to that order along its ridge.  The CPU oracle with re-associated sums
    # ends 5e-5 to 5e-3 from the reference on these data sets (tests/test_oracle_golden.py::
    # test_squarem_path_depends_on_summation_order); the same bound, stated, holds here
    dq = np.abs(fit.get_q(m.pindex) - g.q("accelrun")).max()
    dp = np.abs(fit.get_p(m.pindex) - g.p("accelrun")).max()
    assert dq <= 2e-2 and dp <= 2e-2, (dq, dp)
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


@pytest.mark.parametrize("K,maxal,missing,scheme", [(2, 4, 0.0, 3), (5, 2, 0.0, 3), (8, 4, 0.0, 3), (8, 3, 0.04, 1), (12, 4, 0.02, 2),
                                                     (6, 2, 0.0, 3), (13, 4, 0.0, 3), (8, 4, 0.0, 4)])
def test_dual_individual_pass_equals_the_two_passes(K, maxal, missing, scheme, monkeypatch):
    """A batched accelerated cycle of diploid data takes log L of the second EM iterate and the E step of the extrapolated point
    in one pass over the genotypes (k_individual_sparse<.., DUAL>; K <= 12, not where the all-biallelic kernels apply: the
    (6, 2) and K = 13 cases run the two passes either way).  Same arithmetic lane for lane: whole fits with it and with
    MCHIP_NO_DUAL=1 (the two separate passes) end on the same bits, and both equal the cycle-by-cycle host loop."""
    from multiclust_amd import host
    from synth import make_dataset
    ua, geno = make_dataset(300, 700, K, ploidy=2, max_alleles=maxal, seed=77 + K, missing=missing)
    out = []
    for env in ({}, {"MCHIP_NO_DUAL": "1"}, {"MC_NO_BATCH": "1"}):
        for k in ("MCHIP_NO_DUAL", "MC_NO_BATCH"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fit = host.Fit(ua, geno, K, admixture=1, accel_scheme=scheme, verbosity=1, max_iter=24)
        fit.initialize(4242)
        fit.em()
        m = fit.mod
        assert m.fatal == 0
        out.append((m.n_iter, m.converged, m.iter_stop, m.logL, fit.get_q(m.pindex), fit.get_p(m.pindex), fit.expected_counts()))
        fit.close()
    for other in out[1:]:
        assert out[0][:4] == other[:4], (out[0][:4], other[:4])
        assert np.array_equal(out[0][4], other[4]) and np.array_equal(out[0][5], other[5]) and np.array_equal(out[0][6], other[6])


@pytest.mark.parametrize("K,ploidy,maxal,missing,scheme", [(2, 2, 4, 0.0, 0), (5, 2, 2, 0.0, 0), (8, 2, 4, 0.03, 3), (7, 4, 4, 0.0, 0),
                                                            (17, 2, 3, 0.0, 0), (30, 2, 4, 0.0, 1), (8, 1, 6, 0.0, 0)])
def test_one_launch_finaliser_equals_the_two_launches(K, ploidy, maxal, missing, scheme, monkeypatch):
    """An M step with individual mixing proportions finalises Q and P in ONE launch (k_finalize_qp: the blocks of k_finalize_q and
    of k_finalize_p_tile in one grid, mchip_finalize.h).  Same bodies: whole fits with it and with MCHIP_NO_FUSED_FINALIZE=1 (the
    two launches) end on the same bits -- K on both sides of the 16 at which k_finalize_q's exchange changes form, a K with the
    lane-split passes, plain EM and batched accelerated cycles, haploid to tetraploid."""
    from multiclust_amd import host
    from synth import make_dataset
    ua, geno = make_dataset(260, 500, K, ploidy=ploidy, max_alleles=maxal, seed=31 + K, missing=missing)
    out = []
    for env in ({}, {"MCHIP_NO_FUSED_FINALIZE": "1"}):
        monkeypatch.delenv("MCHIP_NO_FUSED_FINALIZE", raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        fit = host.Fit(ua, geno, K, admixture=1, accel_scheme=scheme, verbosity=1, max_iter=20)
        fit.initialize(99)
        fit.em()
        m = fit.mod
        assert m.fatal == 0
        out.append((m.n_iter, m.converged, m.iter_stop, m.logL, fit.get_q(m.pindex), fit.get_p(m.pindex), fit.expected_counts()))
        fit.close()
    assert out[0][:4] == out[1][:4], (out[0][:4], out[1][:4])
    assert np.array_equal(out[0][4], out[1][4], equal_nan=True) and np.array_equal(out[0][5], out[1][5]) and np.array_equal(out[0][6], out[1][6])
