"""diagnostic (GPU): deviation of the cycle-by-cycle SQUAREM/QN run from the reference's recorded trace"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import Golden
from test_gpu_host_driver import make_fit
for name in sys.argv[1:]:
    g = Golden(name)
    fit = make_fit(g, accel=g.m["accel_scheme"], abs_error=g.m["abs_error"])
    trace = g.f64("accel_trace.f64").reshape(-1, 8)
    m, lib = fit.mod, fit.lib
    dev_emll = dev_ll = dev_s = 0.0
    flips = ties = 0
    for c in range(len(trace) + 1):
        lib.mc_em_2_steps(fit.mp, fit.dat, fit.opt)
        if m.stopped or c >= len(trace):
            break
        emll = lib.mc_log_likelihood(*fit._a(), m.findex)
        dev_emll = max(dev_emll, abs(emll - trace[c, 0]))
        s = lib.mc_step_size(*fit._a())
        accept = False
        if not (np.isnan(s) or np.isinf(s)):
            dev_s = max(dev_s, abs(s - trace[c, 1]) / abs(trace[c, 1]))
            ll = lib.mc_accelerated_update(*fit._a(), s)
            tie = abs(trace[c, 2] - trace[c, 0]) <= 1e-9 * abs(trace[c, 0])
            competitive = trace[c, 2] > trace[c, 0] - 1.0
            if competitive:
                dev_ll = max(dev_ll, abs(ll - trace[c, 2]))
            accept = ll > emll
            if tie:
                ties += 1
                accept = bool(trace[c, 3])
            elif accept != bool(trace[c, 3]):
                flips += 1
                accept = bool(trace[c, 3])
        if accept:
            m.pindex = m.tindex; m.accel_step = 1
        else:
            m.pindex = m.findex
    q, p = fit.get_q(m.pindex), fit.get_p(m.pindex)
    rq, rp = g.q("accelrun"), g.p("accelrun")
    bq, bp = rq > 1e-6, rp > 1e-6
    print("%-28s cycles %3d/%3d n_iter %d/%d stopped %d ties %d flips %d | max dev emll %.2e ll %.2e s %.1e | final dlogL %.2e relQ %.2e relP %.2e absQ %.2e absP %.2e" % (
        name, c, len(trace), m.n_iter, g.m["accel_run_n_iter"], m.stopped, ties, flips, dev_emll, dev_ll, dev_s,
        abs(m.logL - g.m["accel_run_logL"]), np.max(np.abs(q - rq)[bq] / rq[bq]), np.max(np.abs(p - rp)[bp] / rp[bp]),
        np.max(np.abs(q - rq)), np.max(np.abs(p - rp))), flush=True)
    fit.close()
