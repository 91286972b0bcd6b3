"""diagnostic (GPU): every cycle of the reference's accelerated run, restarted from the reference's own iterate"""
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
    nq = g.I * g.K if g.indiv_q else g.K
    st = g.f64("accel_states.f64").reshape(-1, 1 + nq + g.K * g.T)
    states = {int(r[0]): r for r in st}
    m = fit.mod
    d = dict(emll=0.0, s=0.0, ll=0.0, ll_far=0.0, q=0.0, p=0.0, logL=0.0)
    flips = ties = n = nxt = 0
    for c, r in sorted(states.items()):
        if c >= len(trace):
            continue
        fit.reset()
        q = r[1:1 + nq].reshape(g.I, g.K) if g.indiv_q else r[1:1 + nq]
        fit.set_params(q, r[1 + nq:].reshape(g.K, g.T))
        m.n_iter = int(trace[c - 1, 4]) if c else 0
        stop = fit.accelerated_em_step()
        assert not stop and not m.fatal
        n += 1
        d["emll"] = max(d["emll"], abs(m.last_emll - trace[c, 0]))
        d["logL"] = max(d["logL"], abs(m.logL - trace[c, 5]))
        if trace[c, 7]:
            d["s"] = max(d["s"], abs(m.last_step - trace[c, 1]) / abs(trace[c, 1]))
            comp = trace[c, 2] > trace[c, 0] - 1.0
            d["ll" if comp else "ll_far"] = max(d["ll" if comp else "ll_far"], abs(m.last_ll - trace[c, 2]) / (1.0 if comp else abs(trace[c, 2])))
            tie = abs(trace[c, 2] - trace[c, 0]) <= 1e-9 * abs(trace[c, 0])
            ties += tie
            if not tie and m.last_accepted != trace[c, 3]:
                flips += 1
            if (c + 1) in states and m.last_accepted == trace[c, 3]:
                r2 = states[c + 1]
                q2 = r2[1:1 + nq].reshape(g.I, g.K) if g.indiv_q else r2[1:1 + nq]
                p2 = r2[1 + nq:].reshape(g.K, g.T)
                gq, gp = fit.get_q(m.pindex), fit.get_p(m.pindex)
                bq, bp = q2 > 1e-6, p2 > 1e-6
                d["q"] = max(d["q"], np.max(np.abs(gq - q2)[bq] / q2[bq]))
                d["p"] = max(d["p"], np.max(np.abs(gp - p2)[bp] / p2[bp]))
                nxt += 1
    print("%-28s cycles %3d next-states %3d ties %d flips %d | emll %.1e logL %.1e s(rel) %.1e ll %.1e ll_far(rel) %.1e relQ %.1e relP %.1e" % (
        name, n, nxt, ties, flips, d["emll"], d["logL"], d["s"], d["ll"], d["ll_far"], d["q"], d["p"]), flush=True)
    fit.close()
