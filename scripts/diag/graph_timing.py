"""diagnostic (GPU): batched steps of a bench workload WITHOUT the profiling hooks (captured graph, side stream), as
`secondary.c2` / c4 time them.  usage: graph_timing.py [workload [steps [batches]]]; prints rate per batch and the log likelihood
(which must not depend on MCHIP_NO_FORK / MCHIP_NO_GRAPH)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from multiclust_amd import hip, host
name = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
batches = int(sys.argv[3]) if len(sys.argv) > 3 else 5
w = bench.WORKLOADS[name]
ua, geno = bench.gen_dataset(w["I"], w["L"], w["K"], w["ploidy"], w["maxal"], 20250117 + w["dseed"], torch.device("cuda", 0))
accel = w.get("accel", 0)
fit = host.Fit(ua, geno, w["K"], admixture=1, accel_scheme=accel, verbosity=1, abs_error=1e-300)
fit.initialize(bench.SEED)
lib, ctx = hip.load(), C.c_void_p(fit.mod.dev)
per = 2 if accel else 1
rates = []
for rep in range(batches):
    st = hip.RunState(logL=fit.mod.logL, abs_error=1e-300, n_iter=fit.mod.n_iter)
    lib.mchip_synchronize(ctx)
    t0 = time.perf_counter()
    rc = lib.mchip_accel_run(ctx, fit.mod.pindex, accel, steps, C.byref(st)) if accel else lib.mchip_em_run(ctx, 0, steps, C.byref(st))
    dt = time.perf_counter() - t0
    assert rc == 0 and not st.fatal and not st.stopped, (rc, st.fatal, st.stopped)
    fit.mod.n_iter, fit.mod.logL = st.n_iter, st.logL
    rates.append(per * steps / dt)
print("%s %-16s %s it/s  logL %.10f" % (name, os.environ.get("TAG", "-"), " ".join("%.0f" % r for r in rates), fit.mod.logL), flush=True)
