"""diagnostic (GPU): time of SQUAREM cycles at config 3 without the profiling hooks (graph replay, side stream)"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench, torch
from multiclust_amd import hip, host
w = bench.WORKLOADS["c3"]
ua, geno = bench.gen_dataset(w["I"], w["L"], w["K"], w["ploidy"], w["maxal"], 20250117 + 3, torch.device("cuda", 0))
fit = host.Fit(ua, geno, w["K"], admixture=1, accel_scheme=3, verbosity=1, abs_error=1e-300)
fit.initialize(bench.SEED)
lib, ctx = hip.load(), C.c_void_p(fit.mod.dev)
for rep in range(3):
    st = hip.RunState(logL=fit.mod.logL, abs_error=1e-300, n_iter=fit.mod.n_iter)
    lib.mchip_synchronize(ctx)
    t0 = time.perf_counter()
    rc = lib.mchip_accel_run(ctx, fit.mod.pindex, 3, 20, C.byref(st))
    dt = time.perf_counter() - t0
    assert rc == 0 and not st.fatal and not st.stopped
    fit.mod.n_iter, fit.mod.logL = st.n_iter, st.logL
    print("20 cycles: %.3f ms per cycle, logL %.6f" % (dt * 1e3 / 20, st.logL), flush=True)
