"""diagnostic (GPU): where one bootstrap replicate at config-5 size spends its time"""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
from multiclust_amd import host
import torch
w = bench.WORKLOADS["c5"]
ua, geno = bench.gen_dataset(w["I"], w["L"], w["K"], w["ploidy"], w["maxal"], 20250117 + 5, torch.device("cuda", 0))
lib = host.load()
K0 = 7
fit = host.Fit(ua, geno, K0, admixture=1, accel_scheme=0, verbosity=1, abs_error=1e-300, max_iter=20)
fit.fit_unit(bench.SEED, 0)
q, p = np.ascontiguousarray(fit.get_q(0)), np.ascontiguousarray(fit.get_p(0))
opt, dat = fit.opt, fit.dat
rng = host.McRng(); lib.mc_srand(C.byref(rng), 5)
MP = C.POINTER(host.McModel)
for rep in range(int(os.environ.get("REPS", "3"))):
    sim = host.McSimulation()
    t0 = time.perf_counter()
    lib.mc_simulation_begin(C.byref(sim), C.byref(opt), C.byref(dat), K0, q.ctypes.data, p.ctypes.data, C.byref(rng))
    t1 = time.perf_counter()
    for K in (7, 8):
        mp = MP()
        a = time.perf_counter()
        rc = lib.mc_model_create_simulated(C.byref(mp), C.byref(opt), C.byref(dat), K, 0, C.byref(sim)); assert rc == 0
        b = time.perf_counter()
        lib.mc_reset_model_state(mp)
        rc = lib.mc_initialize_model(C.byref(opt), C.byref(dat), mp, C.byref(rng)); assert rc == 0
        c = time.perf_counter()
        lib.mc_em(C.byref(opt), C.byref(dat), mp)
        d = time.perf_counter()
        lib.mc_model_free(mp)
        e = time.perf_counter()
        print("rep %d K=%d: create+simulate %.1f ms, initialise %.1f ms, em(%d it) %.1f ms, free %.1f ms" % (rep, K, (b-a)*1e3, (c-b)*1e3, 21, (d-c)*1e3, (e-d)*1e3), flush=True)
    print("  simulation_begin (host jump) %.2f ms" % ((t1-t0)*1e3))
