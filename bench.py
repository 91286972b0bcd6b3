#!/usr/bin/env python3
"""bench.py -- EM iterations/s of the MI355X EM hot path on BASELINE.json's workload.

  python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (default "c3", BASELINE.json configs[2], the configuration north_star's target is quoted on):
10 000 diploid individuals x 100 000 loci, M_l ~ U{2,3,4} alleles, admixture K = 8, SQUAREM-3 (-s 3), one
initialisation per GPU.  A "step" is one pass of the hot path as the reference sequences it for -s 3: one
accelerated_em_step() cycle = 2 EM iterations (E+M, n_iter += 2) + 2 stand-alone log-likelihood passes +
step size + extrapolation/projection (accel_em.c:35-114), driven by the plain-C host side over the C-ABI.
value = EM iterations/s summed over ranks (n_iter increments / wall time), inputs resident in HBM.

Multi-GPU: the path shards by independent units (random initialisations, multiclust.c:516-653): rank r fits
its own initialisation on its own GPU with no data-path collective; one RCCL all-reduce (MAX) picks the best
log likelihood at the end (inside the timed region).  scaling = "weak".

Also reported on the same JSON line: `roofline` for the dominant kernel (HIP events on the library's own
stream) and `cpu_baseline` (the CPU oracle, timed on a bounded sample on rank 0 at N = 1 only; the oracle is
used here only as the baseline, never as the measured path).  `cpu_baseline.parity` is the metric's "logL delta vs ref":
the HIP path run on that same sample from the same parameters for the same iterations, against the oracle's result.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # vector FP64 peak (spec), for the secondary fraction only
FP64_VALU_MEASURED_TF = 70.5 # best v_fma_f64 rate measured on this chip (profiles/r01_fp64_microbench.txt)

WORKLOADS = {
    # name: I, L, ploidy, max alleles, K, accel scheme, description
    "c3": dict(I=10000, L=100000, ploidy=2, maxal=4, K=8, accel=3,
               desc="10000 diploid x 100000 loci, M_l~U{2,3,4}, admixture K=8, SQUAREM-3 (-s 3)"),
    "c2": dict(I=2000, L=20000, ploidy=2, maxal=2, K=5, accel=0,
               desc="2000 diploid x 20000 biallelic loci, admixture K=5, plain EM (-s 0)"),
    "c1": dict(I=100, L=500, ploidy=2, maxal=2, K=3, accel=0,
               desc="100 diploid x 500 biallelic loci, admixture K=3, plain EM"),
    "c5": dict(I=5000, L=50000, ploidy=4, maxal=4, K=8, accel=0,
               desc="5000 tetraploid x 50000 loci, M_l~U{2,3,4}, admixture K=8, plain EM"),
}


def gen_dataset(I, L, K, ploidy, maxal, seed, device):
    """SURVEY.md 8d generator on the GPU (torch is plumbing here): P_kl ~ Dirichlet(0.5), Q_i ~ Dirichlet(0.2),
    z ~ Cat(Q_i) per allele copy, allele ~ Cat(P_z,l).  Returns host arrays (the C-ABI takes host buffers)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    if maxal > 2:
        ua = torch.randint(2, maxal + 1, (L,), generator=g, device=device)
    else:
        ua = torch.full((L,), 2, device=device, dtype=torch.int64)
    M = int(ua.max())
    conc = torch.full((K, L, M), 0.5, device=device)
    P = torch._standard_gamma(conc, generator=g) + 1e-3
    P = P * (torch.arange(M, device=device)[None, None, :] < ua[None, :, None])
    P = P / P.sum(dim=2, keepdim=True)
    cdf = torch.cumsum(P, dim=2).float()
    Qg = torch._standard_gamma(torch.full((I, K), 0.2, device=device), generator=g) + 1e-6
    qcdf = torch.cumsum(Qg / Qg.sum(dim=1, keepdim=True), dim=1).float()
    geno = np.empty((I, L, ploidy), dtype=np.uint8)
    lidx = torch.arange(L, device=device)[None, :, None]
    chunk = max(1, int(2.5e8 // (L * ploidy * max(M, K))))
    for i0 in range(0, I, chunk):
        i1 = min(I, i0 + chunk)
        n = i1 - i0
        u = torch.rand((n, L, ploidy), generator=g, device=device)
        z = (u[..., None] > qcdf[i0:i1, None, None, :K - 1]).sum(dim=3) if K > 1 else torch.zeros((n, L, ploidy), dtype=torch.int64, device=device)
        u2 = torch.rand((n, L, ploidy), generator=g, device=device)
        c = cdf[z, lidx.expand(n, L, ploidy)]                  # (n, L, ploidy, M)
        a = (u2[..., None] > c[..., :M - 1]).sum(dim=3)
        a = torch.minimum(a, (ua[None, :, None] - 1))
        geno[i0:i1] = a.to(torch.uint8).cpu().numpy()
    return ua.to(torch.int32).cpu().numpy(), geno


def algorithmic_bytes(w, T):
    I, L, p, K = w["I"], w["L"], w["ploidy"], w["K"]
    g = I * L * p
    return {
        "iteration": g + 16 * K * T + 16 * I * K,            # B_it
        "column_pass": g + 16 * K * T + 8 * I * K,           # genotype + read P, write N-side sums + read Q
        "individual_pass": g + 8 * K * T + 16 * I * K,       # genotype + read P + read Q, write S-side sums
        "loglik_pass": g + 8 * K * T + 8 * I * K,
    }


def cpu_baseline(w, ua, geno, accel, budget_s=20.0, device=0):
    """The CPU oracle (oracle/mc_oracle.c, fused order, one host core) on a bounded sample of the same
    workload: the first L_s loci of every individual, sized for about `budget_s` seconds."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind as ob
    from synth import random_params
    I, L, p, K = w["I"], w["L"], w["ploidy"], w["K"]
    per_cell_ns = 2.2 * K                                      # measured on the build container's Xeon
    iters = 4 if accel else 3
    passes = iters * (1.6 if accel else 1.0)
    Ls = int(max(8, min(L, budget_s * 1e9 / (per_cell_ns * I * p * passes))))
    ua_s = np.ascontiguousarray(ua[:Ls])
    geno_s = np.ascontiguousarray(geno[:, :Ls, :])
    lb = ob.lib.mco_lower_bound(1e-8, I, p)
    q0, p0 = random_params(I, ua_s, K, seed=11, lower_bound=lb)
    opt = ob.make_options(lower_bound=lb, fused=1, accel_scheme=accel, abs_error=1e-300)
    mod = ob.Model(ob.Data(I, Ls, p, ua_s, geno_s), opt, K)
    mod.q(0)[...] = q0
    mod.p(0)[...] = p0
    t0 = time.perf_counter()
    if accel:
        for _ in range(iters // 2):
            mod.accelerated_em_step()
    else:
        for _ in range(iters):
            mod.em_step()
    dt = time.perf_counter() - t0
    it_s_sample = mod.n_iter / dt
    # the same sample, parameters and iterations through the HIP path: "logL delta vs ref" of BASELINE.json's metric,
    # measured in this run (the oracle is the checker here, as in tests/)
    from multiclust_amd import host
    fit = host.Fit(ua_s, geno_s, K, device=device, admixture=1, accel_scheme=accel, verbosity=1, abs_error=1e-300)
    fit.set_params(q0, p0)
    for _ in range(iters // 2 if accel else iters):
        fit.accelerated_em_step() if accel else fit.em_step()
    gq, gp = fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex)
    oq, op = mod.q(mod.pindex), mod.p(mod.pindex)
    big_q, big_p = oq > 1e-6, op > 1e-6
    parity = {
        "abs_dlogL": abs(fit.mod.logL - mod.logL), "rel_dlogL": abs(fit.mod.logL - mod.logL) / abs(mod.logL),
        "max_rel_dQ": float(np.max(np.abs(gq - oq)[big_q] / oq[big_q])), "max_rel_dP": float(np.max(np.abs(gp - op)[big_p] / op[big_p])),
        "same_n_iter": int(fit.mod.n_iter == mod.n_iter),
        "note": "HIP path vs CPU oracle on the cpu_baseline sample, same parameters and iterations; Q/P entries > 1e-6",
    }
    fit.close()
    return {
        "value": it_s_sample * Ls / L, "unit": "EM iterations/s", "cores": 1, "kind": "port",
        "sample": "first %d of %d loci, all %d individuals, %d EM iterations (%s) in %.1f s on one core; "
                  "scaled by %d/%d (cost is linear in loci)" % (Ls, L, I, mod.n_iter, "SQUAREM-3 cycles" if accel else "plain EM", dt, Ls, L),
        "sample_value": it_s_sample, "parity": parity,
    }


def secondary_run(name, dev, local_rank, steps=300):
    """A second, smaller BASELINE.json configuration measured the same way (plain numbers only), so that one bench line
    carries both single-GPU configurations: configs[1] (c2) next to the headline configs[2] (c3)."""
    from multiclust_amd import hip, host
    w = WORKLOADS[name]
    ua, geno = gen_dataset(w["I"], w["L"], w["K"], w["ploidy"], w["maxal"], 20250117 + 2, dev)
    fit = host.Fit(ua, geno, w["K"], device=local_rank, admixture=1, accel_scheme=w["accel"], verbosity=1, abs_error=1e-300)
    fit.initialize(1234567)
    for _ in range(5):
        fit.em_step()
    st = hip.RunState(logL=fit.mod.logL, abs_error=1e-300, n_iter=fit.mod.n_iter)
    lib, ctx = hip.load(), C.c_void_p(fit.mod.dev)
    lib.mchip_synchronize(ctx)
    t0 = time.perf_counter()
    rc = lib.mchip_em_run(ctx, 0, steps, C.byref(st))       # plain EM: one batch, stopping rule on the device
    dt = time.perf_counter() - t0
    if rc or st.fatal or st.stopped:
        raise SystemExit("secondary run failed")
    out = {"workload": "%s: %s" % (name, w["desc"]), "value": steps / dt, "unit": "EM iterations/s",
           "ms_per_step": dt * 1e3 / steps, "steps": steps}
    fit.close()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", type=int, default=None, help="override the workload's acceleration scheme (0..6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0)
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra config-2 line (profiling runs: its kernels "
                    "carry the same names as the headline workload's)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    dist = None
    # rehearsal knobs (not used by the driver): several ranks on ONE GPU need gloo and a shared device index
    backend = os.environ.get("MC_BENCH_BACKEND", "nccl")
    if "MC_BENCH_DEVICE" in os.environ:
        local_rank = int(os.environ["MC_BENCH_DEVICE"])
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    cdev = dev if backend == "nccl" else torch.device("cpu")      # where the collective's tensors live

    from multiclust_amd import hip, host
    w = dict(WORKLOADS[args.workload])
    accel = w["accel"] if args.accel is None else args.accel
    ua, geno = gen_dataset(w["I"], w["L"], w["K"], w["ploidy"], w["maxal"], 20250117 + 3, dev)
    T = int(ua.sum())
    torch.cuda.empty_cache()

    fit = host.Fit(ua, geno, w["K"], device=local_rank, admixture=1, accel_scheme=accel, verbosity=1,
                   abs_error=1e-300)          # never "converges": exactly K timed steps
    hlib = hip.load()
    ctx = C.c_void_p(fit.mod.dev)
    # one initialisation per rank = unit `rank` of the serial program: random allele partition (rnd_init.c:456-482)
    # drawn from the glibc-compatible stream jumped ahead to unit * I*L*ploidy draws (mc_rng_jump), first M step on
    # the device (mchip_mstep_from_partition).  Not timed (the metric is the EM loop).
    rng = host.McRng()
    hostlib = host.load()
    hostlib.mc_srand(C.byref(rng), 1234567)
    hostlib.mc_rng_jump(C.byref(rng), rank * hostlib.mc_draws_per_init(C.byref(fit.opt), C.byref(fit.dat), w["K"]))
    rc = hostlib.mc_initialize_model(C.byref(fit.opt), C.byref(fit.dat), fit.mp, C.byref(rng))
    if rc:
        raise SystemExit("mc_initialize_model failed: %s" % hlib.mchip_last_error(ctx).decode())

    iters_per_step = 2 if accel else 1

    def one_step():
        if accel:
            fit.accelerated_em_step()
        else:
            fit.em_step()
        if fit.mod.fatal:
            raise SystemExit("EM stopped with fatal=%d" % fit.mod.fatal)

    for _ in range(args.warmup):
        one_step()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        hlib.mchip_synchronize(ctx)

    def run_steps(n):
        """n steps of the hot path as the host driver (mc_em) runs them: ONE batch whose stopping rule -- and, for the
        accelerated schemes, step size and accept test -- runs on the device."""
        st = hip.RunState(logL=fit.mod.logL, abs_error=1e-300, n_iter=fit.mod.n_iter)
        if accel and 1 <= accel <= 4:
            # one batch of cycles decided on the device (mc_em's own path for -s 1..4)
            rc = hlib.mchip_accel_run(ctx, fit.mod.pindex, accel, n, C.byref(st))
        elif accel:
            for _ in range(n):
                one_step()
            return
        else:
            rc = hlib.mchip_em_run(ctx, 0, n, C.byref(st))
        if rc or st.fatal or st.stopped:
            raise SystemExit("batched run: rc=%d fatal=%d stopped=%d" % (rc, st.fatal, st.stopped))
        fit.mod.n_iter, fit.mod.logL = st.n_iter, st.logL

    barrier()
    n_iter0 = fit.mod.n_iter
    hlib.mchip_profile_begin(ctx)
    t0 = time.perf_counter()
    run_steps(args.steps)
    best = torch.tensor([fit.mod.logL], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(best, op=dist.ReduceOp.MAX)     # the path's one exchange: best log likelihood over units
    barrier()
    dt = time.perf_counter() - t0
    total_ms = C.c_double()
    km = (C.c_double * hip.PROF_KINDS)()
    kl = (C.c_int * hip.PROF_KINDS)()
    hlib.mchip_profile_end(ctx, C.byref(total_ms), km, kl)
    n_iter = fit.mod.n_iter - n_iter0
    tmax = torch.tensor([dt], dtype=torch.float64, device=cdev)
    its = torch.tensor([float(n_iter)], dtype=torch.float64, device=cdev)
    if dist is not None:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(its, op=dist.ReduceOp.SUM)
    dt = float(tmax.item())
    total_iters = float(its.item())

    if rank == 0:
        B = algorithmic_bytes(w, T)
        names = ["column_pass", "individual_pass", "loglik_pass"]
        avg = [km[x] / kl[x] if kl[x] else 0.0 for x in range(hip.PROF_KINDS)]
        dom = max(range(2), key=lambda x: avg[x])
        ach = B[names[dom]] / (avg[dom] * 1e-3) / 1e9 if avg[dom] else 0.0
        nnz_flops = (5 * w["K"] + 5) * w["I"] * T          # SURVEY.md 8d, dense-over-columns upper bound
        value = total_iters / dt
        # HBM traffic per launch of the dominant kernel: PMC FETCH_SIZE/WRITE_SIZE from separate `rocprofv3 --pmc` passes
        # of this same command (scripts/summarize_profile.py; corrected as MI355X_MICROARCH.md prescribes), config 3 only
        traffic = None
        tf = os.path.join(ROOT, "profiles", "r01_v5_c3_traffic.json")
        if args.workload == "c3" and os.path.exists(tf):
            want = "k_column_counts<2, false>" if dom == 0 else "k_individual_sparse<2, true, false, true>"
            traffic = json.load(open(tf)).get(want, {}).get("hbm_bytes_per_launch_corrected")
        out = {
            "metric": "EM iterations/sec, IxLxK admixture",
            "value": value, "unit": "EM iterations/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s: %s" % (args.workload, w["desc"]), "I": w["I"], "L": w["L"], "T": T,
                       "ploidy": w["ploidy"], "K": w["K"], "accel_scheme": accel,
                       "em_iterations_per_step": iters_per_step, "units": "%d initialisation(s), one per GPU" % world,
                       "best_logL": float(best.item())},
            "roofline": {
                "bound": "hbm", "kernel": "k_column_counts (N-side sums)" if dom == 0 else "k_individual_sparse (S-side sums + logL)",
                "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "algorithmic_bytes_per_launch": B[names[dom]], "avg_launch_ms": avg[dom],
                "kernels_ms": {names[x]: avg[x] for x in range(hip.PROF_KINDS)},
                "launches": {names[x]: kl[x] for x in range(hip.PROF_KINDS)},
                "iteration_bytes": B["iteration"],
                "iteration_hbm_frac": B["iteration"] * (value / world) / 1e9 / HBM_PEAK_GBS,
                "fp64_valu_frac": nnz_flops * (value / world) / 1e12 / FP64_VALU_PEAK_TF,
                "fp64_note": "kernels are FP64-issue-bound, not HBM-bound: (5K+5) flop per cell x I x T per iteration over the 78.6 TF/s "
                             "vector-FP64 spec peak (best measured v_fma_f64 rate on this chip: %.1f TF/s); PMC counters of the same "
                             "command at config 3 (profiles/r01_v4_c3_sq_counters.txt): vector ALUs busy 94 %% of the column pass, "
                             "83 %% of the individual pass" % FP64_VALU_MEASURED_TF,
            },
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, ua, geno, accel, args.cpu_budget, local_rank)
            out["cpu_baseline"]["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
    fit.close()
    if rank == 0:
        if world == 1 and args.workload == "c3" and not args.no_secondary:
            out["secondary"] = secondary_run("c2", dev, local_rank)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
