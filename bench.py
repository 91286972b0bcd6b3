#!/usr/bin/env python3
"""bench.py -- EM iterations/s of the MI355X EM hot path on BASELINE.json's workloads.

  python bench.py --gpus N --steps K --warmup W [--workload c3|c4|c5|c2|c1|c5fit]

With N > 1 and no RANK in the environment this process only checks that N devices are visible and starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py <same flags>` as a
child (before anything touches the GPU; a child process, never an exec) and exits with its code; launched by
torch.distributed.run itself (RANK set) it is one rank of N, one rank per GPU over RCCL.

Workloads
  c3 (default; BASELINE.json configs[2], the configuration north_star's target is quoted on): 10 000 diploid individuals x
     100 000 loci, M_l ~ U{2,3,4}, admixture K = 8, SQUAREM-3 (-s 3), one initialisation per GPU.  A step is one
     accelerated_em_step() cycle = 2 EM iterations + 2 stand-alone log-likelihood passes + step size + extrapolation and
     projection (accel_em.c:35-114).  scaling = "weak": rank r fits unit r of the serial program's rand() stream.
  c4 (configs[3]): the same data and model, `--units` (50) random initialisations sharded unit u -> rank u mod N
     (multiclust.c:516-653), every unit = device-side initialisation from its position in the serial rand() stream + K
     SQUAREM cycles (-T 2K-1), all inside the timed region; one all-reduce completes the per-unit result table, every rank
     replays the serial bookkeeping.  scaling = "strong" (the 50 units are the job).
  c5 (configs[4]): 5 000 tetraploid individuals x 50 000 loci, bootstrap of K = 7 against K = 8 (-b 200): replicate b ->
     rank b mod N (multiclust.c:675-708); a replicate = data set generated on the device from the H0 fit at the
     replicate's position in the rand() stream + one initialisation and K+1 plain EM iterations (-T K) of each model; one
     all-reduce completes the table of test statistics.  scaling = "strong".
  c2, c1, c5fit: single fits at the shapes of configs[1], configs[0] and configs[4] (plain EM), measured like c3.

value = EM iterations/s summed over ranks (n_iter increments of stop(), em_alg.c:103, over the barrier-to-barrier wall time,
max over ranks), inputs resident in HBM.  The same JSON line carries `roofline` for the dominant kernel (HIP events on the
library's own stream) and, at N = 1, `cpu_baseline`: rank 0 times the reference's own em() (oracle/_ref/ref_time: the
unmodified sources of the path compiled by oracle/Makefile where /root/reference exists; the binary travels with the tree,
the sources do not; kind "reference") on a bounded sample of the workload -- all individuals, the first loci, the same
iterations -- and, under `port`, the CPU oracle on a larger one (one host core for the single fits, min(units, 16) processes
at once -- one unit each -- for c4 and c5).  Both carry `parity` = the HIP path on their own sample against them.  Without the
binary the object leads with the oracle (kind "port").  Reference and oracle are the baseline and the checker here, never the
measured path.
"""
import argparse
import ctypes as C
import glob
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
FP64_VALU_PEAK_TF = 78.6    # vector FP64 peak (spec), for the secondary fraction only
FP64_VALU_MEASURED_TF = 70.5 # best v_fma_f64 rate measured on this chip (profiles/r01_fp64_microbench.txt)
SEED = 1234567              # -r of every run below

WORKLOADS = {
    # name: I, L, ploidy, max alleles, K, accel scheme, data seed offset, description
    "c3": dict(I=10000, L=100000, ploidy=2, maxal=4, K=8, accel=3, dseed=3,
               desc="10000 diploid x 100000 loci, M_l~U{2,3,4}, admixture K=8, SQUAREM-3 (-s 3)"),
    "c4": dict(I=10000, L=100000, ploidy=2, maxal=4, K=8, accel=3, dseed=3,
               desc="10000 diploid x 100000 loci, M_l~U{2,3,4}, admixture K=8, SQUAREM-3, random initialisations sharded over GPUs"),
    "c5": dict(I=5000, L=50000, ploidy=4, maxal=4, K=8, accel=0, dseed=5,
               desc="5000 tetraploid x 50000 loci, M_l~U{2,3,4}, bootstrap of K=7 vs K=8, replicates sharded over GPUs"),
    "c2": dict(I=2000, L=20000, ploidy=2, maxal=2, K=5, accel=0, dseed=2,
               desc="2000 diploid x 20000 biallelic loci, admixture K=5, plain EM (-s 0)"),
    "c1": dict(I=100, L=500, ploidy=2, maxal=2, K=3, accel=0, dseed=1,
               desc="100 diploid x 500 biallelic loci, admixture K=3, plain EM"),
    # reduced shapes of c4 / c5 for rehearsals of the sharded paths (tests/test_gpu_bench.py: two ranks on one GPU over gloo)
    "c4s": dict(I=600, L=4000, ploidy=2, maxal=4, K=8, accel=3, dseed=3,
                desc="600 diploid x 4000 loci (reduced c4), admixture K=8, SQUAREM-3, random initialisations sharded over GPUs"),
    "c5s": dict(I=400, L=3000, ploidy=4, maxal=4, K=4, accel=0, dseed=5,
                desc="400 tetraploid x 3000 loci (reduced c5), bootstrap of K=3 vs K=4, replicates sharded over GPUs"),
    "c5fit": dict(I=5000, L=50000, ploidy=4, maxal=4, K=8, accel=0, dseed=5,
                  desc="5000 tetraploid x 50000 loci, M_l~U{2,3,4}, admixture K=8, plain EM, single fit"),
}


# ------------------------------------------------------------------------------------------------ launcher
def launch_ranks(n):
    """--gpus N without RANK: start the N ranks as a child process.  torch.cuda.device_count() does not initialise the
    GPU on this image; nothing else here touches it."""
    if "MC_BENCH_DEVICE" not in os.environ:            # rehearsal knob: all ranks share that one device
        import torch
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d device(s) visible\n" % (n, have))
            sys.exit(3)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    # the ranks' stdout is passed on line by line: the one JSON line to stdout, anything else a library printed there (gloo's
    # connection banner in rehearsals) to stderr
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
        sys.stdout.flush()
    sys.exit(proc.wait())


class Env:
    """rank, device and the process group (torch is plumbing: device memory for the synthetic data, the collective)"""

    def __init__(self, gpus):
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != gpus:
            raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (gpus, self.world))
        import torch
        self.torch = torch
        # rehearsal knobs (not used by the driver): several ranks on ONE GPU need gloo and a shared device index
        self.backend = os.environ.get("MC_BENCH_BACKEND", "nccl")
        if "MC_BENCH_DEVICE" in os.environ:
            self.local_rank = int(os.environ["MC_BENCH_DEVICE"])
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        if self.local_rank >= torch.cuda.device_count():
            raise SystemExit("bench.py: rank %d has no device %d" % (self.rank, self.local_rank))
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.dist = None
        # MC_BENCH_FORCE_PG=1: a one-rank run still creates the process group and sends its exchanges through it, so that
        # torch's RCCL initialisation, all_reduce and barrier on this image have run on a one-GPU box (tests/test_gpu_bench.py)
        # before the driver's first N > 1 launch depends on them
        force_pg = self.world == 1 and os.environ.get("MC_BENCH_FORCE_PG") == "1"
        if self.world > 1 or force_pg:
            import torch.distributed as dist
            if force_pg and "MASTER_ADDR" not in os.environ:
                with socket.socket() as s:
                    s.bind(("127.0.0.1", 0))
                    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(s.getsockname()[1]), RANK="0", WORLD_SIZE="1")
            if self.backend == "nccl":
                dist.init_process_group("nccl", device_id=self.dev)
            else:
                dist.init_process_group(self.backend)
            self.dist = dist
        self.cdev = self.dev if self.backend == "nccl" else torch.device("cpu")      # where the collective's tensors live
        self.collectives = 0          # all-reduces and barriers that went through the process group (reported on the line)

    def barrier(self, ctx=None):
        if self.dist is not None:
            self.dist.barrier()
            self.collectives += 1
        self.torch.cuda.synchronize()
        if ctx is not None:
            from multiclust_amd import hip
            hip.load().mchip_synchronize(ctx)

    def reduce(self, values, op):
        t = self.torch.tensor(values, dtype=self.torch.float64, device=self.cdev)
        if self.dist is not None:
            self.dist.all_reduce(t, op=getattr(self.dist.ReduceOp, op))
            self.collectives += 1
        return t.cpu().tolist()

    def same_everywhere(self, what, *arrays):
        """Every rank generates the data set itself (gen_dataset on its own device).  The units of c4 / c5 are fits of ONE data
        set: before anything is timed, all-reduce MIN and MAX of a checksum of it and stop the job if two ranks disagree --
        otherwise the per-unit table would still complete, on different data.  Returns the checksum."""
        import zlib
        crc = [float(zlib.crc32(memoryview(np.ascontiguousarray(a)).cast("B"))) for a in arrays]      # 32-bit: exact in a double
        lo, hi = self.reduce(crc, "MIN"), self.reduce(crc, "MAX")
        if lo != hi or lo != crc:
            sys.stderr.write("bench.py: rank %d: %s differs between ranks (checksums %s, min %s, max %s)\n" % (self.rank, what, crc, lo, hi))
            sys.stderr.flush()
            if self.dist is not None:
                self.dist.barrier()
            sys.exit(4)
        return [int(c) for c in crc]

    def close(self):
        if self.dist is not None:
            self.dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ data
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
    torch.cuda.empty_cache()
    return ua.to(torch.int32).cpu().numpy(), geno


def workload_data(w, env):
    ua, geno = gen_dataset(w["I"], w["L"], w["K"], w["ploidy"], w["maxal"], 20250117 + w["dseed"], env.dev)
    if os.environ.get("MC_BENCH_CORRUPT_RANK") == str(env.rank):      # test knob: this rank's data set differs in one byte
        geno[0, 0, 0] ^= 1
    w["data_crc32"] = env.same_everywhere("the synthetic data set", ua, geno) if env.dist is not None else None
    return ua, geno


def algorithmic_bytes(w, T, K=None):
    I, L, p = w["I"], w["L"], w["ploidy"]
    K = K or w["K"]
    g = I * L * p
    return {
        "iteration": g + 16 * K * T + 16 * I * K,            # B_it
        "column_pass": g + 16 * K * T + 8 * I * K,           # genotype + read P, write N-side sums + read Q
        "individual_pass": g + 8 * K * T + 16 * I * K,       # genotype + read P + read Q, write S-side sums
        "loglik_pass": g + 8 * K * T + 8 * I * K,
        "individual_dual_pass": g + 16 * K * T + 24 * I * K,  # genotype + read P and Q of two parameter sets, write S-side sums
    }


# ------------------------------------------------------------------------------------------------ CPU baseline + parity
def cpu_oracle_run(ob, I, Ls, p, K, ua_s, geno_s, q0, p0, lb, accel, iters):
    """`iters` EM iterations of the CPU oracle (fused order) from (q0, p0): (model, seconds)"""
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
    return mod, time.perf_counter() - t0


def cpu_worker(path):
    """One more host core of a multi-unit cpu_baseline (c4, c5): a child process of bench.py's cpu_baseline leg that never
    touches the GPU.  Loads the sample the parent wrote, says "ready", waits for "go" on stdin, runs the same oracle
    iterations from its own starting point and prints {"iters", "dt"}."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_bind as ob
    d = np.load(path)
    I, Ls, p = d["geno"].shape
    K, accel, iters, lb = int(d["K"]), int(d["accel"]), int(d["iters"]), float(d["lb"])
    ua_s, geno_s = np.ascontiguousarray(d["ua"]), np.ascontiguousarray(d["geno"])
    q0, p0 = np.ascontiguousarray(d["q0"]), np.ascontiguousarray(d["p0"])
    print("ready", flush=True)
    sys.stdin.readline()
    mod, dt = cpu_oracle_run(ob, I, Ls, p, K, ua_s, geno_s, q0, p0, lb, accel, iters)
    print(json.dumps({"iters": int(mod.n_iter), "dt": dt}), flush=True)


def cpu_baseline(w, ua, geno, accel, budget_s=20.0, device=0, units=1, ref_budget_s=15.0):
    """The CPU oracle (oracle/mc_oracle.c, fused order) on a bounded sample of the same workload: the first L_s loci of every
    individual, sized for about `budget_s` seconds of one host core.  A workload of independent units (c4's initialisations,
    c5's replicates: SURVEY.md 8d, the reference's own scaling model) is timed on min(units, host cores) processes at once,
    each fitting one unit's worth of the sample; `value` is then the sum of their rates and `cores` their number."""
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
    # a one-GPU box shares its host: 16 cores are this process's to use, whatever os.cpu_count() says (and every worker
    # holds its own copy of the sample)
    try:
        avail = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        avail = os.cpu_count() or 1
    procs = max(1, min(units, avail, 16))
    workers, tmp = [], None
    if procs > 1:
        import tempfile
        tmp = tempfile.NamedTemporaryFile(suffix=".npz", delete=False)
        tmp.close()
        np.savez(tmp.name, ua=ua_s, geno=geno_s, q0=q0, p0=p0, K=K, accel=accel, iters=iters, lb=lb)
        for _ in range(procs - 1):
            workers.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", tmp.name],
                                            stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for wk in workers:
            if wk.stdout.readline().strip() != "ready":
                raise SystemExit("cpu_baseline worker failed to start")
        for wk in workers:
            wk.stdin.write("go\n")
            wk.stdin.flush()
    mod, dt = cpu_oracle_run(ob, I, Ls, p, K, ua_s, geno_s, q0, p0, lb, accel, iters)
    it_s_sample = mod.n_iter / dt
    rates = [it_s_sample]
    for wk in workers:
        r = json.loads(wk.stdout.readline())
        wk.wait()
        rates.append(r["iters"] / r["dt"])
    if tmp:
        os.unlink(tmp.name)
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
    how = "on one core" if procs == 1 else "on each of %d processes at once (one unit each; rates summed: %.3f-%.3f it/s per process)" % (
        procs, min(rates), max(rates))
    # the reference itself (stock build, one core of the planning container's 2.1 GHz Xeon) costs 34.7 ns per
    # (individual, locus, cluster) cell on diploid biallelic data = 17 ns per (i, l, m, k) cell, the same at 500 x 5 000 and at
    # 2 000 x 20 000.  It cannot run configs 3-5 (0.5 TB of diklm, O(n^2) reader): an extrapolation, labelled as one
    ref_ns_per_cell = 17.0
    ref_cells = float(I) * float(int(ua.sum())) * K
    reference_extrapolated = {
        "value": 1.0 / (ref_ns_per_cell * 1e-9 * ref_cells), "unit": "EM iterations/s", "cores": 1, "kind": "extrapolation",
        "basis": "BASELINE.md section 2: %.0f ns per (i,l,m,k) cell measured for the unmodified reference on one 2.1 GHz Xeon core at "
                 "config-1 and config-2 size, times I*T*K = %.3g cells; the reference cannot run this size (diklm alone is 0.5 TB at "
                 "config 3)" % (ref_ns_per_cell, ref_cells),
    }
    port = {
        "value": sum(rates) * Ls / L, "unit": "EM iterations/s", "cores": procs, "kind": "port",
        "sample": "first %d of %d loci, all %d individuals, %d EM iterations (%s) in %.1f s %s; "
                  "scaled by %d/%d (cost is linear in loci)" % (Ls, L, I, mod.n_iter, "SQUAREM-3 cycles" if accel else "plain EM", dt, how, Ls, L),
        "sample_value": sum(rates), "parity": parity,
    }
    why = "oracle/_ref/ref_time is not in this tree (it is built where /root/reference exists)"
    try:
        ref = reference_leg(w, ua_s, geno_s, q0, p0, accel, iters, procs, ref_budget_s, device, int(ua.sum()))
    except (Exception, SystemExit) as e:      # the baseline must not cost the run its result line
        ref, why = None, "the reference leg failed (%s: %s)" % (type(e).__name__, str(e)[:200])
        sys.stderr.write("bench.py: cpu_baseline: %s; reporting the oracle's figure\n" % why)
    if ref is None:
        port["reference_extrapolated"] = reference_extrapolated
        port["reference"] = why + ": the CPU figure is the oracle's"
        return port
    ref["port"] = port
    ref["reference_extrapolated"] = reference_extrapolated
    return ref


def attach_cpu_baseline(out, *a, **kw):
    """cpu_baseline + the ratios; a failure in the CPU legs is reported inside the object instead of costing the line"""
    try:
        out["cpu_baseline"] = cpu_baseline(*a, **kw)
        gpu_over_cpu(out)
    except (Exception, SystemExit) as e:
        sys.stderr.write("bench.py: cpu_baseline failed: %s: %s\n" % (type(e).__name__, e))
        out["cpu_baseline"] = {"value": None, "unit": "EM iterations/s", "cores": 0, "kind": "port", "sample": None,
                               "error": "%s: %s" % (type(e).__name__, str(e)[:300])}
    # a line whose CPU figure is not the reference's own em() (binary absent, its run failed, or the whole leg failed) says so at
    # the top level, where a reader of the line -- or a check of it -- cannot miss it
    out["cpu_baseline_degraded"] = out["cpu_baseline"].get("kind") != "reference"


def gpu_over_cpu(out):
    """the ratio against the baseline the object leads with (the reference itself where its binary is in the tree) and,
    then, against the oracle too"""
    cb = out["cpu_baseline"]
    cb["gpu_over_cpu"] = out["value"] / cb["value"]
    if "port" in cb:
        cb["port"]["gpu_over_cpu"] = out["value"] / cb["port"]["value"]


REF_TIME = os.environ.get("MC_BENCH_REF_TIME", os.path.join(ROOT, "oracle", "_ref", "ref_time"))      # (the knob: tests of the failure path)


def reference_leg(w, ua_s, geno_s, q0, p0, accel, iters, procs, budget_s, device, T_full):
    """The reference's own em() (oracle/ref_time.c: the unmodified sources of the path, compiled in place by oracle/Makefile;
    the binary travels, the sources do not) on the head of the oracle's sample, sized for about `budget_s` seconds of one host
    core: all individuals, the first L_r loci, `iters` EM iterations from the same starting parameters -- and the same
    iterations through the HIP path, compared with what the reference left.  None where the binary is absent."""
    if not os.access(REF_TIME, os.X_OK) or budget_s <= 0:
        return None
    import tempfile
    I, p, K = w["I"], w["ploidy"], w["K"]
    cum = np.cumsum(ua_s)
    args = ["-f", "sample", "-a", "-k", str(K)] + (["-s", str(accel)] if accel else [])

    def run(seconds, ns_per_cell):
        cols = max(1, int(seconds * 1e9 / (ns_per_cell * I * K * iters)))
        Lr = int(max(1, min(len(ua_s), np.searchsorted(cum, cols, side="right"))))
        Tr = int(cum[Lr - 1])
        ua_r, geno_r = np.ascontiguousarray(ua_s[:Lr], dtype=np.int32), np.ascontiguousarray(geno_s[:, :Lr, :])
        p0_r = np.ascontiguousarray(p0[:, :Tr])
        with tempfile.TemporaryDirectory(prefix="mcref.") as d:
            ua_r.tofile(os.path.join(d, "ua.i32"))
            geno_r.tofile(os.path.join(d, "geno.u8"))
            np.ascontiguousarray(q0).tofile(os.path.join(d, "q0.f64"))
            p0_r.tofile(os.path.join(d, "p0.f64"))
            cmd = [REF_TIME, d, str(I), str(Lr), str(p), str(K), str(iters - 1), "--"] + args
            # independent units (c4, c5): the reference's scaling model is one process per unit (multiclust.c:143-145)
            runs = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for _ in range(procs)]
            try:
                outs = [r.communicate(timeout=max(120.0, 20.0 * seconds)) for r in runs]
            except subprocess.TimeoutExpired:
                for r in runs:
                    r.kill()
                raise SystemExit("cpu_baseline: the reference run did not finish in time")
            if any(r.returncode for r in runs):
                raise SystemExit("cpu_baseline: the reference run failed: %s" % outs[0][1][-400:])
            res = [json.loads(o[0]) for o in outs]
            q_ref = np.fromfile(os.path.join(d, "q_ref.f64")).reshape(I, K)
            p_ref = np.fromfile(os.path.join(d, "p_ref.f64")).reshape(K, Tr)
        return Lr, Tr, ua_r, geno_r, p0_r, res, q_ref, p_ref

    # its cost per (individual, allele column, cluster) cell and counted iteration depends on the host and on I (the jagged
    # diklm rows miss the caches more as I grows): 9 ns on a GPU box's host, 30-65 ns on the build container.  A pilot a
    # tenth of the budget at the slow figure, then the sample the budget allows at the measured one
    Lr, Tr, ua_r, geno_r, p0_r, res, q_ref, p_ref = run(budget_s / 10.0, 60.0)
    pilot_ns = 1e9 * max(r["em_s"] for r in res) / res[0]["n_iter"] / (float(I) * Tr * K)
    if Lr < len(ua_s) and pilot_ns < 50.0:
        Lr, Tr, ua_r, geno_r, p0_r, res, q_ref, p_ref = run(0.85 * budget_s, max(pilot_ns, 1.0))
    r0 = res[0]
    rates = [r["n_iter"] / r["em_s"] for r in res]
    from multiclust_amd import host
    fit = host.Fit(ua_r, geno_r, K, device=device, admixture=1, accel_scheme=accel, verbosity=1, abs_error=1e-300, rel_error=0.0,
                   max_iter=iters - 1)
    fit.set_params(q0, p0_r)
    fit.em()
    gq, gp = fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex)
    big_q, big_p = q_ref > 1e-6, p_ref > 1e-6
    parity = {
        "abs_dlogL": abs(fit.mod.logL - r0["logL"]), "rel_dlogL": abs(fit.mod.logL - r0["logL"]) / abs(r0["logL"]),
        "max_rel_dQ": float(np.max(np.abs(gq - q_ref)[big_q] / q_ref[big_q])),
        "max_rel_dP": float(np.max(np.abs(gp - p_ref)[big_p] / p_ref[big_p])),
        "same_n_iter": int(fit.mod.n_iter == r0["n_iter"]),
        "note": "HIP path (mc_em) vs the reference's em() on this sample: same starting parameters, same stopping rule "
                "(-T %d); Q/P entries > 1e-6" % (iters - 1),
    }
    fit.close()
    how = "on one core" if procs == 1 else "on each of %d processes at once (one unit each; rates summed: %.4f-%.4f it/s per process)" % (
        procs, min(rates), max(rates))
    return {
        "value": sum(rates) * Tr / T_full, "unit": "EM iterations/s", "cores": procs, "kind": "reference",
        "sample": "the reference's em() (unmodified sources, gcc -O3, its reader bypassed: oracle/ref_time.c) on the first %d of %d "
                  "loci (%d of %d allele columns), all %d individuals, %d EM iterations (%s) in %.1f s %s, %.1f s of allocation "
                  "before them; scaled by %d/%d (cost is linear in allele columns)" % (
                      Lr, w["L"], Tr, T_full, I, r0["n_iter"], "SQUAREM-%d cycles" % accel if accel else "plain EM", r0["em_s"], how,
                      r0["setup_s"], Tr, T_full),
        "sample_value": sum(rates), "ns_per_cell": 1e9 * r0["em_s"] / r0["n_iter"] / (float(I) * Tr * K), "parity": parity,
    }


# ------------------------------------------------------------------------------------------------ roofline object
PASS_NAMES = ["column_pass", "individual_pass", "loglik_pass", "individual_dual_pass"]
PASS_KERNELS = {"column_pass": ("k_column_counts", "k_column_counts (N-side sums)"),
                "individual_pass": ("k_individual_sparse_w<2, true, false, true, false>", "k_individual_sparse_w (S-side sums + logL)"),
                "loglik_pass": ("k_individual_sparse_w<2, false", "k_individual_sparse_w (stand-alone log likelihood)"),
                "individual_dual_pass": ("k_individual_sparse_w<2, true, false, true, true>",
                                         "k_individual_sparse_w, dual (S-side sums of the extrapolated point + logL of the second EM iterate)")}
FP64_SUSTAINED_TF = 62.0     # v_fma_f64 stream on operands that keep changing, held for 5 s: 2.00-2.05 GHz (power management; 2.35 GHz
                             # and 71-74 TF/s on constant operands), 4.3-4.4 cycles per wave-instruction
                             # (profiles/r03_fp64_sustained_clock.txt); the spec peak assumes 2.4 GHz and 4.0


def latest_traffic(tag):
    """HBM bytes per launch from the PMC passes of this same command in an EARLIER run (scripts/summarize_profile.py; corrected
    as MI355X_MICROARCH.md prescribes): (newest profiles/r*_<tag>_traffic.json, its name), or (None, None).  Counters cannot be
    read inside a timed run; the line says where the figure comes from (`traffic_source`) and whether that file was taken from
    the very library this run has loaded (`traffic_build_matches`: the file's `_build.library_sha256` against the library's)."""
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_%s_traffic.json" % tag)))
    return (json.load(open(files[-1])), os.path.relpath(files[-1], ROOT)) if files else (None, None)


def library_sha256():
    import hashlib
    from multiclust_amd import hip
    try:
        return hashlib.sha256(open(hip.lib_path(), "rb").read()).hexdigest()
    except OSError:
        return None


def profile_begin(ctx):
    from multiclust_amd import hip
    hip.load().mchip_profile_begin(ctx)


def build_roofline(w, T, K, kernel_ms, launches, steps, it_per_s_per_gpu, nnz, traffic_table=None, traffic_file=None, lib_sha=None):
    """The line's `roofline` object from the library's HIP-event figures: kernel_ms[x] / launches[x] = summed duration and number
    of working launches of pass x (PASS_NAMES) over `steps` timed steps.  The dominant kernel is the pass with the most time per
    step; `achieved` / `peak` / `frac` are its ALGORITHMIC bytes per launch over its average launch time against the HBM peak
    (the contract's definition), `bound` says what actually binds it."""
    B = algorithmic_bytes(w, T, K)
    avg = [kernel_ms[x] / launches[x] if launches[x] else 0.0 for x in range(len(PASS_NAMES))]
    per_step = [kernel_ms[x] / max(steps, 1) for x in range(len(PASS_NAMES))]
    dom = max(range(len(PASS_NAMES)), key=lambda x: per_step[x])
    name = PASS_NAMES[dom]
    ach = B[name] / (avg[dom] * 1e-3) / 1e9 if avg[dom] else 0.0
    flops_cell = 5 * K + 5                                   # flops per non-empty cell (+ one log)
    traffic, traffic_build, pass_traffic = None, None, {}
    if traffic_table:
        traffic_build = traffic_table.get("_build")
        for x in range(len(PASS_NAMES)):
            for kname, rec in traffic_table.items():
                if kname != "_build" and kname.startswith(PASS_KERNELS[PASS_NAMES[x]][0]):
                    pass_traffic[PASS_NAMES[x]] = rec.get("hbm_bytes_per_launch_corrected")
                    break
        traffic = pass_traffic.get(name)
    return {
        "bound": "fp64-valu", "kernel": PASS_KERNELS[name][1],
        "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_source": traffic_file if traffic is not None else None,
        "traffic_build": traffic_build if traffic is not None else None,
        "traffic_build_matches": (bool(traffic_build and lib_sha and traffic_build.get("library_sha256") == lib_sha)
                                  if traffic is not None else None),
        "pass_traffic": pass_traffic or None,
        "algorithmic_bytes_per_launch": B[name], "avg_launch_ms": avg[dom],
        "kernels_ms": {PASS_NAMES[x]: avg[x] for x in range(len(PASS_NAMES))},
        "kernels_ms_per_step": {PASS_NAMES[x]: per_step[x] for x in range(len(PASS_NAMES))},
        "launches": {PASS_NAMES[x]: launches[x] for x in range(len(PASS_NAMES))},
        "pass_hbm_frac": {PASS_NAMES[x]: (B[PASS_NAMES[x]] / (avg[x] * 1e-3) / 1e9 / HBM_PEAK_GBS if avg[x] else 0.0)
                          for x in range(len(PASS_NAMES))},
        "iteration_bytes": B["iteration"],
        "iteration_hbm_frac": B["iteration"] * it_per_s_per_gpu / 1e9 / HBM_PEAK_GBS,
        "nonempty_cells": nnz, "dense_cells": w["I"] * T,
        "fp64_valu_frac": flops_cell * nnz * it_per_s_per_gpu / 1e12 / FP64_VALU_PEAK_TF,
        "fp64_valu_frac_dense_cells": flops_cell * w["I"] * T * it_per_s_per_gpu / 1e12 / FP64_VALU_PEAK_TF,
        "fp64_valu_frac_of_sustained": flops_cell * w["I"] * T * it_per_s_per_gpu / 1e12 / FP64_SUSTAINED_TF,
        "bound_note": "`bound` names what binds the kernel: FP64 vector issue.  `achieved`/`peak`/`frac` are the HBM figures the "
                      "contract defines (algorithmic bytes per launch over the launch time against 8 TB/s); the HBM roof is not "
                      "reachable at FP64 (about 45 flop per genotype byte at K = 8 against a ridge of 10).  fp64_valu_frac = "
                      "(5K+5) flop per NON-EMPTY cell (SURVEY.md 8d; cells counted on the device at upload) x iterations/s over "
                      "the 78.6 TF/s vector-FP64 spec peak; _dense_cells counts every (individual, allele column) cell, which is "
                      "what the column pass multiplies through; _of_sustained is the same over the %.1f TF/s a v_fma_f64 stream on "
                      "changing operands sustains on this chip (power management holds it at 2.0-2.05 GHz; the EM passes run at "
                      "2.02-2.14 GHz: profiles/r03_fp64_sustained_clock.txt)" % FP64_SUSTAINED_TF,
    }


def roofline_object(ctx, w, T, K, it_per_s_per_gpu, workload, nnz, steps):
    from multiclust_amd import hip
    hlib = hip.load()
    total_ms = C.c_double()
    km = (C.c_double * hip.PROF_KINDS)()
    kl = (C.c_int * hip.PROF_KINDS)()
    hlib.mchip_profile_end(ctx, C.byref(total_ms), km, kl)
    table, fname = latest_traffic("c3" if workload == "c4" else workload)
    return build_roofline(w, T, K, list(km), list(kl), steps, it_per_s_per_gpu, nnz, table, fname, library_sha256())


def data_counts(ctx):
    from multiclust_amd import hip
    a, b = C.c_uint64(), C.c_uint64()
    hip.load().mchip_data_counts(ctx, C.byref(a), C.byref(b))
    return int(a.value), int(b.value)


# ------------------------------------------------------------------------------------------------ single fit per GPU (c3, c2, c1, c5fit)
def run_single_fit(args, env, name, ua, geno, steps, warmup, with_roofline=True):
    from multiclust_amd import hip, host
    w = WORKLOADS[name]
    accel = w["accel"] if args.accel is None else args.accel
    T = int(ua.sum())
    fit = host.Fit(ua, geno, w["K"], device=env.local_rank, admixture=1, accel_scheme=accel, verbosity=1,
                   abs_error=1e-300)          # never "converges": exactly `steps` timed steps
    hlib = hip.load()
    ctx = C.c_void_p(fit.mod.dev)
    # one initialisation per rank = unit `rank` of the serial program: random allele partition (rnd_init.c:456-482)
    # drawn on the device from the glibc-compatible stream jumped ahead to unit * I*L*ploidy draws, first M step on
    # the device.  Not timed here (the metric is the EM loop; c4 times it).
    rng = host.McRng()
    hostlib = host.load()
    hostlib.mc_srand(C.byref(rng), SEED)
    hostlib.mc_rng_jump(C.byref(rng), env.rank * hostlib.mc_draws_per_init(C.byref(fit.opt), C.byref(fit.dat), w["K"]))
    rc = hostlib.mc_initialize_model(C.byref(fit.opt), C.byref(fit.dat), fit.mp, C.byref(rng))
    if rc:
        raise SystemExit("mc_initialize_model failed: %s" % hlib.mchip_last_error(ctx).decode())

    def one_step():
        fit.accelerated_em_step() if accel else fit.em_step()
        if fit.mod.fatal:
            raise SystemExit("EM stopped with fatal=%d" % fit.mod.fatal)

    # clocks: a fresh process finds the GPU idle, and the W warm-up steps of a short run (50 ms at config 3, 1 ms at config 2) end
    # before its clocks have come up -- the same kernels ran 3-10 % slower in the timed region than minutes into a long run.
    # Log-likelihood passes over the resident data (no state changes) keep the device busy for --settle seconds first.
    if args.settle > 0:
        ll = C.c_double()
        t_settle = time.perf_counter()
        while time.perf_counter() - t_settle < args.settle:
            for _ in range(8):
                hlib.mchip_loglik(ctx, fit.mod.pindex, C.byref(ll))
    for _ in range(warmup):
        one_step()

    def run_steps(n):
        """n steps of the hot path as the host driver (mc_em) runs them: ONE batch whose stopping rule -- and, for the
        accelerated schemes, step size and accept test -- runs on the device."""
        st = hip.RunState(logL=fit.mod.logL, abs_error=1e-300, n_iter=fit.mod.n_iter)
        if accel and 1 <= accel <= 4:
            rc = hlib.mchip_accel_run(ctx, fit.mod.pindex, accel, n, C.byref(st))
        elif accel:
            for _ in range(n):
                one_step()
            return
        else:
            rc = hlib.mchip_em_run(ctx, 0, n, C.byref(st))
        if rc or st.fatal or st.stopped:
            raise SystemExit("batched run: rc=%d fatal=%d stopped=%d after %d iterations%s" % (
                rc, st.fatal, st.stopped, st.n_iter, " (the fit reached its fixed point to the last bit inside the timed region: use fewer "
                "--steps for a workload this small)" if st.stopped and not (rc or st.fatal) else ""))
        fit.mod.n_iter, fit.mod.logL = st.n_iter, st.logL

    env.barrier(ctx)
    n_iter0 = fit.mod.n_iter
    if with_roofline:
        profile_begin(ctx)
    t0 = time.perf_counter()
    run_steps(steps)
    best = env.reduce([fit.mod.logL], "MAX")[0]     # the path's one exchange: best log likelihood over units
    env.barrier(ctx)
    dt = time.perf_counter() - t0
    dt = env.reduce([dt], "MAX")[0]
    total_iters = env.reduce([float(fit.mod.n_iter - n_iter0)], "SUM")[0]
    value = total_iters / dt
    out = {
        "value": value, "ms_per_step": dt * 1e3 / steps, "steps": steps,
        "config": {"workload": "%s: %s" % (name, w["desc"]), "I": w["I"], "L": w["L"], "T": T, "ploidy": w["ploidy"], "K": w["K"],
                   "accel_scheme": accel, "em_iterations_per_step": 2 if accel else 1,
                   "units": "%d initialisation(s), one per GPU" % env.world, "best_logL": best},
    }
    if accel:      # with an accelerated scheme both rates; a cycle = 2 EM iterations + 2 log-likelihood evaluations
        out["config"]["accelerated_cycles_per_s"] = steps * env.world / dt
    if with_roofline and env.rank == 0:
        nnz, _ = data_counts(ctx)
        out["roofline"] = roofline_object(ctx, w, T, w["K"], value / env.world, name, nnz, steps)
    elif with_roofline:
        hlib.mchip_profile_end(ctx, None, None, None)
    if with_roofline and args.stability > 0:
        # how stable is a figure taken over `steps` steps?  The same batch again, args.stability times, right behind the timed one,
        # launched the same way (event pairs around the passes), each timed by itself on this rank: min / median / max
        rates = []
        profile_begin(ctx)
        for _ in range(args.stability):
            n0, t1 = fit.mod.n_iter, time.perf_counter()
            run_steps(steps)
            hlib.mchip_synchronize(ctx)
            rates.append((fit.mod.n_iter - n0) / (time.perf_counter() - t1))
        hlib.mchip_profile_end(ctx, None, None, None)
        rates.sort()
        out["stability"] = {"unit": "EM iterations/s on one GPU (rank 0)", "batches": len(rates), "steps_per_batch": steps,
                            "min": rates[0], "median": rates[len(rates) // 2], "max": rates[-1],
                            "timed_region": value / env.world,
                            "note": "the timed region's batch repeated right behind it; clocks differ from box to box by +-6 % "
                                    "(DESIGN.md 4.3), within a run by what this shows"}
    return out, fit, accel


# ------------------------------------------------------------------------------------------------ c1: through the reader
def write_structure(path, ua, geno):
    """the synthetic genotype as a STRUCTURE file (read_file.c:38-300): a line of locus names, then `ploidy` lines per individual
    (name, locale, one allele code per locus); allele index m of a locus is written as code m + 1"""
    I, L, p = geno.shape
    with open(path, "w") as f:
        f.write(" ".join("loc%d" % (l + 1) for l in range(L)) + "\n")
        for i in range(I):
            for a in range(p):
                f.write("ind%d pop%d %s\n" % (i, i % 3, " ".join(str(int(m) + 1) if m != 0xFF else "-9" for m in geno[i, :, a])))


def run_c1_reader(env, iters=200):
    """BASELINE.json configs[0] (the reference's own CPU-runnable case: 100 diploid x 500 biallelic loci, -a -k 3, plain EM) on
    the path a user's file takes: STRUCTURE text -> mc_read_structure (host/mc_reader.c) -> device layouts -> random
    initialisation from the seed -> em() with -T `iters` (iters + 1 iterations, em_alg.c:150), the EM loop timed.  The drop-in
    command line then runs the same file, seed and arguments as a child process; its printed log likelihood and iteration count
    must be the timed fit's."""
    import re
    import tempfile
    from multiclust_amd import host
    w = WORKLOADS["c1"]
    ua, geno = workload_data(w, env)
    # every allele of a locus must occur for the reader's allele list (ascending codes seen in the file) to be the generator's
    with tempfile.TemporaryDirectory(prefix="mcc1.") as d:
        path = os.path.join(d, "c1.stru")
        write_structure(path, ua, geno)
        t0 = time.perf_counter()
        rc, dat = host.read_structure(path, ploidy=w["ploidy"])
        read_s = time.perf_counter() - t0
        if rc:
            raise SystemExit("c1: mc_read_structure failed (%d)" % rc)
        fit = host.Fit(dat["ua"], dat["geno"], w["K"], device=env.local_rank, admixture=1, accel_scheme=0, verbosity=1, max_iter=iters,
                       abs_error=1e-300)
        for timed in (0, 1):                      # the first fit is the warm-up (graph capture, first-touch allocations)
            fit.reset()
            fit.initialize(SEED)
            env.barrier(C.c_void_p(fit.mod.dev))
            t0 = time.perf_counter()
            fit.em()
            dt = time.perf_counter() - t0
        n_iter, logL = fit.mod.n_iter, fit.mod.logL
        fit.close()
        exe = os.path.join(ROOT, "multiclust_amd", "bin", "multiclust")
        cli = {"ran": False}
        if os.access(exe, os.X_OK):
            t0 = time.perf_counter()
            res = subprocess.run([exe, "-f", path, "-d", os.path.join(d, ""), "-a", "-k", str(w["K"]), "-n", "1", "-T", str(iters), "-r", str(SEED),
                                  "-E", "1e-300"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
            m = re.search(r"initialization = 0: (-?\d+\.\d+) \(.*?\) in\s+(\d+) iterations", res.stdout)
            cli = {"ran": True, "rc": res.returncode, "wall_s": time.perf_counter() - t0,
                   "logL": float(m.group(1)) if m else None, "n_iter": int(m.group(2)) if m else None}
            cli["agrees"] = bool(m and int(m.group(2)) == n_iter and abs(float(m.group(1)) - logL) <= 1e-6)
    same = bool(dat["geno"].shape == geno.shape and np.array_equal(dat["ua"], ua) and np.array_equal(dat["geno"], geno))
    return {"value": n_iter / dt, "ms_per_step": dt * 1e3 / n_iter, "steps": n_iter, "unit": "EM iterations/s",
            "config": {"workload": "c1: %s, through the STRUCTURE reader" % w["desc"], "I": w["I"], "L": w["L"], "T": int(dat["ua"].sum()),
                       "ploidy": w["ploidy"], "K": w["K"], "accel_scheme": 0, "max_iter": iters, "logL": logL,
                       "reader_s": read_s, "reader_returns_the_generated_genotype": same, "command_line": cli}}


# ------------------------------------------------------------------------------------------------ c4: initialisations sharded
def run_units(env, fit, w, T, n_units, cycles, warmup, with_roofline=True):
    """n_units random initialisations of one data set, unit u on rank u mod N; every unit = mc_fit_unit (stream jumped to the
    unit's first draw, device-side initialisation, em() with -T 2*cycles-1) and is timed whole."""
    from multiclust_amd import hip, host, shard
    hlib = hip.load()
    ctx = C.c_void_p(fit.mod.dev)
    accel = fit.opt.accel_scheme
    per_cycle = 2 if accel else 1
    fit.opt.abs_error = 1e-300
    if warmup > 0:                                    # untimed: graph capture, first-touch allocations of the init path
        fit.opt.max_iter = per_cycle * min(warmup, cycles) - 1
        fit.fit_unit(SEED, env.rank)
    fit.opt.max_iter = per_cycle * cycles - 1         # -T n runs n + 1 iterations (em_alg.c:150)
    mine = shard.units_for_rank(n_units, env.rank, env.world)
    env.barrier(ctx)
    if with_roofline:
        profile_begin(ctx)
    t0 = time.perf_counter()
    local = []
    for u in mine:
        r = fit.fit_unit(SEED, u)
        if r.fatal:
            raise SystemExit("unit %d: fatal=%d" % (u, r.fatal))
        local.append(r)
    results = shard.exchange(local, n_units, env.dist, env.cdev)      # the path's one exchange (all-reduce of disjoint rows)
    env.collectives += env.dist is not None
    env.barrier(ctx)
    dt = time.perf_counter() - t0
    dt = env.reduce([dt], "MAX")[0]
    summary = shard.replay(results, fit.opt, fit.no_parameters(), fit.I)
    total_iters = sum(r.n_iter for r in results)
    steps_on_busiest = max(1, len(shard.units_for_rank(n_units, 0, env.world))) * cycles
    out = {
        "value": total_iters / dt, "ms_per_step": dt * 1e3 / steps_on_busiest, "steps": cycles,
        "config": {"workload": "%s: %s" % ("c4" if w["I"] == 10000 else "c4s", w["desc"]), "I": w["I"], "L": w["L"], "T": T, "ploidy": w["ploidy"], "K": w["K"],
                   "accel_scheme": accel, "em_iterations_per_step": per_cycle, "units": n_units,
                   "unit_to_rank": "u mod %d" % env.world, "max_iter": fit.opt.max_iter,
                   "timed": "rand() jump-ahead + device-side initialisation + em() of every unit + the all-reduce",
                   "time_to_finish_s": dt, "units_per_s": n_units / dt, "em_iterations": total_iters,
                   "best_unit": summary.best_unit, "best_logL": summary.max_logL,
                   "unit_logL": [r.logL for r in results]},
    }
    if with_roofline and env.rank == 0:
        nnz, _ = data_counts(ctx)
        out["roofline"] = roofline_object(ctx, w, T, w["K"], total_iters / dt / env.world, "c4", nnz, len(mine) * cycles)
    elif with_roofline:
        hlib.mchip_profile_end(ctx, None, None, None)
    return out


# ------------------------------------------------------------------------------------------------ c5: bootstrap replicates sharded
def run_bootstrap(env, w, ua, geno, n_rep, budget, n_init=1, n_streams=1):
    """-b n_rep of K-1 vs K (multiclust.c:675-708): the observed-data fits (untimed setup, every rank computes the same bits),
    then replicate b on rank b mod N = mc_fit_replicate, timed whole."""
    from multiclust_amd import host
    lib = host.load()
    K1, K0 = w["K"], w["K"] - 1
    T = int(ua.sum())
    opts = dict(admixture=1, accel_scheme=0, verbosity=1, abs_error=1e-300, max_iter=budget)
    obs = {}
    for K in (K0, K1):                                  # estimate_model on the observed data: H0 then HA
        fit = host.Fit(ua, geno, K, device=env.local_rank, **opts)
        rng = host.McRng()
        lib.mc_srand(C.byref(rng), SEED)
        lib.mc_rng_jump(C.byref(rng), (0 if K == K0 else n_init) * lib.mc_draws_per_init(C.byref(fit.opt), C.byref(fit.dat), K))
        best = None
        for u in range(n_init):
            fit.reset()
            if lib.mc_initialize_model(C.byref(fit.opt), C.byref(fit.dat), fit.mp, C.byref(rng)):
                raise SystemExit("mc_initialize_model failed")
            fit.em()
            if best is None or fit.mod.logL > best[0]:
                best = (fit.mod.logL, fit.get_q(fit.mod.pindex), fit.get_p(fit.mod.pindex))
        obs[K] = best
        opt, dat, keep = fit.opt, fit.dat, (fit.ua, fit.geno)
        per_init = lib.mc_draws_per_init(C.byref(fit.opt), C.byref(fit.dat), K)
        nnz, _ = data_counts(C.c_void_p(fit.mod.dev))
        fit.close()
    ts_obs = obs[K1][0] - obs[K0][0]
    mle_q, mle_p = np.ascontiguousarray(obs[K0][1]), np.ascontiguousarray(obs[K0][2])
    base = host.McRng()
    lib.mc_srand(C.byref(base), SEED)
    lib.mc_rng_jump(C.byref(base), 2 * n_init * per_init)     # the serial stream after the observed-data fits
    mine = list(range(env.rank, n_rep, env.world))
    env.barrier()
    t0 = time.perf_counter()
    rows = np.zeros((n_rep, 5))

    def worker(x):
        """one of this rank's n_streams workers (host thread + its own K-1 and K models, i.e. contexts and streams, re-used
        by every replicate it fits): a replicate's data-set generation and initialisation are latency-bound and overlap with
        another worker's FP64-bound EM kernels -- the command line's --streams"""
        models = (C.POINTER(host.McModel) * 2)()
        for b in mine[x::n_streams]:
            r = host.McReplicateResult()
            rc = lib.mc_fit_replicate(C.byref(opt), C.byref(dat), env.local_rank, C.byref(base), b, K0, K1, n_init, K0,
                                      mle_q.ctypes.data, mle_p.ctypes.data, C.byref(r), models)
            if rc or r.fatal:
                raise SystemExit("replicate %d: rc=%d fatal=%d" % (b, rc, r.fatal))
            rows[b] = (r.ts, r.logL_H0, r.logL_HA, r.n_iter, 1.0)
            if os.environ.get("MC_TIMING"):
                sys.stderr.write("replicate %d done %.3f s after the start\n" % (b, time.perf_counter() - t0))
        for mp in models:
            if mp:
                lib.mc_model_free(mp)

    if n_streams > 1:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(n_streams) as pool:
            list(pool.map(worker, range(n_streams)))         # ctypes releases the GIL inside the library
    else:
        worker(0)
    flat = env.reduce(rows.ravel().tolist(), "SUM")         # the one exchange: disjoint rows
    rows = np.array(flat).reshape(n_rep, 5)
    env.barrier()
    dt = time.perf_counter() - t0
    dt = env.reduce([dt], "MAX")[0]
    if not np.all(rows[:, 4] == 1.0):
        raise SystemExit("a replicate was fitted %s times" % rows[:, 4].tolist())
    total_iters = float(rows[:, 3].sum())
    n_ge = int(np.sum(rows[:, 0] >= ts_obs))
    del keep
    return {
        "value": total_iters / dt, "ms_per_step": dt * 1e3 / max(1, len(range(0, n_rep, env.world))), "steps": n_rep,
        "config": {"workload": "%s: %s" % ("c5" if w["I"] == 5000 else "c5s", w["desc"]), "I": w["I"], "L": w["L"], "T": T, "ploidy": w["ploidy"], "K": [K0, K1],
                   "accel_scheme": 0, "replicates": n_rep, "replicate_to_rank": "b mod %d" % env.world, "n_init": n_init,
                   "streams_per_gpu": n_streams,
                   "max_iter": budget, "step": "one bootstrap replicate",
                   "timed": "device-side generation of every replicate + initialisation + em() of both models + the all-reduce",
                   "time_to_finish_s": dt, "replicates_per_s": n_rep / dt, "em_iterations": total_iters,
                   "ts_obs": ts_obs, "replicates_with_ts_ge_obs": n_ge, "p_value": n_ge / n_rep,
                   "ts_first": rows[:8, 0].tolist(), "ts_checksum": float(rows[:, 0].sum()),
                   "nonempty_cells_observed": nnz},
    }


# ------------------------------------------------------------------------------------------------ main
_REAL_STDOUT = None


def keep_stdout_for_the_result_line():
    """stdout carries ONE JSON line.  Libraries write there too (RCCL prints a five-line version banner on stdout when torch
    creates the first communicator -- seen on the GPU box with world size 1), so a rank's file descriptor 1 points at stderr
    from here on and the result line goes to the descriptor saved here."""
    global _REAL_STDOUT
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)


def emit(text):
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, (text + "\n").encode())


def finish(env, args, out, scaling):
    line = {"metric": "EM iterations/sec, IxLxK admixture", "value": out["value"], "unit": "EM iterations/s", "n_gpus": env.world,
            "steps": out["steps"], "warmup": args.warmup, "ms_per_step": out["ms_per_step"], "higher_is_better": True,
            "scaling": scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": out["config"]}
    # what the exchanges of this run went through: "none" (one rank, nothing to exchange), else the torch.distributed backend
    # (nccl = RCCL), the collectives issued on it and the data-set checksum every rank agreed on before the timed region
    line["exchange"] = {"backend": env.backend if env.dist is not None else "none", "collectives": env.collectives,
                        "data_crc32_all_ranks": WORKLOADS[args.workload].get("data_crc32")}
    for k in ("roofline", "cpu_baseline", "cpu_baseline_degraded", "stability", "secondary"):
        if k in out:
            line[k] = out[k]
    emit(json.dumps(line))


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--cpu-worker":      # child of the cpu_baseline leg: CPU only
        cpu_worker(sys.argv[2])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3", choices=sorted(WORKLOADS))
    ap.add_argument("--accel", type=int, default=None, help="override the workload's acceleration scheme (0..6)")
    ap.add_argument("--units", type=int, default=50, help="c4: random initialisations sharded over the GPUs")
    ap.add_argument("--replicates", type=int, default=200, help="c5: bootstrap replicates sharded over the GPUs")
    ap.add_argument("--streams", type=int, default=2, help="c5: concurrent replicates per GPU (host thread + contexts + streams each)")
    ap.add_argument("--settle", type=float, default=1.5, help="seconds of untimed log-likelihood passes before the warm-up steps of a "
                    "single-fit workload, so that the timed steps run at the clocks of a busy device (0: none)")
    ap.add_argument("--stability", type=int, default=5, help="single-fit workloads: repeat the timed batch this many times behind the "
                    "timed region and report min / median / max EM iterations/s (0: off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=20.0, help="seconds of one host core for the CPU oracle's sample")
    ap.add_argument("--ref-budget", type=float, default=15.0, help="seconds of one host core for the reference's own em() on the "
                    "head of that sample (oracle/_ref/ref_time; 0: skip)")
    ap.add_argument("--no-secondary", action="store_true", help="skip the extra workloads carried on the default line (profiling "
                    "runs: their kernels have the same names as the headline workload's)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if args.gpus > 1 and "RANK" not in os.environ:
        launch_ranks(args.gpus)                        # does not return

    keep_stdout_for_the_result_line()
    env = Env(args.gpus)
    name = args.workload
    w = WORKLOADS[name]
    want_cpu = env.world == 1 and env.rank == 0 and not args.no_cpu_baseline

    if name in ("c5", "c5s"):
        ua, geno = workload_data(w, env)
        out = run_bootstrap(env, w, ua, geno, args.replicates, args.steps, n_streams=args.streams)
        if want_cpu:
            attach_cpu_baseline(out, w, ua, geno, 0, args.cpu_budget, env.local_rank, units=args.replicates, ref_budget_s=args.ref_budget)
        if env.rank == 0:
            finish(env, args, out, "strong")
        env.close()
        return

    ua, geno = workload_data(w, env)
    T = int(ua.sum())
    if name in ("c4", "c4s"):
        from multiclust_amd import host
        accel = w["accel"] if args.accel is None else args.accel
        fit = host.Fit(ua, geno, w["K"], device=env.local_rank, admixture=1, accel_scheme=accel, verbosity=1, abs_error=1e-300)
        out = run_units(env, fit, w, T, args.units, args.steps, args.warmup)
        fit.close()
        if want_cpu:
            attach_cpu_baseline(out, w, ua, geno, accel, args.cpu_budget, env.local_rank, units=args.units, ref_budget_s=args.ref_budget)
        if env.rank == 0:
            finish(env, args, out, "strong")
        env.close()
        return

    out, fit, accel = run_single_fit(args, env, name, ua, geno, args.steps, args.warmup)
    if want_cpu:
        attach_cpu_baseline(out, w, ua, geno, accel, args.cpu_budget, env.local_rank, ref_budget_s=args.ref_budget)
    # the other BASELINE.json configurations on the same line (plain numbers, same measurement rules): configs[3] (c4: 50
    # initialisations sharded) and configs[4] (c5: 200 bootstrap replicates sharded) at every N; configs[1] (c2) and configs[0]
    # (c1, through the STRUCTURE reader) at N = 1
    if name == "c3" and not args.no_secondary:
        sec = {}
        c4 = run_units(env, fit, WORKLOADS["c4"], T, args.units, args.steps, 0, with_roofline=False)
        sec["c4"] = {k: c4[k] for k in ("value", "ms_per_step", "steps", "config")}
        sec["c4"].update(unit="EM iterations/s", scaling="strong")
        fit.close()
        del geno
        if env.world == 1:
            w2 = WORKLOADS["c2"]
            ua2, geno2 = workload_data(w2, env)
            c2, fit2, _ = run_single_fit(args, env, "c2", ua2, geno2, 300, 5, with_roofline=False)
            sec["c2"] = dict(c2, unit="EM iterations/s")
            # its roofline from a second, event-instrumented batch (the timed one above runs as one captured graph; with an
            # event pair around every pass of a 0.15 ms step the same steps read 10-25 % slower, so the two are kept apart)
            from multiclust_amd import hip
            ctx2 = C.c_void_p(fit2.mod.dev)
            profile_begin(ctx2)
            st = hip.RunState(logL=fit2.mod.logL, abs_error=1e-300, n_iter=fit2.mod.n_iter)
            hip.load().mchip_em_run(ctx2, 0, 100, C.byref(st))
            nnz2, _ = data_counts(ctx2)
            sec["c2"]["roofline"] = roofline_object(ctx2, w2, int(ua2.sum()), w2["K"], c2["value"], "c2", nnz2, 100)
            sec["c2"]["roofline"]["measured_in"] = "a separate batch of 100 EM steps with HIP events around every pass"
            fit2.close()
            del geno2
            # configs[4] and configs[0] at N = 1 too: the anchor of c5's strong-scaling curve, and the parity configuration
            w5 = WORKLOADS["c5"]
            ua5, geno5 = workload_data(w5, env)
            c5 = run_bootstrap(env, w5, ua5, geno5, args.replicates, args.steps, n_streams=args.streams)
            sec["c5"] = dict(c5, unit="EM iterations/s", scaling="strong")
            del geno5
            sec["c1"] = run_c1_reader(env)
        else:
            w5 = WORKLOADS["c5"]
            ua5, geno5 = workload_data(w5, env)
            c5 = run_bootstrap(env, w5, ua5, geno5, args.replicates, args.steps, n_streams=args.streams)
            sec["c5"] = dict(c5, unit="EM iterations/s", scaling="strong")
        out["secondary"] = sec
    else:
        fit.close()
    if env.rank == 0:
        finish(env, args, out, "weak")
    env.close()


if __name__ == "__main__":
    main()
