"""-m gpu: the sharded workloads of bench.py (c4: initialisations u mod N; c5: bootstrap replicates b mod N) on small data
sets, against the drop-in command line run on the same file with the same seed: per-unit log likelihoods and per-replicate
test statistics must agree (the command line prints them with six decimals)."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from procutil import run_program

from golden_util import GOLD, Golden

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "multiclust_amd", "bin", "multiclust")
sys.path.insert(0, ROOT)


class OneRank:
    """bench.Env for a single rank without a process group"""
    world, rank, local_rank, dist, cdev, collectives = 1, 0, 0, None, "cpu", 0

    def barrier(self, ctx=None):
        pass

    def reduce(self, values, op):
        return list(values)


def test_c4_units_agree_with_command_line(tmp_path):
    import bench
    from multiclust_amd import host
    g = Golden("multi_admix_k4")
    n_units, cycles = 6, 8
    res = run_program([BIN, "-f", os.path.join(GOLD, "data", "multi.stru"), "-d", str(tmp_path), "-a", "-k", "4", "-n", str(n_units),
                       "-s", "3", "-T", str(2 * cycles - 1), "-r", str(bench.SEED)], timeout=300)
    assert res.returncode == 0, res.stderr
    cli = [(float(m.group(1)), int(m.group(2))) for m in
           re.finditer(r"initialization = \d+: (-?\d+\.\d+) \(.*?\) in\s+(\d+) iterations", res.stdout)]
    assert len(cli) == n_units
    fit = host.Fit(g.ua, g.geno, 4, admixture=1, accel_scheme=3, verbosity=1, abs_error=1e-300)
    w = dict(I=g.I, L=g.L, ploidy=g.ploidy, K=4, desc="multi.stru")
    out = bench.run_units(OneRank(), fit, w, g.T, n_units, cycles, 1, with_roofline=False)
    fit.close()
    cfg = out["config"]
    assert cfg["units"] == n_units and len(cfg["unit_logL"]) == n_units
    assert cfg["em_iterations"] == sum(n for _, n in cli) == n_units * 2 * cycles
    for (ll, _), got in zip(cli, cfg["unit_logL"]):
        assert abs(ll - got) <= 1e-6, (ll, got)
    assert abs(cfg["best_logL"] - max(ll for ll, _ in cli)) <= 1e-6


def test_c5_replicates_agree_with_command_line(tmp_path):
    import bench
    g = Golden("tetra_admix_k3")
    n_rep, budget = 4, 9
    res = run_program([BIN, "-f", os.path.join(GOLD, "data", "tetra.stru"), "-d", str(tmp_path), "-p", "4", "-a", "-k", "3", "-n", "1",
                       "-b", str(n_rep), "-T", str(budget), "-r", str(bench.SEED)], timeout=300)
    assert res.returncode == 0, res.stderr
    cli = [(float(a), float(b)) for a, b in re.findall(r"test statistics bs=(-?\d+\.\d+) obs=(-?\d+\.\d+)", res.stdout)]
    assert len(cli) == n_rep
    w = dict(I=g.I, L=g.L, ploidy=4, K=3, desc="tetra.stru")
    out = bench.run_bootstrap(OneRank(), w, g.ua, g.geno, n_rep, budget, n_init=1)
    cfg = out["config"]
    assert cfg["replicates"] == n_rep and cfg["em_iterations"] == n_rep * 2 * (budget + 1)
    assert abs(cfg["ts_obs"] - cli[0][1]) <= 2e-6
    for (bs, _), got in zip(cli, cfg["ts_first"]):
        assert abs(bs - got) <= 2e-6, (bs, got)


def run_bench(args, env_extra, timeout=900, expect_failure=False):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, timeout=timeout)
    if expect_failure:
        return res
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.split("\n") if l.strip()]
    assert len(lines) == 1, res.stdout                       # exactly one JSON line on stdout
    import json
    return json.loads(lines[0])


@pytest.mark.parametrize("workload", ["c4s", "c5s"])
def test_two_ranks_give_the_one_rank_results(workload):
    """`bench.py --gpus 2` starts two ranks by itself (here both on device 0, exchanging over gloo: the rehearsal knobs) and
    prints n_gpus = 2; units / replicates are sharded u mod 2 and every one is accounted for exactly once; per-unit results
    are bit for bit those of the one-rank run (a unit starts where the serial rand() stream would be, whoever fits it)."""
    args = ["--workload", workload, "--no-cpu-baseline", "--steps", "6", "--units", "7", "--replicates", "5"]
    one = run_bench(args, {})
    two = run_bench(args + ["--gpus", "2"], {"MC_BENCH_DEVICE": "0", "MC_BENCH_BACKEND": "gloo"})
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "strong"
    c1, c2 = one["config"], two["config"]
    if workload == "c4s":
        assert c2["units"] == 7 and len(c2["unit_logL"]) == 7 and c2["em_iterations"] == 7 * 12
        assert c1["unit_logL"] == c2["unit_logL"] and c1["best_unit"] == c2["best_unit"] and c1["best_logL"] == c2["best_logL"]
    else:
        assert c2["replicates"] == 5 and c2["em_iterations"] == 5 * 2 * 7
        assert c1["ts_first"] == c2["ts_first"] and c1["ts_obs"] == c2["ts_obs"] and c1["p_value"] == c2["p_value"]
    assert two["exchange"]["backend"] == "gloo" and two["exchange"]["collectives"] >= 4
    crc = two["exchange"]["data_crc32_all_ranks"]
    assert len(crc) == 2 and all(isinstance(c, int) for c in crc)       # allele counts and genotype: the same on both ranks
    assert one["exchange"] == {"backend": "none", "collectives": 0, "data_crc32_all_ranks": None}


@pytest.mark.parametrize("workload", ["c4s", "c5s", "c1"])
def test_one_rank_through_the_rccl_process_group(workload):
    """MC_BENCH_FORCE_PG=1: one rank, but the process group exists (backend nccl = RCCL, initialised on this device exactly as
    the driver's N > 1 launch initialises it) and the data-set check, the barriers and the run's exchange (all-reduce of the
    per-unit table / the test statistics / the best log likelihood) go through it.  Results are those of the run without it."""
    args = ["--workload", workload, "--no-cpu-baseline", "--no-secondary", "--steps", "6", "--units", "5", "--replicates", "4", "--settle", "0"]
    plain = run_bench(args, {})
    pg = run_bench(args, {"MC_BENCH_FORCE_PG": "1"})
    assert pg["n_gpus"] == 1 and pg["exchange"]["backend"] == "nccl" and pg["exchange"]["collectives"] >= 6
    assert len(pg["exchange"]["data_crc32_all_ranks"]) == 2
    for key in ("unit_logL", "best_logL", "ts_first", "ts_obs", "em_iterations"):
        if key in plain["config"]:
            assert plain["config"][key] == pg["config"][key], key


def test_ranks_with_different_data_sets_stop_before_the_timed_region():
    """every rank generates the synthetic data set on its own device; if two devices' generators ever disagreed the units of c4
    would be fits of different data.  One byte changed on rank 1: both ranks leave with an error, no JSON line."""
    res = run_bench(["--workload", "c4s", "--no-cpu-baseline", "--steps", "2", "--units", "4", "--gpus", "2"],
                    {"MC_BENCH_DEVICE": "0", "MC_BENCH_BACKEND": "gloo", "MC_BENCH_CORRUPT_RANK": "1"}, expect_failure=True)
    assert res.returncode != 0 and '{"metric"' not in res.stdout
    assert "the synthetic data set differs between ranks" in res.stderr


@pytest.mark.parametrize("workload,extra", [("c1", []), ("c4s", ["--units", "3"]), ("c5s", ["--replicates", "3"])])
def test_cpu_baseline_legs_on_small_workloads(workload, extra):
    """`cpu_baseline` leads with the reference's own em() (oracle/_ref/ref_time, kind "reference") where that binary is in the
    tree, the oracle ("port") under it; both carry the HIP path's distance from them on their own sample, inside BASELINE.json's
    tolerances."""
    out = run_bench(["--workload", workload, "--steps", "4", "--warmup", "1", "--settle", "0", "--stability", "0", "--no-secondary",
                     "--cpu-budget", "0.5", "--ref-budget", "0.5"] + extra, {})
    cb = out["cpu_baseline"]
    have_ref = os.access(os.path.join(ROOT, "oracle", "_ref", "ref_time"), os.X_OK)
    assert cb["kind"] == ("reference" if have_ref else "port")
    port = cb["port"] if have_ref else cb
    for leg in ([cb, port] if have_ref else [port]):
        assert leg["unit"] == "EM iterations/s" and leg["value"] > 0 and leg["cores"] >= 1 and leg["sample"]
        par = leg["parity"]
        # logL: 1e-8 absolute at config-1 scale; beyond it 1e-12 relative (the reference's own running sum
        # is no better than that at |logL| ~ 1e6)
        assert par["same_n_iter"] == 1 and (par["abs_dlogL"] <= 1e-8 or par["rel_dlogL"] <= 1e-12), par
        assert par["max_rel_dQ"] <= 1e-6 and par["max_rel_dP"] <= 1e-6, par
        assert abs(leg["gpu_over_cpu"] * leg["value"] - out["value"]) <= 1e-9 * out["value"]
    assert cb["reference_extrapolated"]["kind"] == "extrapolation"
    if have_ref:
        assert 1 < cb["ns_per_cell"] < 500        # (i, allele column, k) cell and iteration: 4-11 ns on a GPU box's host, 30-65 on the build container


def test_a_reference_binary_that_does_not_run_costs_the_line_nothing():
    """oracle/_ref/ref_time is built in one container and run in another: if it cannot run there, cpu_baseline falls back to
    the oracle and says why; the result line is still printed."""
    out = run_bench(["--workload", "c1", "--steps", "4", "--warmup", "1", "--settle", "0", "--stability", "0", "--no-secondary",
                     "--cpu-budget", "0.3", "--ref-budget", "0.3"], {"MC_BENCH_REF_TIME": "/bin/false"})
    cb = out["cpu_baseline"]
    assert cb["kind"] == "port" and cb["value"] > 0 and "the reference leg failed" in cb["reference"]
    assert out["value"] > 0

