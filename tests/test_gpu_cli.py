"""-m gpu: the drop-in command line (multiclust_amd/bin/multiclust) end to end -- reader, initialisation from the
libc-compatible stream, EM on the GPU, bookkeeping, stdout lines and the five output files -- against the
reference's own binary run on the same files with the same arguments (tests/golden/cli_*)."""
import os
import re
import subprocess

import numpy as np
import pytest

from procutil import run_program

from golden_util import GOLD

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "multiclust_amd", "bin", "multiclust")
NUM = re.compile(r"-?\d+\.\d+(?:e[+-]?\d+)?|-?\d+")
CLOCK = re.compile(r"\d\d:\d\d:\d\d")       # elapsed CPU time: not comparable between machines


def run_cli(case, tmp_path, extra=()):
    gdir = os.path.join(GOLD, "cli_" + case)
    args = open(os.path.join(gdir, "ARGS.txt")).read().split()
    stru = os.path.join(GOLD, "data", args[1])
    rest = [os.path.join(GOLD, "data", a) if os.path.exists(os.path.join(GOLD, "data", a)) else a for a in args[2:]]   # -P / -Q files
    # -d with its trailing slash: the reference's own mixture writers size their file-name buffers without the separator they
    # insert when it is missing (write_file.c:628-640,707-711), one byte short; tests/test_gpu_refbind.py runs that code
    cmd = [BIN, "-f", stru, "-d", os.path.join(str(tmp_path), "")] + rest + list(extra)
    res = run_program(cmd, timeout=300)
    assert res.returncode == 0, res.stderr
    run_cli.last_stderr = res.stderr
    return gdir, CLOCK.sub("HH:MM:SS", res.stdout.replace(stru, os.path.basename(stru)))


def numeric_rows(path):
    rows = []
    for line in open(path):
        toks = line.replace(":", " ").split()
        rows.append(toks)
    return rows


def compare_file(ref, got, atol):
    a, b = numeric_rows(ref), numeric_rows(got)
    assert len(a) == len(b), (ref, len(a), len(b))
    for ra, rb in zip(a, b):
        assert len(ra) == len(rb), (ref, ra, rb)
        for x, y in zip(ra, rb):
            try:
                fx, fy = float(x), float(y)
            except ValueError:
                assert x == y, (ref, ra, rb)
                continue
            assert abs(fx - fy) <= atol, (ref, ra, rb)


@pytest.mark.parametrize("case,atol", [
    ("multi_admix_k4", 2e-6),            # plain EM: same iteration counts, same 6-decimal files
    ("multi_mix_k3", 2e-6),
    ("multi_admix_c_k3", 2e-6),
    ("tetra_admix_k3", 2e-6),
    ("missing_admix_k3_s3", 5e-3),       # SQUAREM: the path may differ through accept ties; converged fit within tolerance
    ("multi_admix_k4_i1000", 2e-6),      # -i beyond convergence: em() returns from the warm-up loop (em_alg.c:73-74)
    ("multi_admix_k4_i1000_T5", 2e-6),   # -i beyond the -T cap: one more EM step in the do/while (7 iterations, not 6)
    ("multi_admix_k4_i1000_T5_s3", 2e-6),  # ... one accelerated cycle that stops inside em_2_steps
    ("missing_admix_k3_noproj", 2e-6),   # --projection: the phantom allele slots of loci with missing data keep p = 0
    ("c1_admix_k3_PQ", 2e-6),            # -P / -Q: initial parameters from files (read_file.c:880-959)
    ("c1_admix_k3_PQ_s3", 5e-3),
    ("allmiss_admix_k2", 2e-6),          # three loci at which every individual is missing (uniquealleles = 0: no column, no file row)
])
def test_cli_matches_reference_binary(case, atol, tmp_path):
    gdir, out = run_cli(case, tmp_path)
    ref_lines = CLOCK.sub("HH:MM:SS", open(os.path.join(gdir, "stdout.txt")).read()).strip().split("\n")
    got_lines = out.strip().split("\n")
    assert len(ref_lines) == len(got_lines), out
    exact = atol < 1e-4
    for r, g in zip(ref_lines, got_lines):
        rs, gs = NUM.sub("#", r), NUM.sub("#", g)
        assert rs == gs, (r, g)                                      # same text skeleton, field for field
        rn, gn = [float(x) for x in NUM.findall(r)], [float(x) for x in NUM.findall(g)]
        for idx, (x, y) in enumerate(zip(rn, gn)):
            if x == int(x) and abs(x) < 1e6 and "." not in NUM.findall(r)[idx]:
                if exact:
                    assert x == y, (r, g)                            # iteration counts, K, seed, bookkeeping
            else:
                assert abs(x - y) <= max(atol * 10, 1e-6 * abs(x)) + (0 if exact else 5e-2), (r, g)
    files = sorted(f for f in os.listdir(gdir) if f not in ("stdout.txt", "ARGS.txt"))
    assert len(files) == 5
    for fn in files:
        assert os.path.exists(tmp_path / fn), fn
        compare_file(os.path.join(gdir, fn), tmp_path / fn, atol if not fn.endswith("out.txt") else max(atol * 10, 5e-2 if not exact else 1e-5))


def test_cli_k_range_and_quiet_mode(tmp_path):
    """-1/-2 K range (the reference itself aborts after the first K, so there is no golden: structural checks),
    -M prints only the maximum log likelihood."""
    stru = os.path.join(GOLD, "data", "multi.stru")
    res = run_program([BIN, "-f", stru, "-a", "-1", "2", "-2", "4", "-n", "2", "-r", "3", "-s", "3", "-d", str(tmp_path)], timeout=300)
    assert res.returncode == 0, res.stderr
    lines = res.stdout.strip().split("\n")
    assert len(lines) == 3 * (2 + 1)
    lls = [float(l.split()[9]) for l in lines if l.startswith(stru)]
    assert lls[0] < lls[1] < lls[2]                                    # more clusters fit better
    for K in (2, 3, 4):
        assert os.path.exists(tmp_path / ("multi.stru.admix.K=%d.pklm.txt" % K))
    res = run_program([BIN, "-f", stru, "-a", "-k", "2", "-n", "1", "-M", "-d", str(tmp_path)], timeout=300)
    # -M sets verbosity to SILENT (= 1, still non-zero), so the reference prints the summary line and then max_logL
    out = res.stdout.strip().split("\n")
    assert res.returncode == 0 and len(out) == 2
    assert abs(float(out[1]) - float(out[0].split()[9])) < 1e-6


def test_cli_bootstrap_device_host_and_sharded_agree(tmp_path, monkeypatch):
    """-b: no committed golden (on the data with missing values first tried, the reference's own binary aborts in its second
    model, "free(): invalid pointer"; on data without, it runs, and tests/test_gpu_cli_differential.py compares whole -b runs
    with it live); the replicate generator itself is pinned to the reference's parametric_bootstrap() in
    tests/test_bootstrap_cpu.py / test_gpu_bootstrap.py.  Here: the run completes, and three ways of running it print
    the same thing: replicates generated on the device, replicates drawn on the host and uploaded, and whole replicates
    sharded over devices (rehearsed with one device: worker thread, rand() jump-ahead per replicate, captured stdout,
    RCCL all-reduce of the test statistics, in-order replay)."""
    stru = os.path.join(GOLD, "data", "multi.stru")
    cmd = [BIN, "-f", stru, "-a", "-k", "3", "-n", "2", "-b", "3", "-r", "9", "-s", "3", "-d", str(tmp_path)]
    outs = []
    for mode in ("device", "host", "sharded"):
        monkeypatch.delenv("MC_HOST_BOOTSTRAP", raising=False)
        monkeypatch.delenv("MC_FORCE_SHARDED", raising=False)
        extra = []
        if mode == "host":
            monkeypatch.setenv("MC_HOST_BOOTSTRAP", "1")
        if mode == "sharded":
            monkeypatch.setenv("MC_FORCE_SHARDED", "1")
            monkeypatch.setenv("MC_TRACE_EXCHANGE", "1")
            extra = ["--gpus", "1"]
        res = run_program(cmd + extra, timeout=600)
        assert res.returncode == 0, res.stderr
        if mode == "sharded":
            # H0 and HA fits of the observed data (one all-reduce each), then the replicates' test statistics: one communicator,
            # three exchanges through RCCL
            trace = [l for l in res.stderr.split("\n") if l.startswith("exchange: RCCL")]
            assert [t.split("#")[1].split(",")[0] for t in trace] == ["1", "2", "3"], res.stderr
            assert "bootstrap test statistics" in trace[2] and "6 doubles" in trace[2]
        assert res.stdout.count("Bootstrap dataset") == 3 and "p-value to reject H0: K=2" in res.stdout
        text = CLOCK.sub("HH:MM:SS", res.stdout)
        outs.append(text[text.index("Bootstrap dataset 1"):])
    assert outs[0] == outs[1], outs
    assert outs[0] == outs[2], outs


@pytest.mark.parametrize("case,streams", [("multi_admix_k4", 1), ("tetra_admix_k3", 1), ("multi_admix_k4", 3)])
def test_cli_sharded_path_on_one_gpu_reproduces_serial_reference(case, streams, tmp_path, monkeypatch):
    """The --gpus machinery (host thread per device, unit = initialisation, rand() jump-ahead, RCCL all-reduce of the
    result table, serial-order bookkeeping replay, winner's owner writes the files) rehearsed with one device:
    stdout and files must equal the reference's serial run."""
    monkeypatch.setenv("MC_FORCE_SHARDED", "1")
    monkeypatch.setenv("MC_TRACE_EXCHANGE", "1")
    # streams > 1: several workers (host thread + context + stream each) share the one GPU
    gdir, out = run_cli(case, tmp_path, extra=["--gpus", "1", "--streams", str(streams)])
    # the exchange really went through get_comm() -> ncclCommInitAll -> mchip_comm_all_reduce (one per K fitted), not past it
    trace = [l for l in run_cli.last_stderr.split("\n") if l.startswith("exchange: RCCL")]
    assert len(trace) == 1 and "all-reduce #1" in trace[0] and "over 1 device(s)" in trace[0], run_cli.last_stderr
    assert int(trace[0].split()[2]) >= 20000 and "per-initialisation results" in trace[0]
    ref_lines = CLOCK.sub("HH:MM:SS", open(os.path.join(gdir, "stdout.txt")).read()).strip().split("\n")
    got_lines = out.strip().split("\n")
    assert len(ref_lines) == len(got_lines), out
    for r, g in zip(ref_lines, got_lines):
        assert NUM.sub("#", r) == NUM.sub("#", g), (r, g)
        rn, gn = [float(x) for x in NUM.findall(r)], [float(x) for x in NUM.findall(g)]
        for x, y in zip(rn, gn):
            assert abs(x - y) <= 2e-5 + 1e-9 * abs(x), (r, g)
    for fn in sorted(f for f in os.listdir(gdir) if f not in ("stdout.txt", "ARGS.txt")):
        compare_file(os.path.join(gdir, fn), tmp_path / fn, 2e-5)
