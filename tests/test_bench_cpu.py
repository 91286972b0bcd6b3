"""CPU checks of bench.py's launcher: --gpus N without RANK must start N ranks as a child (never run one rank and print
n_gpus: 1), and must refuse when fewer than N devices are visible."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_gpus_flag_needs_that_many_devices():
    """`--gpus 99` (no node has 99 devices): exit code 3, nothing on stdout, and no rank was started -- the check counts
    devices with torch.cuda.device_count(), which does not initialise the GPU on this image, before the child is launched."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MC_BENCH_DEVICE")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "99"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode == 3 and "--gpus 99 but only" in res.stderr and res.stdout == ""
    assert "torch.distributed" not in res.stderr and "Traceback" not in res.stderr


def test_world_size_must_match_gpus_flag():
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, stdout=subprocess.PIPE,
                         stderr=subprocess.PIPE, text=True, timeout=300)
    assert res.returncode != 0 and "WORLD_SIZE" in res.stderr and res.stdout == ""


def test_roofline_object_names_the_longest_pass_and_what_binds_it():
    """bench.build_roofline (pure: the numbers of BENCH_r02's run): the dominant kernel is the pass with the most time per step
    over ALL kinds (the dual pass there, not the column pass), `bound` says FP64 issue, the HBM fraction stays the contract's
    algorithmic-bytes figure, and the traffic figure names the file it came from."""
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["c3"])
    T, K, steps = 300302, 8, 20
    kernel_ms = [2.142 * 40, 1.77 * 20, 0.0, 2.78 * 20]         # column pass twice per cycle, S-side pass once, dual pass once
    launches = [40, 20, 0, 20]
    table = {"k_column_counts<2, false, false>": {"hbm_bytes_per_launch_corrected": 1.32e9},
             "k_individual_sparse_w<2, true, false, true, true>": {"hbm_bytes_per_launch_corrected": 2.72e9}}
    r = bench.build_roofline(w, T, K, kernel_ms, launches, steps, 216.0, 1523000000, table, "profiles/r02_v6_c3_traffic.json")
    for key in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes_per_launch",
                "avg_launch_ms", "kernels_ms", "kernels_ms_per_step", "launches", "pass_hbm_frac", "fp64_valu_frac",
                "fp64_valu_frac_dense_cells", "fp64_valu_frac_of_sustained", "bound_note"):
        assert key in r, key
    assert r["bound"] == "fp64-valu" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    # per step: column 2 x 2.142 = 4.284 ms > dual 2.78 > S-side 1.77: the column pass dominates the step, by time per step
    assert r["kernel"].startswith("k_column_counts") and abs(r["avg_launch_ms"] - 2.142) < 1e-9
    B = bench.algorithmic_bytes(w, T, K)["column_pass"]
    assert B == 10000 * 100000 * 2 + 16 * K * T + 8 * 10000 * K
    assert abs(r["achieved"] - B / 2.142e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
    assert r["traffic"] == 1.32e9 and r["traffic_source"] == "profiles/r02_v6_c3_traffic.json"
    # one column pass per step instead: the dual pass is the longest kernel of the step and is the one reported
    r2 = bench.build_roofline(w, T, K, [2.142 * 20, 1.77 * 20, 0.0, 2.78 * 20], [20, 20, 0, 20], steps, 216.0, 1523000000, table, "f.json")
    assert r2["kernel"].startswith("k_individual_sparse_w, dual") and r2["traffic"] == 2.72e9
    assert r2["algorithmic_bytes_per_launch"] == 10000 * 100000 * 2 + 16 * K * T + 24 * 10000 * K
    # no counter file: no traffic figure and no source
    r3 = bench.build_roofline(w, T, K, kernel_ms, launches, steps, 216.0, 1523000000)
    assert r3["traffic"] is None and r3["traffic_source"] is None


def test_bench_line_declares_the_new_objects():
    """keys a reader of the JSON line relies on are produced by bench.py's code paths (static check: a GPU is needed to run it)"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    for needle in ('"reference_extrapolated"', '"stability"', '"traffic_source"', '"exchange"', '"fp64-valu"', '"median"',
                   '"traffic_build_matches"', '"traffic_build"', '"cpu_baseline_degraded"',
                   # every BASELINE.json configuration on the driver's N = 1 line: c3 is the headline, the others ride as `secondary`
                   'sec["c4"]', 'sec["c2"]', 'sec["c5"]', 'sec["c1"]'):
        assert needle in src, needle


def test_traffic_figure_is_tied_to_the_build_it_was_taken_from():
    """roofline.traffic comes from a PMC summary of an earlier run; the summary names the library it profiled (sha256) and the line
    says whether that is the library this run has loaded"""
    sys.path.insert(0, ROOT)
    import bench
    w = dict(bench.WORKLOADS["c3"])
    table = {"k_column_counts<2, false, false>": {"hbm_bytes_per_launch_corrected": 1.3e9},
             "k_individual_sparse_w<2, true, false, true, true>": {"hbm_bytes_per_launch_corrected": 2.5e9},
             "_build": {"library_sha256": "abc", "commit": "deadbeef"}}
    args = (w, 300302, 8, [2.1 * 40, 1.7 * 20, 0.0, 2.8 * 20], [40, 20, 0, 20], 20, 216.0, 1523000000, table, "profiles/x_traffic.json")
    same, other = bench.build_roofline(*args, "abc"), bench.build_roofline(*args, "abd")
    assert same["traffic"] == 1.3e9 and same["traffic_build_matches"] is True and same["traffic_build"]["commit"] == "deadbeef"
    assert other["traffic"] == 1.3e9 and other["traffic_build_matches"] is False
    assert same["pass_traffic"] == {"column_pass": 1.3e9, "individual_dual_pass": 2.5e9}
    none = bench.build_roofline(*args[:8])
    assert none["traffic"] is None and none["traffic_build_matches"] is None
    # the summaries the tree ships carry the record (those of this round on)
    import glob
    import json
    newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_c3_traffic.json")))[-1]
    if os.path.basename(newest) >= "r04":
        assert "library_sha256" in json.load(open(newest))["_build"]
