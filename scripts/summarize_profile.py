#!/usr/bin/env python3
"""Condense rocprofv3 output (gpurun_out/...) into the small summaries committed under profiles/.

  python scripts/summarize_profile.py <tag> <kernel-stats-dir> [<pmc-fetch-dir> <pmc-write-dir>]

Writes profiles/<tag>_kernel_stats.csv (library kernels only, from `rocprofv3 --kernel-trace --stats`) and
profiles/<tag>_traffic.json (per kernel: FETCH_SIZE / WRITE_SIZE averaged per launch, from two separate --pmc
passes, corrected as MI355X_MICROARCH.md section HBM prescribes: both counters are in KiB; on gfx950 FETCH_SIZE
reports half the bytes of a wide coalesced read stream, so it is doubled)."""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def ours(name):
    return "at::" not in name and "rocclr" not in name and ("k_" in name)


def build_id():
    """sha256 of the HIP library in the tree (what the profiled command loaded) and, where the tree is a git checkout (the build
    container: the GPU box gets a snapshot without .git), the commit"""
    import hashlib
    import subprocess
    lib = os.path.join(ROOT, "multiclust_amd", "lib", "libmulticlust_hip.so")
    out = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest() if os.path.exists(lib) else None, "commit": None}
    try:
        res = subprocess.run(["git", "-C", ROOT, "rev-parse", "HEAD"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        if res.returncode == 0:
            out["commit"] = res.stdout.strip()
            dirty = subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "multiclust_amd", "include", "Makefile"],
                                   stdout=subprocess.PIPE, text=True).stdout.strip()
            out["tree_dirty"] = bool(dirty)
    except OSError:
        pass
    return out


def stamp_commit(path):
    """`summarize_profile.py --stamp profiles/<file>_traffic.json`: run in the build container on a summary that came back from the
    GPU box, adds the commit to its _build record when the library in this tree is the one that was profiled"""
    d = json.load(open(path))
    here = build_id()
    if d.get("_build", {}).get("library_sha256") != here["library_sha256"]:
        raise SystemExit("%s describes another build of the library than the one in this tree" % path)
    d["_build"].update(commit=here["commit"], tree_dirty=here.get("tree_dirty"))
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print("stamped", path, here["commit"])


def main():
    if len(sys.argv) == 3 and sys.argv[1] == "--stamp":
        stamp_commit(sys.argv[2])
        return
    tag, stats_dir = sys.argv[1], sys.argv[2]
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    f = glob.glob(os.path.join(stats_dir, "**", "*_kernel_stats.csv"), recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if ours(r["Name"])]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    out = os.path.join(ROOT, "profiles", tag + "_kernel_stats.csv")
    with open(out, "w") as g:
        g.write("kernel,calls,total_ms,avg_ms,min_ms,max_ms,share_of_library_time\n")
        for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
            g.write("%s,%s,%.3f,%.4f,%.4f,%.4f,%.3f\n" % (short(r["Name"]), r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                       float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6,
                                                       float(r["MaxNs"]) / 1e6, float(r["TotalDurationNs"]) / tot))
    print("wrote", out)
    # launches that did a pass over the data: batched runs leave no-op launches behind (kernels that return at once after
    # the stopping rule fired, or because their sums were already in place); per-dispatch durations tell them apart
    tr = glob.glob(os.path.join(stats_dir, "**", "*_kernel_trace.csv"), recursive=True)
    if tr:
        dur = collections.defaultdict(list)
        for r in csv.DictReader(open(tr[0])):
            if ours(r["Kernel_Name"]):
                dur[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
        out2 = os.path.join(ROOT, "profiles", tag + "_kernel_stats_working_launches.csv")
        with open(out2, "w") as g:
            g.write("kernel,launches,working_launches,avg_ms_of_working_launches\n")
            for k in sorted(dur, key=lambda k: -sum(dur[k])):
                big = [d for d in dur[k] if d >= 0.05 * max(dur[k])]
                g.write("%s,%d,%d,%.4f\n" % (k, len(dur[k]), len(big), sum(big) / len(big)))
        print("wrote", out2)
    if len(sys.argv) >= 5:
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for d in sys.argv[3:5]:
            for f in glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if ours(r["Kernel_Name"]):
                        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        res = {}
        for k, c in agg.items():
            # launches that return at once (see above) move no data: keep the dispatches above 5 % of the kernel's largest
            for name in ("FETCH_SIZE", "WRITE_SIZE"):
                if c.get(name):
                    c[name] = [v for v in c[name] if v >= 0.05 * max(c[name])]
            fetch = sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) if c.get("FETCH_SIZE") else None
            write = sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if c.get("WRITE_SIZE") else None
            res[k] = {
                "launches_sampled": len(c.get("FETCH_SIZE", [])), "FETCH_SIZE_KiB_per_launch": fetch,
                "WRITE_SIZE_KiB_per_launch": write,
                "hbm_bytes_per_launch_corrected": (2 * fetch * 1024 if fetch is not None else 0) + (write * 1024 if write is not None else 0),
            }
        # which build these counters describe: bench.py compares the digest with the library it has loaded (`traffic_build_matches`)
        res["_build"] = build_id()
        out = os.path.join(ROOT, "profiles", tag + "_traffic.json")
        json.dump(res, open(out, "w"), indent=1, sort_keys=True)
        print("wrote", out)


if __name__ == "__main__":
    main()
