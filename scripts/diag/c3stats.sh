# rocprofv3 kernel-trace statistics of the c3 bench only (no counters): per-kernel average times; usage: c3stats.sh <out.csv>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o v -- python3 bench.py --steps 20 --warmup 3 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/c3stats_trace.log 2>&1 &&
python3 - "$1" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/prof/**/*kernel_stats.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
with open(sys.argv[1], "w") as out:
    out.write("kernel,calls,total_ms,avg_ms\n")
    for r in rows:
        name = r["Name"].split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")
        out.write("%s,%s,%.3f,%.4f\n" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
PY
