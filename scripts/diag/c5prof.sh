# per-kernel times of the config-5 bootstrap workload (one replicate in flight, 20 replicates): rocprofv3 kernel trace
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o c5 -- python3 bench.py --workload c5 --replicates 20 --streams 1 --no-cpu-baseline --no-secondary > gpurun_out/c5prof.log 2>&1
f=$(find /tmp/prof5 -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "at::" not in r["Name"] and "rocclr" not in r["Name"]]
out = open("gpurun_out/c5_kernel_stats.csv", "w")
out.write("kernel,calls,total_ms,avg_us\n")
for r in rows:
    name = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    out.write("%s,%s,%.3f,%.1f\n" % (name, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
PY
