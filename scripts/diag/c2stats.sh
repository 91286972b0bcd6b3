# rocprofv3 kernel statistics of the config-2 bench (plain EM batches): which small kernels an iteration is made of
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_c2
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_c2 -o v -- python3 bench.py --workload c2 --steps 640 --warmup 64 --settle 0 --no-cpu-baseline --stability 0 > gpurun_out/c2_trace.log 2>&1
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/prof_c2/**/*kernel_stats.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r['TotalDurationNs']))
for r in rows[:16]:
    print('%-70s calls %6s  avg %8.2f us  total %8.2f ms' % (r['Name'][:70], r['Calls'], float(r['AverageNs'])/1e3, float(r['TotalDurationNs'])/1e6))
PY
