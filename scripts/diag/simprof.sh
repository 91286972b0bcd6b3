# per-kernel times of one device-generated bootstrap data set (default: config 5's shape); usage: simprof.sh [I L ploidy K]
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf /tmp/profs
I=${1:-5000}; L=${2:-50000}; P=${3:-4}; K=${4:-7}
INITBENCH_HOST=0 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/profs -o s -- python3 scripts/initbench.py --I $I --L $L --ploidy $P --K $K > gpurun_out/simprof.log 2>&1
f=$(find /tmp/profs -name '*kernel_stats.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:8]:
    print(f"{r['Name'][:60]:60s} {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:9.1f} us  min {float(r['MinNs'])/1e3:9.1f}")
PY
