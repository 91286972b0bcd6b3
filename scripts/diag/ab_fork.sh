# the side stream (column pass beside the individual pass, run_estep) on and off, whole processes in alternation on one box:
# c2 and c3 batched without profiling hooks (scripts/diag/graph_timing.py), c5 and c4 through bench.py
cd $GRAFT_REPO_ROOT
val() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-8s %-8s %.1f it/s  %.3f ms/step' % ('$1', '$2', d['value'], d['ms_per_step']))"; }
for r in 1 2; do
  TAG=fork python3 scripts/diag/graph_timing.py c2 300 5 2>&1 | tail -1
  TAG=nofork MCHIP_NO_FORK=1 python3 scripts/diag/graph_timing.py c2 300 5 2>&1 | tail -1
  TAG=fork python3 scripts/diag/graph_timing.py c3 20 4 2>&1 | tail -1
  TAG=nofork MCHIP_NO_FORK=1 python3 scripts/diag/graph_timing.py c3 20 4 2>&1 | tail -1
  python3 bench.py --workload c5 --replicates 40 --no-cpu-baseline 2>/dev/null | val c5 fork
  MCHIP_NO_FORK=1 python3 bench.py --workload c5 --replicates 40 --no-cpu-baseline 2>/dev/null | val c5 nofork
done
python3 bench.py --workload c4 --units 8 --no-cpu-baseline 2>/dev/null | val c4 fork
MCHIP_NO_FORK=1 python3 bench.py --workload c4 --units 8 --no-cpu-baseline 2>/dev/null | val c4 nofork
