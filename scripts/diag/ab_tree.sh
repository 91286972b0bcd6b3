# A/B timing on one box, whole trees: this tree against a copy of an earlier one (scripts/exp/<name>tree/: its bench.py, Python
# views and built libraries; git-ignored, it travels with gpurun), in alternation.  usage: ab_tree.sh [-r rounds] [-w workload] name...
cd $GRAFT_REPO_ROOT
rounds=3; wl=c3; steps=20
while [ "${1#-}" != "$1" ]; do
  case $1 in -r) rounds=$2; shift 2;; -w) wl=$2; shift 2;; -s) steps=$2; shift 2;; *) break;; esac
done
one() { (cd $2 && python3 bench.py --workload $wl --stability 0 --no-cpu-baseline --no-secondary --steps $steps --warmup 5 2>/dev/null) | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('%-10s %.4f ms/step %.1f it/s  col %.3f ind %.3f ll %.3f dual %.3f  passes/step %.3f' % ('$1', d['ms_per_step'], d['value'], k['column_pass'], k['individual_pass'], k['loglik_pass'], k['individual_dual_pass'], sum(d['roofline']['kernels_ms_per_step'].values())))"; }
for rep in $(seq $rounds); do
  one this $GRAFT_REPO_ROOT
  for v in "$@"; do one $v $GRAFT_REPO_ROOT/scripts/exp/${v}tree; done
done
