# A/B timing on one box: the shipped library against experimental builds (scripts/exp/libmulticlust_hip_<name>.so, `make exp-k8`),
# in alternation, config 3 (SQUAREM-3 cycles; per-kernel averages from the library's HIP events).  usage: ab.sh [-r rounds] name...
cd $GRAFT_REPO_ROOT
rounds=3
if [ "$1" = "-r" ]; then rounds=$2; shift 2; fi
one() { python3 bench.py --workload ${AB_WORKLOAD:-c3} --stability 0 --no-cpu-baseline --no-secondary --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('%-10s %.4f ms/cycle %.1f it/s  col %.3f ind %.3f ll %.3f dual %.3f' % ('$1', d['ms_per_step'], d['value'], k['column_pass'], k['individual_pass'], k['loglik_pass'], k['individual_dual_pass']))"; }
for rep in $(seq $rounds); do
  unset MCHIP_LIB_PATH; one shipped
  for v in "$@"; do export MCHIP_ALLOW_PARTIAL_ABI=1 MCHIP_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/libmulticlust_hip_$v.so; one $v; done
done
