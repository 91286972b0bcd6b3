# dual individual pass of the accelerated cycle against the two separate passes, alternating in one call
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --workload $1 --accel 3 --no-cpu-baseline --no-secondary --steps $2 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('%-6s %-8s %.4f ms/cycle  col %.3f ind %.3f ll %.3f dual %.3f  (ind+ll %.3f)' % ('$1', '$3', d['ms_per_step'], k['column_pass'], k['individual_pass'], k['loglik_pass'], k['individual_dual_pass'], k['individual_pass'] + k['loglik_pass']))"; }
for rep in 1 2 3; do
  export MCHIP_NO_DUAL=1
  one c3 20 separate; one c5fit 50 separate
  unset MCHIP_NO_DUAL
  one c3 20 dual; one c5fit 50 dual
done
