# N-side slabs added by k_sum_slabs before k_finalize_p against k_finalize_p reading them all, alternating in one call
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --workload $1 --no-cpu-baseline --no-secondary --steps $2 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; big=2*k['column_pass']+k['individual_pass']+k['individual_dual_pass'] if '$1'=='c3' else k['column_pass']+k['individual_pass']; print('%-6s %-9s %.4f ms/step, outside the passes %.4f' % ('$1', '$3', d['ms_per_step'], d['ms_per_step']-big))"; }
for rep in 1 2 3; do
  export MCHIP_NO_SLAB_SUM=1; one c3 20 direct; one c5fit 200 direct
  unset MCHIP_NO_SLAB_SUM; one c3 20 slab-sum; one c5fit 200 slab-sum
done
