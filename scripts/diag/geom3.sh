# workgroups per CU of the two passes at config 3, re-swept with round 3's kernels (alternating in one call)
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --workload c3 --no-cpu-baseline --no-secondary --stability 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('%-14s %.4f ms/cycle  col %.3f ind %.3f dual %.3f  rest %.3f' % ('$1', d['ms_per_step'], k['column_pass'], k['individual_pass'], k['individual_dual_pass'], d['ms_per_step']-2*k['column_pass']-k['individual_pass']-k['individual_dual_pass']))"; }
for rep in 1 2; do
  unset MCHIP_BLOCKS_PER_CU_COL MCHIP_BLOCKS_PER_CU_IND; one shipped
  for c in 16 32 48; do export MCHIP_BLOCKS_PER_CU_COL=$c; one col=$c; done
  unset MCHIP_BLOCKS_PER_CU_COL
  for c in 24 32 48 96; do export MCHIP_BLOCKS_PER_CU_IND=$c; one ind=$c; done
  unset MCHIP_BLOCKS_PER_CU_IND
done
