# workgroups per CU of the two passes at config 2 (2 000 x 20 000, K = 5, plain EM): a small problem, where the grid is a few
# rounds of resident workgroups and the last partial round costs a whole one
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --workload c2 --no-cpu-baseline --stability 0 --steps 1500 --warmup 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_ms']; print('%-22s %.4f ms/iteration  %7.0f it/s  col %.4f ind %.4f rest %.4f' % ('$1', d['ms_per_step'], d['value'], k['column_pass'], k['individual_pass'], d['ms_per_step']-k['column_pass']-k['individual_pass']))"; }
for rep in 1 2; do
  unset MCHIP_BLOCKS_PER_CU_COL MCHIP_BLOCKS_PER_CU_IND MCHIP_SLAB_FRAC; one shipped
  for c in 4 5 6 7 9 12 14; do export MCHIP_BLOCKS_PER_CU_COL=$c; one col=$c; done
  unset MCHIP_BLOCKS_PER_CU_COL
  for c in 4 6 8 12 16 24 32; do export MCHIP_BLOCKS_PER_CU_IND=$c; one ind=$c; done
  unset MCHIP_BLOCKS_PER_CU_IND
done
