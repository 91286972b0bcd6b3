# column-pass chunk count: A/B runs of the single-fit workloads in one call (box clocks differ between calls);
# MCHIP_BLOCKS_PER_CU_COL=64 = 64 workgroups per CU whatever the slab bytes, unset = the shipped rule
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --workload $1 --no-cpu-baseline --no-secondary --steps $2 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-6s %-8s %.4f ms/step' % ('$1', '$3', d['ms_per_step']), {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"; }
for rep in 1 2 3; do
  export MCHIP_BLOCKS_PER_CU_COL=64
  one c5fit 200 col=64; one c3 20 col=64; one c2 400 col=64
  unset MCHIP_BLOCKS_PER_CU_COL
  one c5fit 200 shipped; one c3 20 shipped; one c2 400 shipped
done
