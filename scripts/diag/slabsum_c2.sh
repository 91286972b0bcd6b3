# config 2: N-side slabs summed by k_sum_slabs + k_finalize_p (shipped above 8 slabs) against k_finalize_p reading the slabs itself
cd $GRAFT_REPO_ROOT
one() { python3 bench.py --workload ${W:-c2} --no-cpu-baseline --no-secondary --stability 0 --steps ${S:-1500} --warmup 50 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-28s %.4f ms/step  %8.1f it/s' % ('$1', d['ms_per_step'], d['value']))"; }
for rep in 1 2 3; do
  unset MCHIP_NO_SLAB_SUM; one "k_sum_slabs + finalize"
  export MCHIP_NO_SLAB_SUM=1; one "finalize reads the slabs"
done
unset MCHIP_NO_SLAB_SUM
W=c1 S=300; for rep in 1 2; do unset MCHIP_NO_SLAB_SUM; one "c1 k_sum_slabs + finalize"; export MCHIP_NO_SLAB_SUM=1; one "c1 finalize reads slabs"; done
