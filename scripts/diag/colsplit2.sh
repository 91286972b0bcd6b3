# column pass lane split: experimental builds with two lanes per column (make exp-k EXPK=<K> ...) against the shipped kernels
cd $GRAFT_REPO_ROOT
for K in 36 44 52 56 64; do
  for rep in 1 2; do
    unset MCHIP_LIB_PATH; echo "K=$K shipped: $(python3 scripts/devbench.py --I 5000 --L 50000 --K $K --maxal 4 --steps 5 2>&1 | grep 'ms/step')"
    export MCHIP_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/libmulticlust_hip_k${K}s2.so; echo "K=$K s2:      $(python3 scripts/devbench.py --I 5000 --L 50000 --K $K --maxal 4 --steps 5 2>&1 | grep 'ms/step')"
  done
done
