# `make exp-k52` first (build container).  On the GPU box: the one-lane K = 52 instance under both spill modes.
cd $GRAFT_REPO_ROOT
for v in agpr scratch; do
  echo "== $v"; MCHIP_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/libmulticlust_hip_k52$v.so python3 scripts/diag/k52_spill.py 2>&1 | tail -12
done
