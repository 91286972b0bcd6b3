# EXPERIMENT, not the product: the S-side pass with the N-side sums scattered into LDS accumulators by ds_add_f64
# (kernel code under MCHIP_EXP_SCATTER, K = 8 object only; `make exp-scatter` builds scripts/exp/libmulticlust_hip_scatter.so;
# DESIGN.md 4.3 (f)).
# Prints the shipped passes and the experimental S-side pass at config 3's shape in alternation.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  echo "shipped:"; python3 scripts/devbench.py --I 10000 --L 100000 --K 8 --maxal 4 --steps 5 2>&1 | grep "ms/step"
  echo "S-side pass + LDS scatter of q*r (no flush):"; MCHIP_LIB_PATH=$GRAFT_REPO_ROOT/scripts/exp/libmulticlust_hip_scatter.so python3 scripts/devbench.py --I 10000 --L 100000 --K 8 --maxal 4 --steps 5 2>&1 | grep "ms/step"
done
