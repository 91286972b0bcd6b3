# column pass for K > 48: lanes-per-column split against the one-lane kernel, alternating (5 000 x 50 000 diploid, M_l ~ U{2,3,4})
cd $GRAFT_REPO_ROOT
for K in 49 52 56 60 64; do
  for rep in 1 2; do
    unset MCHIP_NO_COL_SPLIT; echo "K=$K split:    $(python3 scripts/devbench.py --I 5000 --L 50000 --K $K --maxal 4 --steps 5 2>&1 | grep 'ms/step')"
    export MCHIP_NO_COL_SPLIT=1; echo "K=$K one lane: $(python3 scripts/devbench.py --I 5000 --L 50000 --K $K --maxal 4 --steps 5 2>&1 | grep 'ms/step')"
  done
done
