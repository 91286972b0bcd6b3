# per-step kernel times against K at the 5 000 x 50 000 diploid shape (M_l ~ U{2,3,4}); usage: ksweep.sh > profiles/rNN_k_sweep.txt
cd $GRAFT_REPO_ROOT
echo "# python3 scripts/devbench.py --I 5000 --L 50000 --maxal 4 --steps 5 --K <K>  (column pass: two lanes per column from K = 37; S-side pass: 2 lanes per individual from K = 28, 4 above 48)"
for K in 4 8 12 16 20 24 28 32 36 38 39 40 44 48 52 56 60 64; do
  python3 scripts/devbench.py --I 5000 --L 50000 --K $K --maxal 4 --steps 5 2>&1 | grep 'ms/step'
done
