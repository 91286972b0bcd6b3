# tetraploid S-side pass timing at config 5's shape (several runs: box clocks drift)
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
python3 bench.py --workload c5fit --no-cpu-baseline --no-secondary --steps 200 --warmup 10 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c5fit %.4f ms/step' % d['ms_per_step'], {k: round(v, 4) for k, v in d['roofline']['kernels_ms'].items()})"
done
