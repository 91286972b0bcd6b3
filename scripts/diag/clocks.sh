# what the chip reports while the config-3 cycle runs: shader clock and socket power, sampled once a second beside a 1 500-cycle run
cd $GRAFT_REPO_ROOT
python3 bench.py --workload c3 --no-cpu-baseline --no-secondary --stability 0 --steps 1500 --warmup 5 --settle 0 > /tmp/clk_bench.json 2>/dev/null &
BP=$!
while kill -0 $BP 2>/dev/null; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -i "sclk\|Socket Graphics Package Power" | tr '\n' ' '; echo
  sleep 1
done
wait $BP
python3 -c "import json; d=json.loads(open('/tmp/clk_bench.json').read().strip().splitlines()[-1]); print('bench:', d['value'], 'EM it/s', d['ms_per_step'], 'ms per cycle', d['roofline']['kernels_ms'])"
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" | head -3
