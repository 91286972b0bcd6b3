cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof5 -o c5 -- python3 bench.py --workload c5 --replicates 20 --streams 1 --no-cpu-baseline --no-secondary > gpurun_out/c5prof.log 2>&1
f=$(find /tmp/prof5 -name '*kernel_stats.csv' | head -1)
cp $f gpurun_out/c5_kernel_stats.csv
