# rocprofv3 summaries of a bench line: kernel trace + the two PMC passes; usage: c3prof.sh <tag> [workload [steps]]
# (tag names the files under profiles/: r04_v1_c3 -> profiles/r04_v1_c3_kernel_stats.csv, _traffic.json, ...)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
TAG=${1:-rXX_c3}
WL=${2:-c3}
STEPS=${3:-20}
PSTEPS=$(( STEPS / 5 > 0 ? STEPS / 5 : 1 ))
rm -rf /tmp/prof /tmp/pmc_f /tmp/pmc_w
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof -o v -- python3 bench.py --workload $WL --steps $STEPS --warmup 3 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/${TAG}_trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pmc_f -o f -- python3 bench.py --workload $WL --steps $PSTEPS --warmup 1 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/${TAG}_pmcf.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pmc_w -o w -- python3 bench.py --workload $WL --steps $PSTEPS --warmup 1 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/${TAG}_pmcw.log 2>&1 &&
python3 scripts/summarize_profile.py $TAG /tmp/prof /tmp/pmc_f /tmp/pmc_w && mkdir -p gpurun_out/profiles && cp profiles/${TAG}_* gpurun_out/profiles/
