# SQ / LDS counters per kernel of the c3 bench (three separate --pmc passes, as r01_v4_c3_sq_counters.txt); usage: sq_c3.sh <out>
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=${1:-gpurun_out/sq_c3.txt}
rm -rf /tmp/sq1 /tmp/sq2 /tmp/sq3
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d /tmp/sq1 -o a -- python3 bench.py --steps 3 --warmup 1 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/sq1.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT --kernel-trace --output-format csv -d /tmp/sq2 -o b -- python3 bench.py --steps 3 --warmup 1 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/sq2.log 2>&1 &&
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM --kernel-trace --output-format csv -d /tmp/sq3 -o c -- python3 bench.py --steps 3 --warmup 1 --settle 0 --no-secondary --no-cpu-baseline --stability 0 > gpurun_out/sq3.log 2>&1 &&
python3 scripts/pmc_summary.py /tmp/sq1 /tmp/sq2 /tmp/sq3 > $OUT
