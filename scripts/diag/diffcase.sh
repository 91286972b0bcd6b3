# one drawn case of tests/test_gpu_cli_differential.py with its files kept and shown; usage: diffcase.sh <case-number> [seed-shift] [cases]
cd $GRAFT_REPO_ROOT
N=${1:-107}; export MC_DIFF_SEED=${2:-5}; export MC_DIFF_CASES=${3:-400}
rm -rf /tmp/dcase; python -m pytest tests/test_gpu_cli_differential.py -m gpu -q -k "cases[$N-" --basetemp=/tmp/dcase 2>&1 | tail -3
D=$(ls -d /tmp/dcase/*/ | head -1)
for w in ref hip; do echo "== $w"; cat $D/$w.stdout; tail -5 $D/$w.stderr; ls $D/$w; head -12 $D/$w/*.out.txt; cat $D/$w/*etak.txt 2>/dev/null | head; head -6 $D/$w/*pklm.txt; done
