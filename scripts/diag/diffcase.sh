# one drawn case of tests/test_gpu_cli_differential.py with its files kept and shown; usage: diffcase.sh <case-number> [seed-shift] [cases]
cd $GRAFT_REPO_ROOT
N=${1:-107}; export MC_DIFF_SEED=${2:-5}; export MC_DIFF_CASES=${3:-400}
rm -rf /tmp/dcase; python -m pytest tests/test_gpu_cli_differential.py -m gpu -q -k "cases[$N-" --basetemp=/tmp/dcase 2>&1 | tail -3
D=$(ls -d /tmp/dcase/*/ | head -1)
for w in ref hip; do echo "== $w"; cat $D/$w.stdout; tail -5 $D/$w.stderr; ls $D/$w; head -12 $D/$w/*.out.txt; cat $D/$w/*etak.txt 2>/dev/null | head; head -6 $D/$w/*pklm.txt; done
# the per-iteration lines of both programs (-v 4), side by side where they first differ
ARGS=$(head -1 $D/hip.stderr)
mkdir -p /tmp/dv1 /tmp/dv2
oracle/_ref/multiclust_ref -f $D/d$N.stru -d /tmp/dv1/ $ARGS -v 4 > /tmp/dv1/out 2> /tmp/dv1/err
multiclust_amd/bin/multiclust -f $D/d$N.stru -d /tmp/dv2/ $ARGS -v 4 > /tmp/dv2/out 2> /tmp/dv2/err
echo "== -v 4: reference | this build (first 60 iteration lines)"
paste -d'|' <(grep -a "^ *[0-9]* (" /tmp/dv1/err | head -60) <(grep -a "^ *[0-9]* (" /tmp/dv2/err | head -60)
tail -2 /tmp/dv1/err; tail -2 /tmp/dv2/err
