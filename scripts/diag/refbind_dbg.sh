# the bound reference program beside our own command line on one small fit (debugging aid); usage: refbind_dbg.sh
cd $GRAFT_REPO_ROOT
mkdir -p /tmp/rb1 /tmp/rb2 /tmp/rb3
S=tests/golden/data/multi.stru
for i in 1 2; do oracle/_ref/multiclust_ref_hip -f $S -d /tmp/rb$i -a -k 4 -r 7 -n 2 -T 3 2>&1 | tail -4; echo "rc=$?"; done
multiclust_amd/bin/multiclust -f $S -d /tmp/rb3 -a -k 4 -r 7 -n 2 -T 3 2>&1 | tail -4
cat tests/golden/cli_multi_admix_k4_i1000_T5/ARGS.txt
