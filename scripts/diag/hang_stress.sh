# does a launch of the command line ever fail to return?  N launches of one tiny mixture fit (0.3 s each); a launch alive after 20 s
# has its threads' wait channels recorded before it is killed.  usage: hang_stress.sh [N]
cd $GRAFT_REPO_ROOT
N=${1:-1200}
python3 scripts/diag/hang184.py > /dev/null 2>&1      # builds /tmp/h184/d.stru
mkdir -p /tmp/hs; hung=0; t0=$(date +%s)
for i in $(seq 1 $N); do
  multiclust_amd/bin/multiclust -f /tmp/h184/d.stru -d /tmp/hs/ -p 1 -k 2 -r 3593 -n 2 -g 3 -T 40 -s 3 > /tmp/hs/out 2> /tmp/hs/err &
  pid=$!
  for w in $(seq 1 200); do kill -0 $pid 2>/dev/null || break; sleep 0.1; done
  if kill -0 $pid 2>/dev/null; then
    hung=$((hung+1)); echo "== launch $i (pid $pid) alive after 20 s"
    for t in /proc/$pid/task/*; do echo "$(cat $t/comm 2>/dev/null) state=$(grep State $t/status 2>/dev/null | tr -s '\t ' ' ') wchan=$(cat $t/wchan 2>/dev/null)"; done
    cat /proc/$pid/stack 2>/dev/null | head -20
    echo "stdout so far:"; cat /tmp/hs/out | head -5; echo "files:"; ls /tmp/hs | head
    kill -9 $pid
  fi
  wait $pid 2>/dev/null
  if [ $((i % 200)) -eq 0 ]; then echo "$i launches, $hung hung, $(( $(date +%s) - t0 )) s"; fi
done
echo "done: $N launches, $hung hung"
