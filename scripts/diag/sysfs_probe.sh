# which clock / power files of the GPU an ordinary user can read, idle and under the config-3 cycle (for an in-run sample in bench.py)
cd $GRAFT_REPO_ROOT
for d in /sys/class/drm/card*/device; do
  echo "== $d"; cat $d/vendor 2>/dev/null
  ls $d | grep -i "pp_dpm_sclk\|gpu_busy\|hwmon" | tr '\n' ' '; echo
  for h in $d/hwmon/hwmon*; do echo "-- $h"; ls $h | tr '\n' ' '; echo; for f in freq1_input power1_average power1_input; do [ -r $h/$f ] && echo "$f = $(cat $h/$f)"; done; done
  [ -r $d/pp_dpm_sclk ] && cat $d/pp_dpm_sclk
done
python3 bench.py --workload c3 --no-cpu-baseline --no-secondary --stability 0 --steps 800 --warmup 5 --settle 0 > /tmp/p_bench.json 2>/dev/null &
BP=$!
sleep 12
for i in 1 2 3 4 5; do for d in /sys/class/drm/card*/device; do for h in $d/hwmon/hwmon*; do echo "$(cat $h/freq1_input 2>/dev/null) Hz $(cat $h/power1_average 2>/dev/null) $(cat $h/power1_input 2>/dev/null) uW busy $(cat $d/gpu_busy_percent 2>/dev/null)"; done; done; sleep 0.5; done
wait $BP
