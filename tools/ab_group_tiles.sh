R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for cfg in "DML_GROUP_WGRAD=0" "DML_GROUP_TILES=17" "DML_GROUP_TILES=34" "DML_GROUP_TILES=48"; do
    env $cfg python3 $R/bench.py --no-cpu-baseline --no-fp32-companion --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$cfg  %.1f img/s  %.2f ms' % (d['value'], d['ms_per_step']))"
  done
done
