#!/bin/bash
# GPU box: interleaved A/B of bench.py under an environment switch.  usage: ab_bench.sh VAR A_VALUE B_VALUE [rounds]
VAR=$1; A=$2; B=$3; N=${4:-3}
R=$GRAFT_REPO_ROOT
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python3 $R/bench.py --no-cpu-baseline --no-fp32-companion --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$VAR=$v  %.1f img/s  %.2f ms' % (d['value'], d['ms_per_step']))"
  done
done
