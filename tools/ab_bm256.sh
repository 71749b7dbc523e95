#!/bin/bash
# GPU box: interleaved whole-step runs over the K threshold of the 256-row-tile rule (0 = off, N >= 64 = threshold)
R=$GRAFT_REPO_ROOT
for i in 1 2; do
  for v in 0 1152 2304 4608; do
    env DML_CONV_BM256=$v python3 $R/bench.py --no-cpu-baseline --no-fp32-companion --no-profile 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('BM256 K>=$v  %.1f img/s  %.2f ms' % (d['value'], d['ms_per_step']))"
  done
done
