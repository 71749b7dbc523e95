#!/bin/bash
# GPU box: interleaved whole-step comparison of several builds of the library (DML_LIB_PATH): libdmlnet_hip_<tag>.so for every tag given,
# "main" = the in-tree libdmlnet_hip.so.     gpurun -- bash tools/ab_libs3.sh 2 A B main
R=$GRAFT_REPO_ROOT
L=$R/open-world-semantic-segmentation_amd/dmlnet
N=${1:-2}; shift
for i in $(seq 1 $N); do
  for v in "$@"; do
    P=$L/libdmlnet_hip_$v.so; [ "$v" = main ] && P=$L/libdmlnet_hip.so
    DML_LIB_PATH=$P python3 $R/bench.py --no-cpu-baseline --no-companions --no-profile --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v run $i: %.1f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
