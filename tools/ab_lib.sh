#!/bin/bash
# GPU box: A/B of two builds of the library (dmlnet/libdmlnet_hip_A.so / _B.so) on the whole step, interleaved pairs on one box;
# the build is selected through DML_LIB_PATH (dmlnet/_lib.py), the in-tree libdmlnet_hip.so is not touched.
#   gpurun -- bash tools/ab_lib.sh 3 [bench.py args]
R=$GRAFT_REPO_ROOT
L=$R/open-world-semantic-segmentation_amd/dmlnet
N=${1:-3}; shift
for i in $(seq 1 $N); do
  for v in A B; do
    DML_LIB_PATH=$L/libdmlnet_hip_$v.so python3 $R/bench.py --no-cpu-baseline --no-companions --no-profile --steps 30 --warmup 5 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v run $i: %.1f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
