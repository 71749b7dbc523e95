#!/bin/bash
# GPU box: A/B of two builds of the library (dmlnet/libdmlnet_hip_A.so / _B.so, copied over libdmlnet_hip.so in turn) on the
# whole bf16 step, interleaved pairs on one box:   gpurun -- bash tools/ab_lib.sh 3
R=$GRAFT_REPO_ROOT
L=$R/open-world-semantic-segmentation_amd/dmlnet
N=${1:-3}
for i in $(seq 1 $N); do
  for v in A B; do
    cp $L/libdmlnet_hip_$v.so $L/libdmlnet_hip.so
    python3 $R/bench.py --no-cpu-baseline --no-fp32-companion --no-profile --steps 30 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$v run $i: %.1f images/s  %.3f ms' % (d['value'], d['ms_per_step']))"
  done
done
cp $L/libdmlnet_hip_B.so $L/libdmlnet_hip.so
