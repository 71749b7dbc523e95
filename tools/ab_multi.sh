#!/bin/bash
# GPU box: interleaved whole-step comparison of several (library build, environment) variants on ONE box.
#   bash tools/ab_multi.sh ROUNDS "name|lib-suffix|ENV=1 ENV2=x" ...       lib-suffix: A / B (tools/build_ab.sh) or - for the in-tree build
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/open-world-semantic-segmentation_amd/dmlnet
N=$1; shift
for i in $(seq 1 $N); do
  for spec in "$@"; do
    IFS='|' read -r name lib envs <<< "$spec"
    libenv=""
    [ "$lib" != "-" ] && libenv="DML_LIB_PATH=$L/libdmlnet_hip_$lib.so"
    env $libenv $envs python3 $R/bench.py --no-cpu-baseline --no-companions --no-profile --steps ${AB_STEPS:-30} --warmup 5 ${AB_ARGS:-} 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-24s run $i: %.1f images/s  %.3f ms' % ('$name', d['value'], d['ms_per_step']))"
  done
done
