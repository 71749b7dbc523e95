#!/bin/bash
# GPU box: where do the short-K layer3 1x1 launches spend their time?  Timing ablations of the two-plane kernel (DML_WS_ABL builds,
# tools/build_ablations.sh 4 8 12 16 24): 4 = no DMA, 8 = no epilogue, 16 = no MFMAs, 24 = DMA + flags only, 12 = K loop compute only.
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
L=$R/open-world-semantic-segmentation_amd/dmlnet
export BENCH_SHAPES="16,48,48,256,1024,1,1;16,48,48,1024,256,1,1;16,48,48,256,256,3,1"
for v in full abl4 abl8 abl16 abl24 abl12; do
  lib=$L/libdmlnet_hip_$v.so; [ $v = full ] && lib=$L/libdmlnet_hip.so
  echo "== $v: forward with BN statistics"
  DML_LIB_PATH=$lib python3 $R/tools/bench_h2.py fwd only=h2 2>/dev/null
  echo "== $v: data gradient, no epilogue operand"
  DML_LIB_PATH=$lib python3 $R/tools/bench_h2.py dgrad only=h2 2>/dev/null
  echo "== $v: data gradient + fused BN-backward sums (EPI 2)"
  BENCH_EPI=2 DML_LIB_PATH=$lib python3 $R/tools/bench_h2.py dgrad only=h2 2>/dev/null
  echo "== $v: data gradient + identity-branch gradient + sums (EPI 3)"
  BENCH_EPI=3 DML_LIB_PATH=$lib python3 $R/tools/bench_h2.py dgrad only=h2 2>/dev/null
done
