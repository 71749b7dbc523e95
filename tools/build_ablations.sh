#!/bin/bash
# Authoring container: timing-ablation builds of the two-plane conv K loop (DML_WS_ABL bit mask, conv_igemm.hip) as
# dmlnet/libdmlnet_hip_abl<N>.so -- selected through DML_LIB_PATH by tools/bench_h2.py runs on the GPU box.  Results are garbage.
#   bash tools/build_ablations.sh 1 2 3 4 5 7
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd); P=$R/open-world-semantic-segmentation_amd
make -s -j8 -C $P/csrc >/dev/null
for n in "$@"; do
  ( d=$(mktemp -d /tmp/abl.XXXX)
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function -Wno-pass-failed -DDML_WS_ABL=$n -c $P/csrc/conv_igemm.hip -o $d/conv_igemm.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $d/conv_igemm.o $P/csrc/build/bn.o $P/csrc/build/pool_resize.o $P/csrc/build/head.o $P/csrc/build/optim_layout.o $P/csrc/build/augment.o $P/csrc/build/ood_measures.o $P/csrc/build/plan_exec.o -o $P/dmlnet/libdmlnet_hip_abl$n.so
    rm -rf $d ) &
done
wait
ls $P/dmlnet/libdmlnet_hip_abl*.so
