#!/bin/bash
# Authoring container: two builds of the library for tools/ab_lib.sh -- A from the csrc/ + include/ of a git ref (default HEAD),
# B from the working tree -- as dmlnet/libdmlnet_hip_A.so / _B.so (git-ignored, travel with gpurun).
#   bash tools/build_ab.sh [git-ref]
set -euo pipefail
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/open-world-semantic-segmentation_amd
REF=${1:-HEAD}
T=$(mktemp -d /tmp/dml_ab.XXXX)
mkdir -p $T/pkg/csrc $T/include $T/pkg/dmlnet
git -C $R archive $REF open-world-semantic-segmentation_amd/csrc include | tar -x -C $T
mv $T/open-world-semantic-segmentation_amd/csrc/* $T/pkg/csrc/
rm -rf $T/pkg/csrc/build
make -s -j8 -C $T/pkg/csrc >/dev/null
cp $T/pkg/dmlnet/libdmlnet_hip.so $P/dmlnet/libdmlnet_hip_A.so
make -s -j8 -C $P/csrc >/dev/null
cp $P/dmlnet/libdmlnet_hip.so $P/dmlnet/libdmlnet_hip_B.so
rm -rf $T
ls -la $P/dmlnet/libdmlnet_hip_[AB].so
