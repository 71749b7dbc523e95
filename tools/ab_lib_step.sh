#!/bin/bash
# GPU box: interleaved whole-step A/B of dmlnet/libdmlnet_hip_A.so / _B.so (tools/build_ab.sh)
R=$GRAFT_REPO_ROOT
bash $R/tools/ab_multi.sh ${1:-3} "A|A|" "B|B|"
