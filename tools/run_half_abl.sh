R=${GRAFT_REPO_ROOT}
L=$R/open-world-semantic-segmentation_amd/dmlnet
export BENCH_SHAPES="16,48,48,256,1024,1,1;16,48,48,1024,256,1,1"
for v in full abl8 abl24 abl12; do
  lib=$L/libdmlnet_hip_$v.so; [ $v = full ] && lib=$L/libdmlnet_hip.so
  for h in 0 1; do
    echo "== $v DML_WS_HALF=$h: fwd | dgrad EPI 3"
    DML_WS_HALF=$h DML_LIB_PATH=$lib python3 $R/tools/bench_h2.py fwd only=h2 2>/dev/null
    DML_WS_HALF=$h BENCH_EPI=3 DML_LIB_PATH=$lib python3 $R/tools/bench_h2.py dgrad only=h2 2>/dev/null
  done
done
