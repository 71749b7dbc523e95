#!/bin/bash
# GPU box: vector-memory-path and SQ counters of the full-tile two-plane kernel against the half-tile / two-workgroups-per-CU kernel
# (DML_WS_HALF=1) on the short-K forward launch of layer3 (1x1 256 -> 1024 at 48 x 48, with BN statistics).  Small --pmc groups, one
# pass each, kernel trace only; per kernel: counter sums per launch.   -> gpurun_out/r06_half_pmc.txt
: "${GRAFT_REPO_ROOT:?run under gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/halfpmc
rm -rf $OUT; mkdir -p $OUT
export BENCH_SHAPES="16,48,48,256,1024,1,1"
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  for h in 0 1; do
    DML_WS_HALF=$h timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p${i}_$h -- python3 $R/tools/bench_h2.py fwd only=h2 > $OUT/p${i}_$h.log 2>&1
    find $OUT/p${i}_$h -name "*counter_collection.csv" -exec cp {} $OUT/pass${i}_$h.csv \;
    rm -rf $OUT/p${i}_$h
  done
done
python3 - $OUT $R/gpurun_out/r06_half_pmc.txt <<'PY'
import csv, glob, sys, collections
acc = {0: collections.defaultdict(float), 1: collections.defaultdict(float)}
n = {0: collections.Counter(), 1: collections.Counter()}
for f in sorted(glob.glob(sys.argv[1] + "/pass*_*.csv")):
    h = int(f[-5])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "conv_ws" not in k: continue
        acc[h][r["Counter_Name"]] += float(r["Counter_Value"])
        n[h][r["Counter_Name"]] += 1
out = open(sys.argv[2], "w")
out.write("%-40s %16s %16s %8s\n" % ("counter (mean per launch)", "full tile", "half tile x2", "ratio"))
for c in sorted(set(acc[0]) | set(acc[1])):
    a = acc[0][c] / max(1, n[0][c]); b = acc[1][c] / max(1, n[1][c])
    out.write("%-40s %16.4g %16.4g %8.2f\n" % (c, a, b, b / a if a else 0))
out.close()
print(open(sys.argv[2]).read())
PY
