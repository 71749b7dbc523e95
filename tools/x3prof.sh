#!/bin/bash
# GPU box: SQ / LDS counters of the split-product kernels over tools/bench_x3.py (one --pmc pass, kernel trace only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/x3prof
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES --kernel-trace --output-format csv -d $OUT/sq -- python3 $R/tools/bench_x3.py > $OUT/sq.log 2>&1
find $OUT/sq -name "*counter_collection.csv" -exec cp {} $OUT/x3_counters.csv \;
rm -rf $OUT/sq
ls -la $OUT
