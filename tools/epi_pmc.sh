#!/bin/bash
# GPU box: vector-memory pipeline counters (TA / TCP / TD) of the data-gradient kernel with its fused epilogue options
# (tools/bench_conv.py epi, one shape), small --pmc groups (a group the hardware cannot collect aborts and hangs: timeout),
# kernel trace only.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/epi_pmc
rm -rf $OUT; mkdir -p $OUT
export BENCH_SHAPES="16,48,48,1024,256,1,1"
i=0
for grp in "GRBM_GUI_ACTIVE TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TA_DATA_STALLED_BY_TC_CYCLES_sum TA_BUFFER_WAVEFRONTS_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" \
           "TD_TD_BUSY_sum TD_TC_STALL_sum" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_conv.py epi > $OUT/p$i.log 2>&1
  find $OUT/p$i -name "*counter_collection.csv" -exec cp {} $OUT/pass$i.csv \;
  rm -rf $OUT/p$i
done
ls -la $OUT
