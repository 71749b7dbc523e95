#!/bin/bash
# GPU box: SQ / LDS / TA counters of the two-plane conv kernel (conv_ws_kernel<.., 2>) on one shape (tools/bench_h2.py fwd only=h2),
# small --pmc groups, kernel trace only.   gpurun -- bash tools/pmc_h2.sh "16,48,48,4096,256,1,1" [fwd|dgrad|wgrad]
set -uo pipefail
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04/pmc_h2
rm -rf $OUT; mkdir -p $OUT
export BENCH_SHAPES="${1:-16,48,48,4096,256,1,1}"
WHICH=${2:-fwd}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" \
           "GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_SALU SQ_INSTS_VALU" \
           "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_INSTS_MFMA" \
           "TA_TA_BUSY_sum TA_BUFFER_WAVEFRONTS_sum TCP_PENDING_STALL_CYCLES_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/p$i -- python3 $R/tools/bench_h2.py $WHICH only=h2 > $OUT/p$i.log 2>&1
done
python3 $R/tools/pmc_summary.py $OUT conv_ > $OUT/summary.txt 2>&1
cat $OUT/summary.txt
find $OUT -name "*.csv" -size +2M -delete
