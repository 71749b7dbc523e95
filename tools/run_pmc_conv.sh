#!/bin/bash
# GPU box: SQ counter passes over the conv micro-benchmark (forward only)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_conv
mkdir -p $OUT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/p1 -- python3 $R/tools/bench_conv.py fwd > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $OUT/p2 -- python3 $R/tools/bench_conv.py fwd > $OUT/p2.log 2>&1
python3 $R/tools/pmc_summary.py $OUT/p1 conv_igemm > $OUT/p1_summary.txt 2>&1
python3 $R/tools/pmc_summary.py $OUT/p2 conv_igemm > $OUT/p2_summary.txt 2>&1
rm -rf $OUT/p1 $OUT/p2
