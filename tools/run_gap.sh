#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gap
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 6 --warmup 3 --no-cpu-baseline --no-profile --no-companions > $OUT/log.txt 2>&1
F=$(find $OUT/p -name "*kernel_trace.csv" | head -1)
python3 $R/tools/gap_analysis.py $F
cp $F $OUT/trace.csv; rm -rf $OUT/p
