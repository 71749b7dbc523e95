#!/bin/bash
# GPU box: memory-copy + kernel trace of a short bench run (no counters).  Used to place the 1352 __amd_rocclr_copyBuffer
# dispatches of a bench trace: they are the 2 x 676 slice copies of the parameter-store flattening at model setup (runs of
# consecutive copies next to torch's elementwise kernels), none of them inside a train step.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/memcopy
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-fp32-companion > $OUT/log.txt 2>&1
find $OUT/p -name "*memory_copy_trace.csv" -exec cp {} $OUT/memory_copy_trace.csv \;
find $OUT/p -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/p
ls -la $OUT
