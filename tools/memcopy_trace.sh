cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/memcopy
rm -rf $OUT; mkdir -p $OUT
timeout 200 rocprofv3 --memory-copy-trace --kernel-trace --output-format csv -d $OUT/p -- python3 $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-fp32-companion > $OUT/log.txt 2>&1
find $OUT/p -name "*memory_copy_trace.csv" -exec cp {} $OUT/memory_copy_trace.csv \;
find $OUT/p -name "*kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
rm -rf $OUT/p
ls -la $OUT
