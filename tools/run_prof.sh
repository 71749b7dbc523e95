#!/bin/bash
# GPU box: per-kernel statistics of the bench step (serial streams so that durations add up)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$1
mkdir -p $OUT
DML_OVERLAP_WGRAD=${2:-0} rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/bench.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
find $OUT -name "*_kernel_trace.csv" -exec cp {} $OUT/kernel_trace.csv \;
find $OUT -mindepth 1 -maxdepth 1 -type d -exec rm -rf {} +
python3 - <<PY
import csv, collections
rows = list(csv.DictReader(open("$OUT/kernel_trace.csv")))
print(len(rows), "dispatches")
PY
gzip -f $OUT/kernel_trace.csv
