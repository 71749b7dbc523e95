#!/bin/bash
# GPU box: the artefacts kept under profiles/ -- bench line (with cpu_baseline), per-kernel stats (overlapped and
# serial streams), per-shape conv table
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_$TAG
rm -rf $OUT; mkdir -p $OUT
python3 $R/bench.py --dump-conv $OUT/conv_table.json > $OUT/bench.json 2> $OUT/bench.err
for mode in 1 0; do
  DML_OVERLAP_WGRAD=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$mode -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile > $OUT/p$mode.log 2>&1
  find $OUT/p$mode -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_overlap$mode.csv \;
  rm -rf $OUT/p$mode
done
