#!/bin/bash
# GPU box: the artefacts kept under profiles/ -- bench line (with fp32 companion and cpu_baseline), per-kernel stats
# (overlapped and serial streams), per-shape conv table, PMC traffic (whole step + the head kernels).
#   gpurun -- bash tools/run_final_profiles.sh r02_v1      ->  gpurun_out/final_r02_v1/
TAG=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_$TAG
rm -rf $OUT; mkdir -p $OUT
SHA=$(python3 $R/bench.py --print-csrc-sha)
python3 $R/bench.py --dump-conv $OUT/conv_table.json > $OUT/bench.json 2> $OUT/bench.err
for mode in 1 0; do
  DML_OVERLAP_WGRAD=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$mode -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-fp32-companion > $OUT/p$mode.log 2>&1
  find $OUT/p$mode -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_overlap$mode.csv \;
  rm -rf $OUT/p$mode
done
# HBM traffic, separate --pmc passes (MI355X_MICROARCH.md: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024)
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp32-companion > $OUT/rd.log 2>&1
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile --no-fp32-companion > $OUT/wr.log 2>&1
python3 $R/tools/traffic_summary.py $OUT/rd $OUT/wr 6 $SHA > $OUT/traffic_pmc.json 2> $OUT/traffic.err
rm -rf $OUT/rd $OUT/wr
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $R/tools/bench_dist.py > $OUT/drd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/tools/bench_dist.py > $OUT/dwr.log 2>&1
python3 $R/tools/traffic_dist_summary.py $OUT/rd $OUT/wr $SHA > $OUT/traffic_dist_pmc.json 2> $OUT/traffic_dist.err
rm -rf $OUT/rd $OUT/wr
ls -la $OUT
