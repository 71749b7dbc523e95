#!/bin/bash
# GPU box: the artefacts kept under profiles/ for round 3 -- bench line (bf16 headline + first-class fp32 companion + cpu
# baseline + per-class two-roof table), rocprofv3 kernel stats (bf16 overlapped / serial, fp32 serial, the standalone
# distance kernel), PMC HBM traffic per kernel and per conv shape class (bf16 and fp32), SQ counters per kernel template.
#   gpurun -- bash tools/run_r03_profiles.sh v1   ->  gpurun_out/final_r03_v1/
# Every profiled program is `python3 bench.py ...` directly after `--` (no env / bash -c hop), counters in their own runs.
set -euo pipefail
TAG=${1:?tag}
: "${GRAFT_REPO_ROOT:?run under gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r03_$TAG
rm -rf $OUT; mkdir -p $OUT
SHA=$(python3 $R/bench.py --print-csrc-sha)
B="$R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-fp32-companion"
python3 $R/bench.py --dump-conv $OUT/conv_table.json > $OUT/bench.json 2> $OUT/bench.err
# per-kernel time: bf16 with the weight gradients on the side stream (as timed) and serial (un-contended), fp32 serial
for mode in 1 0; do
  DML_OVERLAP_WGRAD=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$mode -- python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-fp32-companion > $OUT/p$mode.log 2>&1
  find $OUT/p$mode -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_overlap$mode.csv \;
  rm -rf $OUT/p$mode
done
DML_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf -- python3 $R/bench.py --dtype f32 --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $OUT/pf.log 2>&1
find $OUT/pf -name "*kernel_stats.csv" -exec cp {} $OUT/fp32_kernel_stats_serial.csv \;
rm -rf $OUT/pf
DML_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/px -- python3 $R/bench.py --dtype f32x3 --steps 4 --warmup 2 --no-cpu-baseline --no-profile > $OUT/px.log 2>&1
find $OUT/px -name "*kernel_stats.csv" -exec cp {} $OUT/fp32x3_kernel_stats_serial.csv \;
rm -rf $OUT/px
# kernel stats of the SAME short serial command the counter passes use (durations next to the SQ counters)
DML_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ps -- python3 $B > $OUT/ps.log 2>&1
find $OUT/ps -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats_pmc_cmd.csv \;
rm -rf $OUT/ps
# HBM traffic, separate --pmc passes (MI355X_MICROARCH.md: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024), bf16 then fp32
export DML_BENCH_OPLOG=$OUT/oplog_bf16.json
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $B > $OUT/rd.log 2>&1
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $B > $OUT/wr.log 2>&1
python3 $R/tools/pmc_by_class.py traffic $OUT/rd $OUT/wr $OUT/oplog_bf16.json $SHA > $OUT/traffic_pmc.json 2> $OUT/traffic.err
rm -rf $OUT/rd $OUT/wr
# SQ counters (8 SQ slots), one pass
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/sq -- python3 $B > $OUT/sq.log 2>&1
python3 $R/tools/pmc_by_class.py sq $OUT/sq $OUT/oplog_bf16.json $OUT/kernel_stats_pmc_cmd.csv $SHA > $OUT/conv_pmc_sq.json 2> $OUT/sq.err
rm -rf $OUT/sq
export DML_BENCH_OPLOG=$OUT/oplog_fp32.json
F="$R/bench.py --dtype f32 --steps 2 --warmup 1 --no-cpu-baseline --no-profile"
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $F > $OUT/frd.log 2>&1
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $F > $OUT/fwr.log 2>&1
python3 $R/tools/pmc_by_class.py traffic $OUT/rd $OUT/wr $OUT/oplog_fp32.json $SHA > $OUT/fp32_traffic_pmc.json 2> $OUT/ftraffic.err
rm -rf $OUT/rd $OUT/wr
unset DML_BENCH_OPLOG
# the standalone distance kernel: duration from the kernel trace + its PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pd -- python3 $R/tools/bench_dist.py > $OUT/pd.log 2>&1
find $OUT/pd -name "*kernel_stats.csv" -exec cp {} $OUT/dist_kernel_stats.csv \;
rm -rf $OUT/pd
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $R/tools/bench_dist.py > $OUT/drd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/tools/bench_dist.py > $OUT/dwr.log 2>&1
python3 $R/tools/traffic_dist_summary.py $OUT/rd $OUT/wr $SHA > $OUT/traffic_dist_pmc.json 2> $OUT/traffic_dist.err
rm -rf $OUT/rd $OUT/wr
ls -la $OUT
# the bench line again with this run's PMC summaries in place (bench.py quotes `traffic` only from files whose csrc_sha matches)
cp $OUT/traffic_pmc.json $R/profiles/r03_traffic_pmc.json; cp $OUT/fp32_traffic_pmc.json $R/profiles/r03_fp32_traffic_pmc.json
cp $OUT/traffic_dist_pmc.json $R/profiles/r03_traffic_dist_pmc.json
python3 $R/bench.py --dump-conv $OUT/conv_table_final.json > $OUT/bench_final.json 2> $OUT/bench_final.err
