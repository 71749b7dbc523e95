#!/bin/bash
# GPU box: the artefacts kept under profiles/ for round 6 -- the default bench line (f16x2 headline + bf16 / exact-fp32 / three-term
# companions + cpu baseline + per-class two-roof table), rocprofv3 kernel stats of the headline command (weight gradients on the side
# stream as timed, and serial), of the bf16 and exact-fp32 steps (serial), PMC HBM traffic per kernel and per conv shape class
# (f16x2 and bf16), SQ counters per kernel template of the headline.
#   gpurun -- bash tools/run_r06_profiles.sh v1   ->  gpurun_out/final_r06_v1/
# Every profiled program is `python3 bench.py ...` directly after `--` (no env / bash -c hop), counters in their own runs.
set -euo pipefail
TAG=${1:?tag}
: "${GRAFT_REPO_ROOT:?run under gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final_r06_$TAG
rm -rf $OUT; mkdir -p $OUT
SHA=$(python3 $R/bench.py --print-csrc-sha)
valid_json() { python3 -c "import json,sys; json.load(open(sys.argv[1]))" "$1" || { echo "INVALID JSON: $1" >&2; exit 1; }; }
last_json() { tail -n 1 "$1" > "$1.line"; valid_json "$1.line"; rm -f "$1.line"; }
stats() {      # stats <dir> <dest.csv>
  local f; f=$(find "$1" -name "*kernel_stats.csv" | head -n 1)
  [ -n "$f" ] || { echo "no kernel_stats.csv under $1" >&2; exit 1; }
  cp "$f" "$2"; rm -rf "$1"
}
HB="$R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-profile --no-companions"
B="$R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-profile --no-companions"
python3 $R/bench.py --dump-conv $OUT/conv_table.json > $OUT/bench.json 2> $OUT/bench.err
last_json $OUT/bench.json
# per-kernel time of the headline command: weight gradients on the side stream (as timed) and serial (un-contended)
for mode in 1 0; do
  DML_OVERLAP_WGRAD=$mode rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/p$mode -- python3 $HB > $OUT/p$mode.log 2>&1
  stats $OUT/p$mode $OUT/kernel_stats_overlap$mode.csv
done
DML_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pb -- python3 $HB --dtype bf16 > $OUT/pb.log 2>&1
stats $OUT/pb $OUT/bf16_kernel_stats_serial.csv
DML_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pf -- python3 $B --dtype f32 > $OUT/pf.log 2>&1
stats $OUT/pf $OUT/fp32_kernel_stats_serial.csv
# kernel stats of the SAME short serial command the counter passes use (durations next to the SQ counters)
DML_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/ps -- python3 $B > $OUT/ps.log 2>&1
stats $OUT/ps $OUT/kernel_stats_pmc_cmd.csv
# HBM traffic, separate --pmc passes (MI355X_MICROARCH.md: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024): headline, then bf16
export DML_BENCH_OPLOG=$OUT/oplog_f16x2.json
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $B > $OUT/rd.log 2>&1
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $B > $OUT/wr.log 2>&1
python3 $R/tools/pmc_by_class.py traffic $OUT/rd $OUT/wr $OUT/oplog_f16x2.json $SHA > $OUT/traffic_pmc.json 2> $OUT/traffic.err
valid_json $OUT/traffic_pmc.json
rm -rf $OUT/rd $OUT/wr
# SQ counters (8 SQ slots), one pass
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS --kernel-trace --output-format csv -d $OUT/sq -- python3 $B > $OUT/sq.log 2>&1
python3 $R/tools/pmc_by_class.py sq $OUT/sq $OUT/oplog_f16x2.json $OUT/kernel_stats_pmc_cmd.csv $SHA > $OUT/conv_pmc_sq.json 2> $OUT/sq.err
valid_json $OUT/conv_pmc_sq.json
rm -rf $OUT/sq
export DML_BENCH_OPLOG=$OUT/oplog_bf16.json
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $B --dtype bf16 > $OUT/brd.log 2>&1
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $B --dtype bf16 > $OUT/bwr.log 2>&1
python3 $R/tools/pmc_by_class.py traffic $OUT/rd $OUT/wr $OUT/oplog_bf16.json $SHA > $OUT/bf16_traffic_pmc.json 2> $OUT/btraffic.err
valid_json $OUT/bf16_traffic_pmc.json
rm -rf $OUT/rd $OUT/wr
unset DML_BENCH_OPLOG
# the standalone distance kernel: duration from the kernel trace + its PMC traffic
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/pd -- python3 $R/tools/bench_dist.py > $OUT/pd.log 2>&1
stats $OUT/pd $OUT/dist_kernel_stats.csv
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $R/tools/bench_dist.py > $OUT/drd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/tools/bench_dist.py > $OUT/dwr.log 2>&1
python3 $R/tools/traffic_dist_summary.py $OUT/rd $OUT/wr $SHA > $OUT/traffic_dist_pmc.json 2> $OUT/traffic_dist.err
valid_json $OUT/traffic_dist_pmc.json
rm -rf $OUT/rd $OUT/wr
# the non-headline configurations of BASELINE.json on the same sources: #5 open-world inference at 1024 x 2048 (f16x2 / bf16, batch 1 / 4),
# #2 forward-only 768 x 768 x 8 bf16 -- each with the conv roofline of its forward plan
for cfg in "infer f16x2 1" "infer f16x2 4" "infer bf16 1" "infer bf16 4" "fwd bf16 8" "fwd f16x2 8"; do
  set -- $cfg
  python3 $R/bench.py --mode $1 --dtype $2 --batch $3 --steps 20 --warmup 5 2> /dev/null | grep "^{" > $OUT/bench_$1_$2_b$3.json
  valid_json $OUT/bench_$1_$2_b$3.json
done
# the bench line again with this run's PMC summaries in place (bench.py quotes `traffic` only from files whose csrc_sha matches)
cp $OUT/traffic_pmc.json $R/profiles/r06_traffic_pmc.json; cp $OUT/bf16_traffic_pmc.json $R/profiles/r06_bf16_traffic_pmc.json
cp $OUT/traffic_dist_pmc.json $R/profiles/r06_traffic_dist_pmc.json
python3 $R/bench.py --dump-conv $OUT/conv_table_final.json > $OUT/bench_final.json 2> $OUT/bench_final.err
last_json $OUT/bench_final.json
find $OUT -name "*.log" -size +200k -delete
ls -la $OUT
