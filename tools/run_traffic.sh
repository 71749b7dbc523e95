#!/bin/bash
# GPU box: HBM traffic per kernel class from the TCC counters, two separate --pmc passes (MI355X_MICROARCH.md, HBM):
# bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024  (FETCH_SIZE counts 128-byte requests as 64 on gfx950).
# The summary carries the fingerprint of the kernel sources it was collected on (bench.py quotes it only on a match).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic
rm -rf $OUT; mkdir -p $OUT
SHA=$(python3 $R/bench.py --print-csrc-sha)
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/rd.log 2>&1
DML_OVERLAP_WGRAD=0 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-profile > $OUT/wr.log 2>&1
# steps seen by the profiler: 1 warm-up + 2 timed + 3 host-enqueue probes
python3 $R/tools/traffic_summary.py $OUT/rd $OUT/wr 6 $SHA > $OUT/traffic.json 2> $OUT/summary.err
rm -rf $OUT/rd $OUT/wr
