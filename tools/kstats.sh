#!/bin/bash
# GPU box: rocprofv3 kernel-trace statistics of one bench.py command (serial: weight gradients on the main stream), summary on stdout
# and the csv under gpurun_out/<tag>_kernel_stats.csv.     gpurun -- bash tools/kstats.sh <tag> <bench.py args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
rm -rf $OUT.d; mkdir -p $OUT.d
export DML_OVERLAP_WGRAD=${DML_OVERLAP_WGRAD:-0}
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT.d -- python3 $R/bench.py "$@" > $OUT.log 2>&1
find $OUT.d -name "*kernel_stats.csv" -exec cp {} ${OUT}_kernel_stats.csv \;
rm -rf $OUT.d
python3 - "$OUT" "$@" <<'PY'
import csv, sys, json
out = sys.argv[1]
rows = list(csv.DictReader(open(out + "_kernel_stats.csv")))
steps = 1
a = sys.argv[2:]
for i, v in enumerate(a):
    if v == "--steps": steps = int(a[i + 1])
    if v == "--warmup": steps += int(a[i + 1])
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("kernel time per step %.2f ms (%d steps incl. warm-up)" % (tot / 1e6 / steps, steps))
for r in rows[:32]:
    print("%-86s n/step %6.1f  %8.3f ms/step  avg %8.1f us  %5.1f%%" % (r["Name"][:86], float(r["Calls"]) / steps,
          float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
PY
