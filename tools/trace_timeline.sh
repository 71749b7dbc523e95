#!/bin/bash
# GPU box: kernel trace (start / end timestamps, queue) of a few overlapped f16x2 train steps -> gpurun_out/r05/timeline_trace.csv
: "${GRAFT_REPO_ROOT:?run under gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
rm -rf /tmp/tl
rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-profile --no-companions > /tmp/tl.log 2>&1
f=$(find /tmp/tl -name "*kernel_trace.csv" | head -n 1)
python3 - "$f" $R/gpurun_out/r05/timeline_trace.csv <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# keep the last step only: from the last pack_input kernel on
idx = max(i for i, r in enumerate(rows) if "pack_input" in r["Kernel_Name"])
rows = rows[idx:]
t0 = int(rows[0]["Start_Timestamp"])
with open(sys.argv[2], "w") as f:
    f.write("start_us,end_us,queue,kernel\n")
    for r in rows:
        n = r["Kernel_Name"]
        n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:60]
        f.write("%.1f,%.1f,%s,%s\n" % ((int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3, r.get("Queue_Id", "?"), n))
print(len(rows), "kernels in the last step")
PY
