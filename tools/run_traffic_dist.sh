#!/bin/bash
# GPU box: HBM traffic of the standalone distance kernel (bench.py's hbm_kernel), two separate --pmc passes
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic_dist
rm -rf $OUT; mkdir -p $OUT
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/rd -- python3 $R/tools/bench_dist.py > $OUT/rd.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $OUT/wr -- python3 $R/tools/bench_dist.py > $OUT/wr.log 2>&1
python3 - <<PY > $OUT/traffic_dist.json
import csv, glob, json, os
def load(d, counter):
    tot, n = 0.0, 0
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "proto_dist_fwd_c16_kernel" in r["Kernel_Name"]:
                tot += float(r["Counter_Value"]); n += 1
    return tot, n
r, nr = load("$OUT/rd", "FETCH_SIZE"); w, nw = load("$OUT/wr", "WRITE_SIZE")
print(json.dumps({"kernel": "proto_dist_fwd_c16_kernel", "formula": "bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024, separate --pmc passes",
                  "launches": [nr, nw], "read_bytes_per_launch": 2 * r * 1024 / max(nr, 1), "write_bytes_per_launch": w * 1024 / max(nw, 1),
                  "bytes_per_launch": 2 * r * 1024 / max(nr, 1) + w * 1024 / max(nw, 1), "algorithmic_bytes_per_launch": 192 * 16 * 768 * 768}))
PY
rm -rf $OUT/rd $OUT/wr
cat $OUT/traffic_dist.json
