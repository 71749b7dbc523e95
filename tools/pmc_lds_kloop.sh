#!/bin/bash
# GPU box: LDS counters of the two-plane forward kernel on the ASPP 3x3 (long K) and on the layer3 1x1 -> gpurun_out/r05/pmc_lds_kloop.txt
: "${GRAFT_REPO_ROOT:?run under gpurun}"
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r05
export BENCH_SHAPES="16,48,48,2048,256,3,12;16,48,48,256,1024,1,1"
rm -rf /tmp/pl
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES --kernel-trace --output-format csv -d /tmp/pl -- python3 $R/tools/bench_h2.py fwd only=h2 > /tmp/pl.log 2>&1
python3 - /tmp/pl $R/gpurun_out/r05/pmc_lds_kloop.txt <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
out = open(sys.argv[2], "w")
if not f:
    out.write("no counter file; log tail:\n" + open("/tmp/pl.log").read()[-2000:]); sys.exit(0)
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"][:70]
    if "conv_ws_kernel" not in k: continue
    key = (k, r.get("Grid_Size", ""))
    acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
    n[key] += 1
for key, c in acc.items():
    out.write("%s grid %s\n" % key)
    for name, v in sorted(c.items()):
        out.write("   %-24s %.4g\n" % (name, v))
    if "SQ_LDS_IDX_ACTIVE" in c and c["SQ_LDS_IDX_ACTIVE"]:
        out.write("   bank conflict cycles / LDS active cycles = %.3f\n" % (c.get("SQ_LDS_BANK_CONFLICT", 0) / c["SQ_LDS_IDX_ACTIVE"]))
PY
cat $R/gpurun_out/r05/pmc_lds_kloop.txt | head -40
