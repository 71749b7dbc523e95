#!/bin/bash
# GPU box: per-shape conv table with and without the BN-backward sums fused into the data-gradient epilogues
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/fuse_ab
rm -rf $OUT; mkdir -p $OUT
python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-conv $OUT/conv_fused.json > $OUT/fused.json 2>/dev/null
DML_FUSE_BN_REDUCE=0 python3 $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --dump-conv $OUT/conv_unfused.json > $OUT/unfused.json 2>/dev/null
python3 - <<PY
import json,collections
for tag in ("fused","unfused"):
    d=json.load(open("$OUT/%s.json"%tag)); print(tag, d["value"], d["ms_per_step"])
    t=json.load(open("$OUT/conv_%s.json"%tag))
    agg=collections.OrderedDict()
    for r in t:
        if r["kind"]!="dgrad": continue
        k=(r["Hi"],r["C"],r["N"],r["R"],r["dil"]); a=agg.setdefault(k,[0,0.0]); a[0]+=1; a[1]+=r["ms"]
    for k,(n,ms) in sorted(agg.items(), key=lambda kv:-kv[1][1])[:10]:
        print("   ",k,n,"%.3f ms  each %.1f us"%(ms,ms/n*1e3))
PY
