#!/bin/bash
# GPU box: A/B of the 128 x 64 wave-tile variant (DML_CONV_WW = K threshold) on the whole step and per class
R=$GRAFT_REPO_ROOT
for ww in 0 2304 1024 512 256; do
  DML_CONV_WW=$ww python3 $R/bench.py --no-cpu-baseline --no-fp32-companion > $R/gpurun_out/ab_ww_$ww.json 2>/dev/null
  python3 - <<PY
import json
d=json.load(open("$R/gpurun_out/ab_ww_$ww.json"))
r=d["roofline"]
print("WW=$ww: %.1f img/s  %.2f ms | conv %.2f ms (igemm %.2f wgrad %.2f) frac %.4f" % (d["value"], d["ms_per_step"], r["conv_ms_per_step"], r["igemm_ms_per_step"], r["wgrad_ms_per_step"], r["frac"]))
for c in r["classes"]:
    if c["kind"]!="wgrad" and ("256->256 @48" in c["shape"] or "1024->256 @48" in c["shape"] or "256->1024 @48" in c["shape"] or "304->256" in c["shape"] or "2048->256 @48x48 d12" in c["shape"] or "512->512" in c["shape"] or "512->2048" in c["shape"] or "2048->512" in c["shape"]):
        print("   %-6s %-28s %3d launches %.3f ms %6.0f TF" % (c["kind"], c["shape"], c["launches"], c["ms"], c["tflops"]))
PY
done
