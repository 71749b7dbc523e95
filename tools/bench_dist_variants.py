"""GPU box: the distance-head kernels under their build switches (child processes, the switches are read once):
DML_DIST_NT = non-temporal stores of logits / features, DML_UPS_STAGED = LDS-staged x4 upsample kernel."""
import json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, json
sys.path[:0] = [%r, %r]
import torch, bench
r = bench.bench_distance_kernel(16, 768, torch.device("cuda", 0))
print(json.dumps({"proto_dist_fwd_ms": r["ms"], "frac": r["frac"], "step": [(e["kernel"][:28], round(e["ms"], 4), round(e["frac"], 3)) for e in r["step_path"]]}))
''' % (ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"))
for nt in ("0", "1"):
    for staged in ("0", "1"):
        env = dict(os.environ, DML_DIST_NT=nt, DML_UPS_STAGED=staged)
        for rep in range(2):
            out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            print("NT=%s STAGED=%s run %d: %s" % (nt, staged, rep, line[-1] if line else out.stderr[-500:]), flush=True)
