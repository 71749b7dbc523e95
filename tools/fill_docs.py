"""Authoring container: fill the R5_* placeholders of README.md / DESIGN.md from a bench.py JSON line (profiles/r05_bench_v2.json).
    python tools/fill_docs.py profiles/r05_bench_v2.json"""
import json, sys, re, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
vals = {
    "R5_VALUE": "%.1f" % d["value"], "R5_MS": "%.1f" % d["ms_per_step"],
    "R5_CONV_MS": "%.1f" % r["conv_ms_per_step"], "R5_CONV_TF": "%.1f" % r["achieved"], "R5_CONV_FRAC": "%.1f" % (100 * r["frac"]),
    "R5_CONV_F16": "%.0f" % (3 * r["achieved"]),
    "R5_TRAFFIC": ("%.2f" % r["traffic_over_algorithmic"]) if r.get("traffic_over_algorithmic") else "n/a",
    "R5_BF16_FRAC": "%.1f" % (100 * d["bf16_companion"]["roofline"]["frac"]), "R5_BF16": "%.1f" % d["bf16_companion"]["value"],
    "R5_F32_FRAC": "%.1f" % (100 * d["fp32_exact_companion"]["roofline"]["frac"]), "R5_F32": "%.1f" % d["fp32_exact_companion"]["value"],
    "R5_DIST": "%.1f" % (100 * d["hbm_kernel"]["frac"]),
    "R5_CPU_CORES": "%d" % d["cpu_baseline"]["cores"], "R5_CPU": "%.2f" % d["cpu_baseline"]["value"],
}
for f in ("README.md", "DESIGN.md"):
    p = os.path.join(ROOT, f)
    s = open(p).read()
    for k in sorted(vals, key=len, reverse=True):
        s = s.replace(k, vals[k])
    left = re.findall(r"R5_[A-Z0-9_]+", s)
    assert not left, (f, left)
    open(p, "w").write(s)
print(vals)
