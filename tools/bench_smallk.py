"""GPU box: the short-K-loop 1x1 convolutions (forward with statistics and data gradient) under DML_CONV_SMALLK (child
processes: the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
shapes = "16,192,192,64,256,1,1;16,192,192,256,64,1,1;16,96,96,128,512,1,1;16,96,96,512,128,1,1;16,48,48,256,1024,1,1;16,48,48,1024,256,1,1;16,48,48,256,256,3,1"
for val in ("0", "256", "1024"):
    env = dict(os.environ, DML_CONV_SMALLK=val, BENCH_SHAPES=shapes)
    for rep in range(2):
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "bench_conv.py"), "fwd"], env=env, capture_output=True, text=True)
        print("SMALLK=%s run %d" % (val, rep))
        print("\n".join(l for l in out.stdout.splitlines() if l.startswith("B16")) or out.stderr[-800:], flush=True)
