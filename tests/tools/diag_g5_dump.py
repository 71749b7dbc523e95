"""GPU box diagnostic (test infrastructure): dump / compare every parameter gradient of the g5 f16x2 train step under two builds.
   DML_LIB_PATH=<build> python3 tests/tools/diag_g5_dump.py dump <file.npz> ;  python3 tests/tools/diag_g5_dump.py cmp a.npz b.npz"""
import os, sys
import numpy as np
if sys.argv[1] == "cmp":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    names = [str(n) for n in a["names"]]
    for i, k in enumerate(names):
        x, y = a["g%d" % i].astype(np.float64), b["g%d" % i].astype(np.float64)
        d = np.abs(x - y).max() / max(np.abs(x).max(), 1e-30)
        if d > 2e-5:
            print("%-46s max|a-b|/max|a| %.2e   (|a| max %.3e)" % (k, d, np.abs(x).max()))
        if d > 1e-2 and x.ndim == 1:          # a BatchNorm parameter: which channels, by how much (a flipped ReLU mask element = one channel)
            e = np.abs(x - y) / np.abs(x).max()
            top = np.argsort(-e)[:4]
            print("    channels > 1e-3: %d of %d; top: %s" % ((e > 1e-3).sum(), e.size,
                  ", ".join("ch %d a %.5e b %.5e" % (c, x[c], y[c]) for c in top)))
    sys.exit(0)
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests")]
import test_gpu_model as TM
import utils
m = TM.build(fp32_products="f16x2")
img, lab = TM.g5_inputs()
lg, ctr, ft = m(img)
loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(lg, lab, ft)
loss.backward()
out = {"names": np.array([k for k, _ in m.named_parameters()])}
for i, (k, p) in enumerate(m.named_parameters()):
    out["g%d" % i] = p.grad.detach().float().cpu().numpy()
np.savez(sys.argv[2], **out)
