"""GPU box diagnostic (test infrastructure): the g5 full train step in f16x2 mode against the fixture's gradient checksums, worst
tensors first.      DML_LIB_PATH=<build> python3 tests/tools/diag_g5_f16x2.py [products]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests")]
import numpy as np
import test_gpu_model as TM
import helpers as H
import utils

products = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
g = H.load_golden("g5_full_train")
for rep in range(2):
    m = TM.build(fp32_products=products)
    img, lab = TM.g5_inputs()
    lg, ctr, ft = m(img)
    loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(lg, lab, ft)
    loss.backward()
    dev = []
    for (k, p), cs in zip(m.named_parameters(), g["grad_checksums"]):
        got = H.checksum(p.grad)
        dev.append((float(np.max(np.abs(got[1:] - cs[1:]) / np.abs(cs[1:]))), k))
    dev.sort(reverse=True)
    print("rep %d loss %.6f (fixture %.6f); worst checksum deviations: %s" % (rep, loss.item(), float(g["loss"]),
          ", ".join("%s %.1e" % (k, d) for d, k in dev[:6])), "| > 2e-3: %d" % sum(d > 2e-3 for d, _ in dev), flush=True)
