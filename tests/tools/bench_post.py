"""Micro-benchmark (GPU box): the kernels of the step after the path at 1024 x 2048 -- confusion matrix update and
OOD measures (device radix sort + rank statistics) -- next to their numpy restatements on one host core."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import numpy as np
import torch
import anom_utils, metrics
from oracle import ood_measures_ref as OR, metrics_ref as MR

rs = np.random.RandomState(4)
Hh, Ww = 1024, 2048
lab = rs.randint(0, 14, (Hh, Ww)).astype(np.int64)
conf = (rs.randn(Hh, Ww) + (lab >= 12) * 0.8).astype(np.float32)
pred = rs.randint(0, 14, (Hh, Ww)).astype(np.int64)
dl, dc, dp = torch.from_numpy(lab).cuda(), torch.from_numpy(conf).cuda(), torch.from_numpy(pred).cuda()

def timeit(fn, n=10):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n

m = metrics.StreamSegMetrics(14)
t = timeit(lambda: m.update(dl, dp))
print("confusion update 2.1 Mpx: %.1f us (%.0f GB/s of 16 B/px)" % (t * 1e6, 16 * Hh * Ww / t / 1e9))
t0 = time.perf_counter(); MR.fast_hist(lab.flatten(), pred.flatten(), 14); print("   numpy bincount: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
t = timeit(lambda: anom_utils.eval_ood_measure(dc, dl, [12, 13]))
print("OOD measures 2.1 Mpx (sort + ranks + 40 B D2H): %.2f ms" % (t * 1e3))
t0 = time.perf_counter(); r = OR.eval_ood_measure(conf, lab, [12, 13]); print("   numpy restatement: %.1f ms" % ((time.perf_counter() - t0) * 1e3), r)
print("   device:", anom_utils.eval_ood_measure(dc, dl, [12, 13]))
