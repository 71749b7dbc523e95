"""Mint tests/golden/g10_metrics.npz from the reference's StreamSegMetrics (authoring container only)."""
import contextlib, importlib.util, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_stream_metrics",
                                              "/root/reference/DeepLabV3Plus-Pytorch/metrics/stream_metrics.py")
sm = importlib.util.module_from_spec(spec)
spec.loader.exec_module(sm)
rs = np.random.RandomState(21)
n = 19                                    # the reference hard-codes self.n_classes = 19 (stream_metrics.py:29)
B, Hh, Ww = 3, 37, 53
lt = rs.randint(0, n, (B, Hh, Ww)).astype(np.int64)
lt[rs.rand(B, Hh, Ww) < 0.07] = 255
lt[:, :, :3] = 17                         # one class never predicted, one never present (nan paths)
lp = np.where(rs.rand(B, Hh, Ww) < 0.7, np.minimum(lt, n - 1), rs.randint(0, n - 2, (B, Hh, Ww))).astype(np.int64)
lp[lp == 5] = 6
lt[lt == 9] = 10
m = sm.StreamSegMetrics(n)
m.update(lt[:2], lp[:2])
m.update(lt[2:], lp[2:])
with contextlib.redirect_stdout(io.StringIO()):
    r = m.get_results()
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g10_metrics.npz"), lt=lt, lp=lp, n=n,
                    hist=m.confusion_matrix, overall=r["Overall Acc"], mean_acc=r["Mean Acc"], fw=r["FreqW Acc"],
                    miou=r["Mean IoU"], class_iou=np.array([r["Class IoU"][k] for k in range(n)]))
print(r["Overall Acc"], r["Mean IoU"], np.isnan(np.array(list(r["Class IoU"].values()))).sum())
