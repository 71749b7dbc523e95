"""Mint tests/golden/g10_metrics.npz from the reference's StreamSegMetrics (authoring container only)."""
import contextlib, importlib.util, io, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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


# ---------------------------------------------------------------------------------------------------------------------
# G11: pixel-level OOD measures from anomaly/anom_utils.py (get_measures -> sklearn roc_auc_score,
# average_precision_score, fpr_and_fdr_at_recall), as called by eval_ood_traditional.py:128-148
# ---------------------------------------------------------------------------------------------------------------------
def mint_ood():
    spec = importlib.util.spec_from_file_location("ref_anom_utils", "/root/reference/anomaly/anom_utils.py")
    au = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(au)
    rs = np.random.RandomState(33)
    cases = {}
    # continuous scores, overlapping classes
    cases["cont"] = (rs.randn(700).astype(np.float32) + 1.0, rs.randn(5000).astype(np.float32))
    # heavy ties (scores on a coarse grid, as after clipping / normalisation) incl. equal values across classes
    cases["ties"] = ((rs.randint(0, 12, 900) / 8.0).astype(np.float32), (rs.randint(-4, 9, 3000) / 8.0).astype(np.float32))
    # few positives, separable
    cases["sep"] = (rs.rand(17).astype(np.float32) + 2.0, rs.rand(2000).astype(np.float32))
    # inverted (AUROC < 0.5), negatives above all positives, clipped plateau at the top
    n_ = rs.randn(1500).astype(np.float32) + 1.5
    n_[n_ > 2.0] = 2.0
    cases["inv"] = (np.minimum(rs.randn(400).astype(np.float32), 2.0), n_)
    # recall-level tie structure: exactly 20 positives (0.95 = 19/20)
    cases["r20"] = (np.arange(20, dtype=np.float32) / 4.0, (rs.randint(0, 24, 800) / 4.0 - 0.5).astype(np.float32))
    out = {}
    for k, (pos, neg) in cases.items():
        a, p, f = au.get_measures(pos, neg)
        out[k + "_pos"], out[k + "_neg"], out[k + "_res"] = pos, neg, np.array([a, p, f], dtype=np.float64)
        print(k, a, p, f)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "g11_ood_measures.npz"), **out)


mint_ood()
