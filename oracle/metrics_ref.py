"""CPU restatement of the reference's streaming segmentation metrics -- TEST INFRASTRUCTURE ONLY
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Restates DeepLabV3Plus-Pytorch/metrics/stream_metrics.py:33-35 (update), :49-55 (_fast_hist: bincount of
n*true + pred over pixels with 0 <= true < n) and :57-83 (overall / mean accuracy, IoU, frequency-weighted
accuracy from the confusion matrix) -- SURVEY 8(f) rank 3, "the step after the path".
Pinned by tests/golden/g10_metrics.npz, minted from the reference class itself (tests/tools/mint_golden_metrics.py).
"""
import numpy as np


def fast_hist(label_true, label_pred, n):
    mask = (label_true >= 0) & (label_true < n)
    return np.bincount(n * label_true[mask].astype(int) + label_pred[mask], minlength=n ** 2).reshape(n, n)


def results(hist):
    hist = hist.astype(np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        acc = np.diag(hist).sum() / hist.sum()
        acc_cls = np.nanmean(np.diag(hist) / hist.sum(axis=1))
        iu = np.diag(hist) / (hist.sum(axis=1) + hist.sum(axis=0) - np.diag(hist))
        mean_iu = np.nanmean(iu)
        freq = hist.sum(axis=1) / hist.sum()
        fwavacc = (freq[freq > 0] * iu[freq > 0]).sum()
    return {"Overall Acc": acc, "Mean Acc": acc_cls, "FreqW Acc": fwavacc, "Mean IoU": mean_iu,
            "Class IoU": dict(zip(range(hist.shape[0]), iu))}
