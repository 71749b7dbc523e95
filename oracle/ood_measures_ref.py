"""CPU restatement of the reference's pixel-level OOD measures -- TEST INFRASTRUCTURE ONLY
(only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it).

Restates anomaly/anom_utils.py:25-78 (fpr_and_fdr_at_recall, get_measures) and the caller
anomaly/eval_ood_traditional.py:128-148 (eval_ood_measure: scores = -conf, positives = pixels whose label is in
cfg.OOD.out_labels) -- SURVEY 8(f) rank 3, "the step after the path".  AUROC / AUPR are sklearn's
roc_auc_score / average_precision_score in the reference (third-party, not under /root/reference; the environment
pins scikit-learn through requirements.txt); their published definitions restated here in the rank form the device
kernels use:
  AUROC = sum over positives of (#negatives below + #negatives not above) / (2 P N)        (trapezoid with ties)
  AP    = sum over distinct positive scores t of (R(t) - R(next higher t)) * tp(t) / (tp(t) + fp(t)),
          R = tp / P, tp/fp = positives / negatives with score >= t
  FPR@r = fps[cutoff] / N at the threshold whose recall is nearest to r, scanning thresholds from the lowest that
          reaches full recall upwards and taking the first minimum (anom_utils.py:58-66).
Pinned by tests/golden/g11_ood_measures.npz minted from anom_utils.get_measures itself (tests/tools/mint_golden_metrics.py).
"""
import numpy as np


def get_measures(pos, neg, recall_level=0.95):
    pos = np.sort(np.asarray(pos, dtype=np.float64))
    neg = np.sort(np.asarray(neg, dtype=np.float64))
    P, N = len(pos), len(neg)
    lo = np.searchsorted(neg, pos, side="left")
    hi = np.searchsorted(neg, pos, side="right")
    auroc = float(int(lo.sum()) + int(hi.sum())) / (2.0 * P * N)
    # distinct positive values, ascending; tp / fp at each as a threshold (score >= t)
    vals, first, cnt = np.unique(pos, return_index=True, return_counts=True)
    tp = P - first
    fp = N - np.searchsorted(neg, vals, side="left")
    r = tp.astype(np.float64) / P
    rprev = (tp - cnt).astype(np.float64) / P
    aupr = float(np.sum((r - rprev) * (tp.astype(np.float64) / (tp + fp).astype(np.float64))))
    # FPR at the recall level: per recall plateau the candidate is its lowest threshold (largest index in the
    # reference's decreasing-score order); the plateau of full recall is cut at its first threshold
    fp_cand = np.empty(len(vals))
    fp_cand[0] = fp[0]                                        # lowest positive value: first threshold with tp == P
    if len(vals) > 1:
        fp_cand[1:] = N - np.searchsorted(neg, vals[:-1], side="right")     # negatives above the next lower positive
    d = np.abs(r - recall_level)
    best = np.flatnonzero(d == d.min())
    k = best[np.argmax(r[best])]                              # ties: the higher recall comes first in the reference's scan
    return auroc, aupr, float(fp_cand[k]) / N


def eval_ood_measure(conf, seg_label, out_labels, mask=None, recall_level=0.95):
    if mask is not None:
        conf, seg_label = conf[mask], seg_label[mask]
    out = np.isin(seg_label, out_labels)
    if out.sum() == 0 or (~out).sum() == 0:
        return None
    return get_measures(-conf[out], -conf[~out], recall_level)
