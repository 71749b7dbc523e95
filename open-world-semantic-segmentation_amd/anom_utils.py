"""The reference's pixel-level OOD measures (anomaly/anom_utils.py there) computed on the MI355X.

Same function names; the arguments are CUDA tensors where the reference takes numpy arrays, so that the
per-image score map (2 M pixels at 1024 x 2048) is sorted and ranked where the model left it instead of being
copied to the host and argsorted there (eval_ood_traditional.py:128-148, :566-569).  Three doubles come back.
No CPU fallback.
"""
import ctypes as C

import numpy as np
import torch

from dmlnet import _lib

recall_level_default = 0.95


def _measures(conf, seg_label, out_labels, mask, recall_level):
    lib = _lib.load()
    if not (isinstance(conf, torch.Tensor) and conf.is_cuda and conf.dtype == torch.float32):
        raise TypeError("scores must be a float32 CUDA tensor (there is no CPU fallback)")
    if not (seg_label.is_cuda and seg_label.dtype == torch.int64 and seg_label.numel() == conf.numel()):
        raise TypeError("labels must be an int64 CUDA tensor with one entry per score")
    conf, seg_label = conf.contiguous(), seg_label.contiguous()
    if mask is not None:
        mask = mask.to(torch.uint8).contiguous()
        if mask.numel() != conf.numel() or not mask.is_cuda:
            raise TypeError("mask must be a CUDA tensor with one entry per score")
    n = conf.numel()
    work = torch.empty(lib.dml_ood_workspace_bytes(n), dtype=torch.uint8, device=conf.device)
    res = torch.empty(5, dtype=torch.float64, device=conf.device)
    ol = (C.c_int64 * len(out_labels))(*[int(v) for v in out_labels])
    st = torch.cuda.current_stream(conf.device).cuda_stream
    _lib.check(lib.dml_ood_measures(conf.data_ptr(), seg_label.data_ptr(), mask.data_ptr() if mask is not None else None, n,
                                    ol, len(out_labels), float(recall_level), work.data_ptr(), work.numel(), res.data_ptr(),
                                    st), "dml_ood_measures")
    r = res.cpu().numpy()
    if r[3] == 0 or r[4] == 0:
        return None
    return float(r[0]), float(r[1]), float(r[2])


def get_measures(_pos, _neg, recall_level=recall_level_default):
    """(auroc, aupr, fpr) with `_pos` the scores of the positive class (anom_utils.py:68-78)"""
    pos, neg = _pos.reshape(-1), _neg.reshape(-1)
    scores = torch.cat([pos, neg]).to(torch.float32)
    labels = torch.zeros(scores.numel(), dtype=torch.int64, device=scores.device)
    labels[: pos.numel()] = 1
    res = _measures(-scores, labels, [1], None, recall_level)       # the kernel negates conf
    if res is None:
        raise ValueError("both classes need at least one sample")
    return res


def print_measures(auroc, aupr, fpr, method_name="Ours", recall_level=recall_level_default):
    print("\t\t\t\t" + method_name)
    print("FPR{:d}:\t\t\t{:.2f}".format(int(100 * recall_level), 100 * fpr))
    print("AUROC: \t\t\t{:.2f}".format(100 * auroc))
    print("AUPR:  \t\t\t{:.2f}".format(100 * aupr))


def get_and_print_results(out_score, in_score, num_to_avg=1):
    """anom_utils.py:96-105: one evaluation, returned as means"""
    auroc, aupr, fpr = get_measures(out_score, in_score)
    return float(np.mean([auroc])), float(np.mean([aupr])), float(np.mean([fpr]))


def eval_ood_measure(conf, seg_label, out_labels, mask=None, recall_level=recall_level_default):
    """eval_ood_traditional.py:128-148: scores -conf, positives = pixels labelled with one of `out_labels`
    (cfg.OOD.out_labels there); None when the image has no OOD pixel or only OOD pixels."""
    res = _measures(conf, seg_label, out_labels, mask, recall_level)
    if res is None:
        print("This image does not contain any OOD pixels or is only OOD.")
    return res
