"""Open-world scores computed on the device instead of on 134 MB/img host copies
(test_embedding.py:339-350,365,428-445; anomaly/eval_ood_traditional.py:301-305)."""
from __future__ import annotations

import numpy as np
import torch

from dmlnet import _lib


def _st(t):
    return torch.cuda.current_stream(t.device).cuda_stream


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("open-world scores run on the HIP path only (no CPU fallback)")


def argmax_msp(logits: torch.Tensor):
    """preds = argmax_k logits, scores = 1 - max softmax."""
    _need_cuda(logits)
    lib = _lib.load()
    logits = logits.contiguous().float()
    B, K, H, W = logits.shape
    preds = torch.empty((B, H, W), dtype=torch.int64, device=logits.device)
    msp = torch.empty((B, H, W), dtype=torch.float32, device=logits.device)
    _lib.check(lib.dml_argmax_msp(logits.data_ptr(), preds.data_ptr(), msp.data_ptr(), B, K, H, W, _st(logits)),
               "dml_argmax_msp")
    return preds, msp


def dissum_score(logits: torch.Tensor, clip: float = 1000.0, inclusive: bool = False) -> torch.Tensor:
    """-sum_k logit_k, clipped (`> clip` for the DeepLab driver, `>= clip` with clip=400 for anomaly/),
    min-max normalised per image."""
    _need_cuda(logits)
    lib = _lib.load()
    logits = logits.contiguous().float()
    B, K, H, W = logits.shape
    score = torch.empty((B, H, W), dtype=torch.float32, device=logits.device)
    work = torch.empty(2 * B, dtype=torch.float32, device=logits.device)
    _lib.check(lib.dml_dissum_score(logits.data_ptr(), score.data_ptr(), work.data_ptr(), B, K, H, W, float(clip),
                                    1 if inclusive else 0, _st(logits)), "dml_dissum_score")
    return score


def mean_prototype(shots) -> np.ndarray:
    """test_embedding.py:254-257."""
    return np.mean(np.asarray(shots, dtype=np.float64), axis=0)


def novel_relabel(preds: torch.Tensor, logits: torch.Tensor, feats: torch.Tensor, proto, thresh=-1.5,
                  new_label=16) -> torch.Tensor:
    """In place: preds[p] = new_label where -|f_p - proto|^2 > thresh and > max_k logit_k."""
    _need_cuda(logits)
    lib = _lib.load()
    logits, feats = logits.contiguous().float(), feats.contiguous().float()
    B, K, H, W = logits.shape
    C = feats.shape[-1]
    pr = torch.as_tensor(np.asarray(proto), dtype=torch.float32, device=logits.device).contiguous()
    assert preds.is_contiguous() and preds.dtype == torch.int64
    _lib.check(lib.dml_novel_relabel(feats.data_ptr(), logits.data_ptr(), pr.data_ptr(), preds.data_ptr(), B, C, K,
                                     H, W, float(thresh), int(new_label), _st(logits)), "dml_novel_relabel")
    return preds


def extract_prototype(features: torch.Tensor, labels_true: torch.Tensor, class_id: int, min_fraction: float = 0.05):
    """One shot of a novel-class prototype: the mean of `features_out` over the pixels labelled `class_id`, or None
    when the class covers less than `min_fraction` of the image -- the recipe the reference keeps commented out at
    test_embedding.py:413-425 (`features.cpu().numpy()[labels_true == 15]`, `np.mean(..., axis=0)`, json).  The
    reduction runs on the device; collect the returned lists and `json.dump` them as the reference does."""
    _need_cuda(features)
    lib = _lib.load()
    f = features.contiguous().float()
    C = f.shape[-1]
    f = f.view(-1, C)
    lab = labels_true.contiguous().view(-1)
    if lab.dtype != torch.int64 or lab.numel() != f.shape[0] or not lab.is_cuda:
        raise ValueError("labels_true must be an int64 CUDA tensor with one entry per pixel of `features`")
    sums = torch.empty(C, dtype=torch.float64, device=f.device)
    cnt = torch.empty(1, dtype=torch.int64, device=f.device)
    _lib.check(lib.dml_class_feature_sum(f.data_ptr(), lab.data_ptr(), f.shape[0], C, int(class_id), sums.data_ptr(),
                                         cnt.data_ptr(), _st(f)), "dml_class_feature_sum")
    n = int(cnt.item())
    if n == 0 or n / f.shape[0] <= min_fraction:
        return None
    return (sums / n).float().cpu().tolist()
