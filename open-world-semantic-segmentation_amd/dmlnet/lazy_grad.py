"""Deferred d(loss)/d(logits) of the DML loss.

With `fused_backward=True` the loss does not materialise its 16-plane gradient tensor (64 B per pixel written, read
again by the head's backward): its autograd backward returns a MARKER -- a one-element NaN tensor expanded to the
logits' shape -- and records here what the gradient is (labels, loss sums, alpha, upstream scalar).  The model's
backward recognises the marker by address and runs ONE kernel for loss gradient + distance-head gradient + the
transposed x4 upsample (dml_head_bwd_fused), or materialises the gradient with dml_loss_bwd when the fused kernel
does not cover the shape.

Validity: the marker stands for the gradient only if the loss is the SOLE consumer of the logits tensor, which is how
the reference's drivers use it (main_embedding.py:466-470: criterion(outputs, labels) -> backward).  If anything else
also consumed the logits, autograd would add the marker to that other gradient: the NaN payload then poisons the sum
(visible), and the engine refuses a non-marker gradient while a marker of the same forward is outstanding.  The flag is
therefore opt-in (utils.DMLLoss / utils.CrossEntropyLoss, set by this package's drivers and bench.py).
"""
from __future__ import annotations

import torch


class LazyLossGrad:
    __slots__ = ("cell", "labels", "sums", "gout", "alpha", "ignore_index", "n_images", "logits")

    def __init__(self, cell, labels, sums, gout, alpha, ignore_index, n_images, logits):
        self.cell, self.labels, self.sums, self.gout = cell, labels, sums, gout
        self.alpha, self.ignore_index, self.n_images, self.logits = alpha, ignore_index, n_images, logits


_pending = {}          # marker address -> LazyLossGrad


def issue(logits, labels, sums, gout, alpha, ignore_index, n_images):
    cell = torch.full((1,), float("nan"), dtype=torch.float32, device=logits.device)
    _pending[cell.data_ptr()] = LazyLossGrad(cell, labels, sums, gout, alpha, ignore_index, n_images, logits)
    return cell.expand(logits.shape)


def take(grad):
    """the record behind `grad` if it is an untouched marker, else None"""
    if grad is None or not _pending:
        return None
    if grad.dim() == 4 and all(s == 0 for s in grad.stride()):
        return _pending.pop(grad.data_ptr(), None)
    return None


def outstanding_for(logits) -> bool:
    return any(rec.logits is logits for rec in _pending.values())


def drop_for(logits):
    if logits is None or not _pending:
        return
    for k in [k for k, rec in _pending.items() if rec.logits is logits]:
        del _pending[k]
