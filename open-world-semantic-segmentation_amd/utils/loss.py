"""DML losses on the HIP path.

CrossEntropyLoss mirrors the LIVE behaviour of the reference's utils/loss.py:25-42: CE(ignore_index, mean
over valid pixels) / n; alpha/beta/gamma are accepted and -- exactly as in the reference, whose VAR / Inter /
Center terms sit behind an early `return` (loss.py:42-82) -- do not change the value.
DMLLoss is the live DCE + VL loss of anomaly/models/models.py:42-78: CE/n + alpha * VAR/n.
Both run dml_loss_fwd / dml_loss_bwd (one fused pass each instead of a per-image per-class Python loop).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from dmlnet import _lib


def _stream(t):
    return torch.cuda.current_stream(t.device).cuda_stream


_COUNTS = {}


def _image_count(device, n):
    """a cached one-element fp64 device tensor holding `n` (device-to-device copies never wait for the stream)"""
    key = (str(device), int(n))
    t = _COUNTS.get(key)
    if t is None:
        t = torch.full((1,), float(n), dtype=torch.float64, device=device)
        _COUNTS[key] = t
    return t


class _DMLLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logit, target, alpha, ignore_index, group, fused=False):
        if not logit.is_cuda:
            raise RuntimeError("the DML loss runs on the HIP path only (no CPU fallback)")
        lib = _lib.load()
        logit = logit.contiguous().float()
        target = target.contiguous().long()
        B, K, H, W = logit.shape
        if target.shape != (B, H, W):
            raise RuntimeError("target shape %s does not match logits %s" % (tuple(target.shape), tuple(logit.shape)))
        sums = torch.empty(5, dtype=torch.float64, device=logit.device)
        part = torch.empty(_lib.LOSS_BLOCKS * 4, dtype=torch.float32, device=logit.device)
        st = _stream(logit)
        _lib.check(lib.dml_loss_fwd(logit.data_ptr(), target.data_ptr(), sums.data_ptr(), part.data_ptr(), B, K, H, W,
                                    int(ignore_index), st), "dml_loss_fwd")
        n_images = float(B)
        if group is not None:
            import torch.distributed as dist
            # the reference normalises by the size of the gathered batch (utils/loss.py:38-41 on DataParallel's
            # gather); shards may be uneven (parallel.shard_range spreads a remainder), so the image count travels
            # with the sums (sums[4], read by the kernels on the device: no host sync) instead of assuming B * world
            # (not `sums[4] = float(B)`: assigning a Python scalar to a CUDA element copies from pageable host memory and BLOCKS the
            # host until the stream has drained -- the whole forward, 26 ms per step at 16 x 768 x 768 -- tools/probe_allreduce_host_block.py)
            sums[4:5].copy_(_image_count(logit.device, B), non_blocking=True)
            dist.all_reduce(sums, group=group if group is not True else None)
            n_images = 0.0
        loss = torch.empty((), dtype=torch.float32, device=logit.device)
        _lib.check(lib.dml_loss_finalize(sums.data_ptr(), loss.data_ptr(), float(alpha), n_images, st),
                   "dml_loss_finalize")
        ctx.save_for_backward(logit, target, sums)
        ctx.cfg = (float(alpha), int(ignore_index), n_images, bool(fused))
        return loss

    @staticmethod
    def backward(ctx, gout):
        logit, target, sums = ctx.saved_tensors
        alpha, ignore_index, n_images, fused = ctx.cfg
        if fused:
            # no gradient tensor: a marker the model's backward resolves in its fused head kernel (dmlnet/lazy_grad.py)
            from dmlnet import lazy_grad
            return lazy_grad.issue(logit, target, sums, gout.contiguous().float(), alpha, ignore_index, n_images), \
                None, None, None, None, None
        lib = _lib.load()
        B, K, H, W = logit.shape
        g = torch.empty_like(logit)
        gout = gout.contiguous().float()
        _lib.check(lib.dml_loss_bwd(logit.data_ptr(), target.data_ptr(), sums.data_ptr(), gout.data_ptr(),
                                    g.data_ptr(), B, K, H, W, ignore_index, alpha, n_images, _stream(logit)),
                   "dml_loss_bwd")
        return g, None, None, None, None, None


class DMLLoss(nn.Module):
    """loss = CE/n + alpha*VAR/n, VAR = sum_i (1/HW_i) sum_{valid p} dist^2(p, own prototype).

    `sync` (True or a process group): sums are all-reduced so that every rank sees the loss of the global
    batch -- the value nn.DataParallel's gather gives the reference (main_embedding.py:466-467)."""

    def __init__(self, alpha=0.01, ignore_index=-1, sync=None, fused_backward=False):
        super().__init__()
        self.alpha, self.ignore_index, self.sync = alpha, ignore_index, sync
        # True: d(loss)/d(logits) is not materialised; valid when this loss is the only consumer of `logit`, as in the
        # reference's drivers (dmlnet/lazy_grad.py)
        self.fused_backward = fused_backward

    def forward(self, logit, target, features_in=None):
        return _DMLLossFn.apply(logit, target, self.alpha, self.ignore_index, self.sync, self.fused_backward)


class CrossEntropyLoss(nn.Module):
    def __init__(self, alpha=0, beta=0, gamma=0, size_average=True, ignore_index=255, sync=None, fused_backward=False):
        super().__init__()
        self.alpha, self.beta, self.gamma = alpha, beta, gamma
        self.ignore_index, self.size_average, self.sync = ignore_index, size_average, sync
        self.fused_backward = fused_backward
        if not size_average:
            raise NotImplementedError("CrossEntropyLoss(size_average=False) (the reference's utils/loss.py:26,35 sum reduction) is "
                                      "never used by the embedding drivers and has no HIP kernel here")

    def forward(self, logit, target, features_in=None):
        return _DMLLossFn.apply(logit, target, 0.0, self.ignore_index, self.sync, self.fused_backward)
