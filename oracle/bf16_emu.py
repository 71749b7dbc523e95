"""CPU ORACLE, bf16-storage mode (test infrastructure, NOT product code).

The reference computes in fp32 (network/utils.py:84-118 of /root/reference/DeepLabV3Plus-Pytorch); the HIP
path's throughput mode stores activations, activation gradients and the compute copies of the weights as bf16 and
accumulates in fp32.  `emulate_bf16_storage(model)` turns an instance of the pinned oracle model
(oracle/dmlnet_ref.py) into a model of THAT arithmetic: the reference's graph and formulas stay as they are, and a
round-to-nearest-even bf16 quantisation is inserted at every point where the HIP plan stores a tensor:

  forward                                                   backward
  ----------------------------------------------------------------------------------------------------------
  conv input (z of the previous unit, packed image)  q      its gradient (dgrad result)                    q
  compute copy of every conv weight                  q      weight gradient                                fp32
  conv output y: statistics from the fp32            -      d(loss)/dy (bn_bwd_apply result)               q
    accumulators, then stored                        q
  BN + residual + ReLU result z                      q      dz                                             q
  bottleneck output (residual operand of the next)   q      (same tensor)
  ASPP output before / after the x4 bilinear         q      both gradients                                 q
  ASPP image-pooling branch (BN over B samples):     fp32 storage in / inside the unit, bf16 out (training plans)
  embedding (final 1x1 conv + bias): fp32            -      its gradient (bilinear_bwd result)             q
  upsample + distances + loss: fp32                  -      fp32

Eval mode (running statistics) has no stored y: BatchNorm + residual + ReLU run on the fp32 accumulators in the conv
epilogue, only z is rounded.

What is NOT modelled: the summation order inside the fp32 accumulations (a 1-ulp difference before a rounding point can
flip that bf16 value).  Gradients with several producers: here every producer's term is rounded (the conv input
quantisation's backward) and the fp32 sum is rounded again; the HIP plan sums the unrounded terms in an fp32 staging
tensor and rounds once (DmlConvDesc.acc32) -- one rounding fewer per producer, i.e. the plan is the more exact of the two.

Parity status: the underlying model is PINNED (see dmlnet_ref.py); the quantisation points restate DESIGN.md section 2
and csrc/bn.hip / conv_igemm.hip of this repository -- they describe the implementation under test, not the reference,
which has no reduced-precision mode.
"""
from __future__ import annotations

import types

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dmlnet_ref as O


def q(t: torch.Tensor) -> torch.Tensor:
    """round-to-nearest-even to bf16, returned as fp32 (v_cvt_pk_bf16_f32 semantics)"""
    return t.to(torch.bfloat16).to(t.dtype)


class _QBoth(torch.autograd.Function):
    """a stored bf16 tensor whose gradient is stored as bf16 too"""

    @staticmethod
    def forward(ctx, x):
        return q(x)

    @staticmethod
    def backward(ctx, g):
        return q(g)


class _QFwd(torch.autograd.Function):
    """bf16 value, gradient passed through (weights: the master copy receives the fp32 weight gradient)"""

    @staticmethod
    def forward(ctx, x):
        return q(x)

    @staticmethod
    def backward(ctx, g):
        return g


class _QGrad(torch.autograd.Function):
    """fp32 value whose gradient is stored as bf16"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return q(g)


def _conv_forward(self, x):
    return F.conv2d(_QBoth.apply(x), _QFwd.apply(self.weight), self.bias, self.stride, self.padding, self.dilation,
                    self.groups)


def _bn_forward(self, y32):
    if not self.training:
        # inference plan: BN on the fp32 accumulators inside the conv epilogue (DmlConvDesc.post_*)
        return F.batch_norm(y32, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)
    # training plan: statistics from the fp32 accumulators (conv epilogue), y stored as bf16, normalised from the
    # stored copy (bn_apply_cols_kernel: (y - mean) * gamma * invstd + beta); d(loss)/dy is stored as bf16
    y32 = _QGrad.apply(y32)
    dims = (0, 2, 3)
    n = y32.numel() // y32.shape[1]
    mean = y32.mean(dims)
    var = y32.var(dims, unbiased=False)
    with torch.no_grad():
        m = self.momentum
        self.running_mean.mul_(1 - m).add_(m * mean)
        self.running_var.mul_(1 - m).add_(m * var * (n / max(n - 1, 1)))
        self.num_batches_tracked += 1
    yq = _QFwd.apply(y32)
    inv = torch.rsqrt(var + self.eps)
    sh = (1, -1, 1, 1)
    return (yq - mean.view(sh)) * (self.weight * inv).view(sh) + self.bias.view(sh)


def emulate_bf16_storage(model: nn.Module) -> nn.Module:
    """Patch an oracle model instance (DeepLabV3PlusEmbeddingRef) in place; returns it."""
    for mod in model.modules():
        if isinstance(mod, nn.Conv2d):
            mod.forward = types.MethodType(_conv_forward, mod)
        elif isinstance(mod, nn.BatchNorm2d):
            mod.forward = types.MethodType(_bn_forward, mod)
        elif isinstance(mod, (O._Bottleneck, O._ASPP)):
            # block outputs feed the next block's residual add, the ASPP output feeds the bilinear resize: stored
            mod.register_forward_hook(lambda m, inp, out: _QBoth.apply(out))
        elif isinstance(mod, O._Head):
            # the embedding stays fp32 (y_f32), the gradient that bilinear_bwd hands back is bf16
            mod.register_forward_hook(lambda m, inp, out: _QGrad.apply(out))
    for mod in model.modules():
        if isinstance(mod, O._ASPP) and mod.training:
            # the image-pooling branch (AdaptiveAvgPool2d -> 1x1 conv -> BN over only B samples -> ReLU) is kept in fp32
            # storage by the bf16 TRAINING plans (engine.py, _head_fwd): fp32 pooled input (the mean of the stored bf16
            # `out`), fp32 weights, no rounding of y / dy inside the unit, fp32 output gradient and input gradient; only
            # its broadcast output is rounded, by the consumer's conv input quantisation
            pool = mod.convs[4]
            pool[1].__dict__.pop("forward", None)
            pool[2].__dict__.pop("forward", None)
    return model
