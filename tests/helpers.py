"""Shared test helpers: deterministic synthetic weights / inputs (numpy PCG64, platform-stable).

The same generators are used by ``tests/tools/mint_golden.py`` (authoring container, real reference)
and by the tests (oracle on CPU, HIP path on the GPU box), so a golden vector only needs to
store seeds and expected outputs, never a 235 MB state_dict.
"""
from __future__ import annotations

import os
import sys
import zlib
from collections import OrderedDict

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "open-world-semantic-segmentation_amd")
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def rng_for(seed: int, name: str) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64([int(seed), zlib.crc32(name.encode())]))


def synth_tensor(seed: int, name: str, shape, kind="normal", scale=1.0) -> torch.Tensor:
    g = rng_for(seed, name)
    if kind == "normal":
        a = g.standard_normal(tuple(shape)) * scale
    elif kind == "uniform":
        a = g.uniform(-scale, scale, tuple(shape))
    else:
        raise ValueError(kind)
    return torch.from_numpy(a.astype(np.float32))


def synth_state_dict(shapes: "OrderedDict[str, tuple]", seed: int = 1) -> "OrderedDict[str, torch.Tensor]":
    """Deterministic, well-conditioned weights for any module whose keys follow the reference.

    conv weights ~ N(0, 2/fan) (fan_out under ``backbone.``, fan_in elsewhere, mirroring
    backbone/resnet.py:156 and network/utils.py:37); BN gamma ~ U(0.5,1.5), beta ~ N(0,0.1),
    running_mean ~ N(0,0.1), running_var ~ U(0.5,1.5); conv bias ~ U(-0.05,0.05).
    """
    out = OrderedDict()
    for key, shape in shapes.items():
        shape = tuple(shape)
        g = rng_for(seed, key)
        leaf = key.rsplit(".", 1)[-1]
        if leaf == "num_batches_tracked":
            out[key] = torch.zeros((), dtype=torch.int64)
            continue
        if len(shape) == 4:
            o, i, kh, kw = shape
            fan = (o if key.startswith("backbone.") else i) * kh * kw
            a = g.standard_normal(shape) * np.sqrt(2.0 / fan)
        elif leaf == "running_mean":
            a = g.standard_normal(shape) * 0.1
        elif leaf == "running_var":
            a = g.uniform(0.5, 1.5, shape)
        elif leaf == "weight":
            # BN gamma.  The last BN of every residual branch gets a small gamma (as zero_init_residual /
            # a trained network would): with gamma ~ 1 everywhere a random-init ResNet-101 amplifies fp32
            # rounding ~5000x through depth and the reference's OWN gradients move by 5-20 % between fp32 and
            # fp64 (or 8 vs 1 CPU threads), which would make any 1e-3 parity bar meaningless.
            a = g.uniform(0.05, 0.15, shape) if key.endswith("bn3.weight") else g.uniform(0.5, 1.5, shape)
        elif leaf == "bias":
            # BN beta or the final conv bias (network/utils.py:23)
            a = g.standard_normal(shape) * 0.1 if not key.endswith("classifier.3.bias") \
                else g.uniform(-0.05, 0.05, shape)
        else:
            raise KeyError(key)
        out[key] = torch.from_numpy(np.asarray(a, dtype=np.float32))
    return out


def conditioned_state_dict(shapes: "OrderedDict[str, tuple]", seed: int, beta_idx, beta_val) -> "OrderedDict[str, torch.Tensor]":
    """synth_state_dict with the BatchNorm betas a large fixture moved (tests/tools/mint_golden_large.py: no ReLU input of the
    network within 64 * eps32 * sum|terms| of zero).  `beta_idx` indexes the concatenation of all BatchNorm biases in state_dict
    order, `beta_val` holds the fp32 values that replace them."""
    sd = synth_state_dict(shapes, seed)
    keys = [k for k in sd if k.endswith(".bias") and (k[:-4] + "running_mean") in sd]
    flat = torch.cat([sd[k].flatten() for k in keys])
    flat[torch.as_tensor(np.asarray(beta_idx), dtype=torch.long)] = torch.as_tensor(np.asarray(beta_val), dtype=torch.float32)
    off = 0
    for k in keys:
        n = sd[k].numel()
        sd[k] = flat[off:off + n].clone().view(sd[k].shape)
        off += n
    return sd


def shapes_of(module: torch.nn.Module) -> "OrderedDict[str, tuple]":
    return OrderedDict((k, tuple(v.shape)) for k, v in module.state_dict().items())


def synth_labels(seed: int, name: str, shape, num_classes: int, ignore_index: int,
                 ignore_rows: int = 0, ignore_frac: float = 0.0) -> torch.Tensor:
    g = rng_for(seed, name)
    lab = g.integers(0, num_classes, size=tuple(shape)).astype(np.int64)
    if ignore_frac > 0:
        lab[g.random(tuple(shape)) < ignore_frac] = ignore_index
    if ignore_rows > 0:
        lab[..., :ignore_rows, :] = ignore_index
    return torch.from_numpy(lab)


def checksum(t: torch.Tensor):
    t = t.detach().double().flatten()
    return np.array([t.sum().item(), t.abs().sum().item(), (t * t).sum().item()], dtype=np.float64)


def load_golden(name: str):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return ((a - b).abs().max() / (b.abs().max() + 1e-12)).item()


def max_abs(a, b) -> float:
    return (a.detach().double().cpu() - b.detach().double().cpu()).abs().max().item()


# ---------------------------------------------------------------------------------------------------
# test-only: plans of ONE piece of the network, built from the product's own plan pieces (dmlnet.engine.Plan.block_fwd /
# block_bwd / _head_fwd / _head_bwd), so that the per-block and head-only fixtures (G4, G3: SURVEY 8(c) calls them "the
# binding fixtures") bind the HIP path and not only the oracle
# ---------------------------------------------------------------------------------------------------
def piece_plan(kind: str, module: torch.nn.Module, in_shapes, dtype=torch.float32):
    """kind = "block": module is a network.modeling.Bottleneck, in_shapes = [(B, C, H, W)];
    kind = "head": module is a network.modeling.DeepLabHeadV3Plus, in_shapes = [low (B,256,h,w), out (B,2048,h',w')].
    Returns a PiecePlan with .run(inputs, grad_out) -> (output, input gradients); parameter gradients / running
    statistics land in the module (p.grad, buffers) exactly as in the full model."""
    import torch.nn as nn
    from dmlnet import engine as E

    class _Holder(nn.Module):
        def __init__(self):
            super().__init__()
            self.backbone = nn.ModuleDict({"blk": module} if kind == "block" else {})
            self.classifier = module if kind == "head" else nn.Module()

    class PiecePlan(E.Plan):
        def build(self):
            self.bn_eval, self.pre_prep, self.nbt_inc = [], [], None
            self.to_backbone_ops, self.head_bwd_range = [], {}
            if kind == "block":
                B, Cc, Hh, Ww = in_shapes[0]
                self.inputs = [self.new(B, Hh, Ww, Cc)]
                rec = self.block_fwd(self.inputs[0], module)
                self.output = rec[3].z
                self.grad_out = self.grad_of(self.output)
                self.output.root.grad_init = True
                self.block_bwd(rec)
                self.flush_wgrad()
                self.skip = []
            else:
                (B, Cl, hl, wl), (_, Co, ho, wo) = in_shapes
                low, out = self.new(B, hl, wl, Cl), self.new(B, ho, wo, Co)
                self.inputs = [low, out]
                rec = self._head_fwd(module, low, out)
                self.heads = [rec]
                self.n_fwd = len(self.fwd) - 1              # without the final upsample + distance op (needs per-call outputs)
                self.output = rec.emb
                self._head_bwd(rec, low, out)
                self.flush_wgrad()
                self.grad_out = rec.de
                self.skip = [r for r in (rec.fused_range, rec.unfused_range) if r is not None]

        @staticmethod
        def _fill(act, t):          # NCHW cpu tensor -> the plan's NHWC buffer
            v = t.permute(0, 2, 3, 1).contiguous().to(act.t.device, act.t.dtype)
            act.t.view(act.B, act.H, act.W, act.ld)[..., :act.C].copy_(v)

        @staticmethod
        def _read(act):
            return act.t.view(act.B, act.H, act.W, act.ld)[..., :act.C].permute(0, 3, 1, 2).float().cpu()

        def run(self, inputs, grad_out):
            st = self.e.store
            stream = torch.cuda.current_stream(self.device).cuda_stream
            self.refresh_weights(stream)
            for a, t in zip(self.inputs, inputs):
                self._fill(a, t)
            for args, idx, bn in self.momentum_slots:
                args[idx] = float(bn.momentum)
            st.flat_nbt.add_(1)
            self._exec(self.fwd, stream, 0, getattr(self, "n_fwd", None))
            y = self._read(self.output)
            self._fill(self.grad_out, grad_out)
            st.begin_backward()
            self.skip_ranges = self.skip
            self.run_backward()
            st.end_backward()
            torch.cuda.synchronize()
            return y, [self._read(self.grad_of(a)) for a in self.inputs]

    holder = _Holder().cuda()
    eng = E.Engine(holder)
    eng.store.bind(torch.device("cuda", torch.cuda.current_device()))
    (B, _, Hh, Ww) = in_shapes[0]
    H_full, W_full = (Hh, Ww) if kind == "block" else (4 * Hh, 4 * Ww)
    return PiecePlan(eng, B, H_full, W_full, dtype, True)
