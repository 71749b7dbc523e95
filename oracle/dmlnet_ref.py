"""CPU ORACLE (test infrastructure, NOT product code).

Plain-PyTorch fp32 restatement of the DMLNet hot path of
Jun-CEN/Open-World-Semantic-Segmentation.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this file; the product package (``open-world-semantic-segmentation_amd/``)
never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tests/tools/mint_golden.py`` imports the real reference
from ``/root/reference`` (authoring container only), checks this restatement
against it (same state_dict keys, same outputs/gradients on seeded inputs) and
writes the golden vectors in ``tests/golden/`` that ``tests/test_oracle.py``
re-checks everywhere.  The reference has no tests of its own for this path
(SURVEY.md §4), so those fixtures are the pin.

Each function cites the reference lines (relative to ``/root/reference``) it
follows.  The arithmetic that lives in ATen (conv / batch-norm / bilinear /
cross-entropy; reference pin torch==1.5.0, ``requirements.txt:131``) is taken
from ``torch.nn.functional`` on CPU.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

# --------------------------------------------------------------------------
# Architecture description (DeepLabV3Plus-Pytorch/network/backbone/resnet.py:118-193,
# network/modeling.py:6-43, network/utils.py:8-32,308-361)
# --------------------------------------------------------------------------

RESNET101_BLOCKS = (3, 4, 23, 3)          # backbone/resnet.py:266
STAGE_PLANES = (64, 128, 256, 512)
EXPANSION = 4                              # backbone/resnet.py:76


def stage_plan(output_stride: int) -> List[dict]:
    """Per-bottleneck (stride, dilation, has_downsample) for ResNet-101.

    Follows ``ResNet._make_layer`` (backbone/resnet.py:171-193): the first block
    of a dilated stage keeps the *previous* dilation, later blocks use the new one;
    OS16 dilates layer4 only, OS8 dilates layer3 and layer4 (modeling.py:8-13).
    """
    if output_stride == 8:
        dilate = (False, True, True)
    else:
        dilate = (False, False, True)
    plan, inplanes, dilation = [], 64, 1
    for si, (planes, nblocks) in enumerate(zip(STAGE_PLANES, RESNET101_BLOCKS)):
        stride = 1 if si == 0 else 2
        prev_dil = dilation
        if si > 0 and dilate[si - 1]:
            dilation *= stride
            stride = 1
        for bi in range(nblocks):
            first = bi == 0
            plan.append(dict(
                stage=si + 1, index=bi, inplanes=inplanes, planes=planes,
                stride=stride if first else 1,
                dilation=prev_dil if first else dilation,
                downsample=first and (stride != 1 or inplanes != planes * EXPANSION),
            ))
            inplanes = planes * EXPANSION
    return plan


def aspp_rates(output_stride: int) -> Tuple[int, int, int]:
    return (12, 24, 36) if output_stride == 8 else (6, 12, 18)   # modeling.py:8-13


def _conv(cin, cout, k, stride=1, dilation=1, bias=False):
    pad = dilation * (k // 2) if k == 3 else k // 2
    return nn.Conv2d(cin, cout, k, stride=stride, padding=pad, dilation=dilation, bias=bias)


class _Bottleneck(nn.Module):
    """backbone/resnet.py:75-115."""

    def __init__(self, cfg):
        super().__init__()
        w, out = cfg["planes"], cfg["planes"] * EXPANSION
        self.conv1 = _conv(cfg["inplanes"], w, 1)
        self.bn1 = nn.BatchNorm2d(w)
        self.conv2 = _conv(w, w, 3, cfg["stride"], cfg["dilation"])
        self.bn2 = nn.BatchNorm2d(w)
        self.conv3 = _conv(w, out, 1)
        self.bn3 = nn.BatchNorm2d(out)
        if cfg["downsample"]:
            self.downsample = nn.Sequential(_conv(cfg["inplanes"], out, 1, cfg["stride"]),
                                            nn.BatchNorm2d(out))
        else:
            self.downsample = None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        idt = x if self.downsample is None else self.downsample(x)
        return F.relu(y + idt)


class _Backbone(nn.Module):
    """Truncated ResNet-101 = what IntermediateLayerGetter keeps (network/utils.py:227-251)."""

    def __init__(self, output_stride):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)   # resnet.py:139
        self.bn1 = nn.BatchNorm2d(64)
        plan = stage_plan(output_stride)
        for s in (1, 2, 3, 4):
            setattr(self, "layer%d" % s,
                    nn.Sequential(*[_Bottleneck(c) for c in plan if c["stage"] == s]))
        for m in self.modules():                                           # resnet.py:154-159
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.max_pool2d(x, 3, 2, 1)                                       # resnet.py:143
        low = self.layer1(x)
        out = self.layer4(self.layer3(self.layer2(low)))
        return OrderedDict(low_level=low, out=out)


class _ASPP(nn.Module):
    """network/utils.py:308-361."""

    def __init__(self, cin, rates):
        super().__init__()
        branches = [nn.Sequential(_conv(cin, 256, 1), nn.BatchNorm2d(256), nn.ReLU())]
        for r in rates:
            branches.append(nn.Sequential(_conv(cin, 256, 3, 1, r), nn.BatchNorm2d(256), nn.ReLU()))
        branches.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), _conv(cin, 256, 1),
                                      nn.BatchNorm2d(256), nn.ReLU()))
        self.convs = nn.ModuleList(branches)
        self.project = nn.Sequential(_conv(5 * 256, 256, 1), nn.BatchNorm2d(256), nn.ReLU(),
                                     nn.Dropout(0.1))

    def forward(self, x):
        hw = x.shape[-2:]
        outs = [b(x) for b in self.convs[:4]]
        outs.append(F.interpolate(self.convs[4](x), size=hw, mode="bilinear", align_corners=False))
        return self.project(torch.cat(outs, 1))


class _Head(nn.Module):
    """DeepLabHeadV3Plus, network/utils.py:8-40."""

    def __init__(self, num_classes, rates):
        super().__init__()
        self.project = nn.Sequential(_conv(256, 48, 1), nn.BatchNorm2d(48), nn.ReLU())
        self.aspp = _ASPP(2048, rates)
        self.classifier = nn.Sequential(_conv(304, 256, 3), nn.BatchNorm2d(256), nn.ReLU(),
                                        _conv(256, num_classes, 1, bias=True))
        for m in self.modules():                                           # utils.py:34-40
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)

    def forward(self, feats):
        low = self.project(feats["low_level"])
        hi = F.interpolate(self.aspp(feats["out"]), size=low.shape[2:], mode="bilinear",
                           align_corners=False)
        return self.classifier(torch.cat([low, hi], 1))


def prototypes_3I(k: int, device=None) -> torch.Tensor:
    """centers = 3*I_K rebuilt every forward (network/utils.py:103-106)."""
    return 3.0 * torch.eye(k, dtype=torch.float32, device=device)


def distance_head(x_nchw: torch.Tensor, centers: Optional[torch.Tensor] = None):
    """network/utils.py:92-118: logits[b,k,h,w] = -sum_c (x[b,c,h,w] - M[k,c])^2.

    Returns (logits NCHW, centers, features_out NHWC).  Does not materialise the
    B x HW x K x C tensor the reference builds (same arithmetic, expanded per k).
    """
    feats = x_nchw.permute(0, 2, 3, 1).contiguous()
    if centers is None:
        centers = prototypes_3I(x_nchw.shape[1], x_nchw.device)
    d = feats.unsqueeze(3) - centers.to(feats.dtype)        # B,H,W,K,C
    logits = -(d ** 2).sum(-1).permute(0, 3, 1, 2).contiguous()
    return logits, centers, feats


class DeepLabV3PlusEmbeddingRef(nn.Module):
    """deeplabv3plus_embedding_resnet101 (modeling.py:140-148; utils.py:56-118)."""

    def __init__(self, num_classes=21, output_stride=8):
        super().__init__()
        self.backbone = _Backbone(output_stride)
        self.classifier = _Head(num_classes, aspp_rates(output_stride))

    def embed(self, x):
        e = self.classifier(self.backbone(x))
        return F.interpolate(e, size=x.shape[-2:], mode="bilinear", align_corners=False)  # utils.py:88

    def forward(self, x):
        return distance_head(self.embed(x))


def deeplabv3plus_embedding_resnet101(num_classes=21, output_stride=8, pretrained_backbone=False):
    if pretrained_backbone:
        raise RuntimeError("oracle: no network here, pass pretrained_backbone=False")
    return DeepLabV3PlusEmbeddingRef(num_classes, output_stride)


class DeepLabV3PlusEmbeddingSelfDistillationRef(nn.Module):
    """deeplabv3plus_embedding_self_distillation_resnet101 (modeling.py:150-158; utils.py:120-193): one backbone, a
    16-prototype base head `classifier` and cls_novel = 1 incremental head `classifier_1` with 17; forward returns
    lists (logits, centers, features_out), one entry per head.  Pinned by tests/golden/g12_multihead.npz
    (tests/tools/mint_golden_multihead.py)."""

    def __init__(self, output_stride=8, cls_novel=1, base_classes=16):
        super().__init__()
        self.backbone = _Backbone(output_stride)
        self.classifier_list = ["classifier"] + ["classifier_%d" % (i + 1) for i in range(cls_novel)]
        self.classifier = _Head(base_classes, aspp_rates(output_stride))
        for i in range(cls_novel):
            setattr(self, self.classifier_list[i + 1], _Head(base_classes + i + 1, aspp_rates(output_stride)))

    def forward(self, x):
        feats = self.backbone(x)
        logits, centers, features = [], [], []
        for name in self.classifier_list:
            e = F.interpolate(getattr(self, name)(feats), size=x.shape[-2:], mode="bilinear", align_corners=False)
            lg, ctr, ft = distance_head(e)
            logits.append(lg)
            centers.append(ctr)
            features.append(ft)
        return logits, centers, features


def deeplabv3plus_embedding_self_distillation_resnet101(num_classes=21, output_stride=8, pretrained_backbone=False):
    if pretrained_backbone:
        raise RuntimeError("oracle: no network here, pass pretrained_backbone=False")
    return DeepLabV3PlusEmbeddingSelfDistillationRef(output_stride)


def set_bn_momentum(model: nn.Module, momentum=0.1):
    """utils/utils.py:26-29."""
    for m in model.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = momentum


# --------------------------------------------------------------------------
# Losses
# --------------------------------------------------------------------------

def ce_over_n(logit: torch.Tensor, target: torch.Tensor, ignore_index=255) -> torch.Tensor:
    """Live part of utils/loss.py:34-42: mean CE over valid pixels of the batch, then / n."""
    n = logit.shape[0]
    return F.cross_entropy(logit, target.long(), ignore_index=ignore_index, reduction="mean") / n


def dml_loss(logit: torch.Tensor, target: torch.Tensor, alpha=0.01, ignore_index=-1):
    """Live DML loss of anomaly/models/models.py:42-78 in closed form.

    loss = CE/n + alpha * VAR/n,  VAR = sum_i (1/HW_i) sum_{p: y_p != ignore} (-logit[i, y_p, p]);
    HW_i counts every pixel of image i including ignored ones (models.py:56-58).
    """
    n, k, h, w = logit.shape
    ce = F.cross_entropy(logit, target.long(), ignore_index=ignore_index, reduction="mean")
    valid = target != ignore_index
    tgt = target.long().clamp(0, k - 1)
    own = logit.gather(1, tgt.unsqueeze(1)).squeeze(1)                     # B,H,W
    var = (-(own * valid).flatten(1).sum(1) / float(h * w)).sum()
    return ce / n + alpha * var / n


def dml_loss_loop(logit, target, alpha=0.01, ignore_index=-1):
    """Same loss written as the reference's per-image / per-class loop (models.py:48-78)."""
    n, k, h, w = logit.shape
    ce = F.cross_entropy(logit, target.long(), ignore_index=ignore_index, reduction="mean")
    var = logit.new_zeros(())
    for i in range(n):
        lab = target[i].flatten()
        vec = logit[i].permute(1, 2, 0).reshape(h * w, k)
        total = lab.numel()
        for c in torch.unique(lab).tolist():
            if c == ignore_index:
                continue
            idx = torch.nonzero(lab == c).flatten()
            var = var + (-vec[idx, int(c)]).sum() / total
    return ce / n + alpha * var / n


def pixel_acc(pred: torch.Tensor, label: torch.Tensor) -> torch.Tensor:
    """anomaly/models/models.py:13-21 (valid = label >= 0)."""
    p = pred.argmax(1)
    valid = (label >= 0).long()
    return ((p == label).long() * valid).sum().float() / (valid.sum().float() + 1e-10)


# --------------------------------------------------------------------------
# Open-world scoring (harness arithmetic, numpy in the reference)
# --------------------------------------------------------------------------

def dissum_score(logits_khw: np.ndarray, clip: float, inclusive: bool) -> np.ndarray:
    """s = -sum_k logit_k, clipped, min-max normalised per image.

    anomaly/eval_ood_traditional.py:301-305 (``>= 400 -> 400``; inclusive=True) and
    DeepLabV3Plus-Pytorch/test_embedding.py:349-350,365 (``> 1000 -> 1000``; inclusive=False).
    """
    s = -np.sum(logits_khw, axis=0)
    if inclusive:
        s[s >= clip] = clip
    else:
        s[s > clip] = clip
    return (s - np.min(s)) / (np.max(s) - np.min(s))


def msp_score(logits_bkhw: torch.Tensor) -> torch.Tensor:
    """1 - max softmax (test_embedding.py:340-341)."""
    return 1 - F.softmax(logits_bkhw, dim=1).max(dim=1)[0]


def novel_relabel(preds_hw: np.ndarray, logits_khw: np.ndarray, feats_hwc: np.ndarray,
                  proto: np.ndarray, thresh=-1.5, new_label=16) -> np.ndarray:
    """test_embedding.py:428-445: pixel -> new_label iff -|f-p|^2 > thresh and > max_k logit_k."""
    h, w, c = feats_hwc.shape
    d = -np.sum((feats_hwc.reshape(h * w, c) - proto) ** 2, axis=1).reshape(h, w)
    out = preds_hw.copy()
    out[np.logical_and(d > thresh, d > logits_khw.max(axis=0))] = new_label
    return out


def mean_prototype(shots: Sequence[Sequence[float]]) -> np.ndarray:
    """test_embedding.py:254-257 (float64 mean of the k-shot vectors)."""
    p = np.zeros((len(shots[0]),))
    for v in shots:
        p += np.array(v)
    return p / len(shots)


# --------------------------------------------------------------------------
# Train step (main_embedding.py:385-392,458-507; utils/scheduler.py:3-11)
# --------------------------------------------------------------------------

def poly_lr(base_lr: float, it: int, max_iters: int, power=0.9, min_lr=1e-6) -> float:
    return max(base_lr * (1 - it / max_iters) ** power, min_lr)


def make_optimizer(model, lr=0.01, weight_decay=1e-4):
    return torch.optim.SGD(
        [{"params": model.backbone.parameters(), "lr": 0.1 * lr},
         {"params": model.classifier.parameters(), "lr": lr}],
        lr=lr, momentum=0.9, weight_decay=weight_decay)


def train_step(model, opt, images, labels, it, max_iters, base_lrs, loss_fn):
    """One iteration in the reference's order: zero_grad, fwd, loss, bwd, step, then PolyLR."""
    opt.zero_grad()
    logits, _, feats = model(images)
    loss = loss_fn(logits, labels)
    loss.backward()
    opt.step()
    for g, b in zip(opt.param_groups, base_lrs):
        g["lr"] = poly_lr(b, it + 1, max_iters)
    return loss.detach()


def bilinear(x: torch.Tensor, size) -> torch.Tensor:
    return F.interpolate(x, size=size, mode="bilinear", align_corners=False)
