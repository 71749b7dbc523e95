"""Model factories with the reference's names and signatures (network/modeling.py:86-170 there).

Only the embedding DeepLabV3+ on a dilated ResNet is in scope (BASELINE.json north_star); the other
names of the drivers' `model_map` (main_embedding.py:368-374) exist and raise NotImplementedError, the
reference's own behaviour for an unknown backbone (modeling.py:80).

The returned nn.Module is a parameter container with the reference's module tree and 674-key
state_dict; its forward is executed by hand-written gfx950 kernels through `dmlnet.engine`.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from dmlnet.engine import Engine

__all__ = ["deeplabv3_resnet50", "deeplabv3plus_resnet50", "deeplabv3_resnet101", "deeplabv3plus_resnet101",
           "deeplabv3plus_embedding_resnet101", "deeplabv3plus_embedding_resnet50",
           "deeplabv3plus_embedding_self_distillation_resnet101", "deeplabv3_mobilenet",
           "deeplabv3plus_mobilenet", "convert_to_separable_conv"]

_DEPTHS = {"resnet50": (3, 4, 6, 3), "resnet101": (3, 4, 23, 3)}


class Bottleneck(nn.Module):
    """Parameter holder for network/backbone/resnet.py:75-115 (1x1 -> 3x3(stride, dilation) -> 1x1, x4)."""
    expansion = 4

    def __init__(self, inplanes, planes, stride, dilation, downsample):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResNetTrunk(nn.Module):
    """conv1 .. layer4 of a dilated ResNet = what IntermediateLayerGetter keeps (network/utils.py:227-242)."""

    def __init__(self, depths, replace_stride_with_dilation):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        inplanes, dilation = 64, 1
        for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), depths)):
            stride = 1 if li == 0 else 2
            prev = dilation
            if li > 0 and replace_stride_with_dilation[li - 1]:       # resnet.py:174-177
                dilation *= stride
                stride = 1
            blocks = []
            for bi in range(nblk):
                first = bi == 0
                ds = None
                if first and (stride != 1 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False),
                                       nn.BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if first else 1, prev if first else dilation, ds))
                inplanes = planes * 4
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        for m in self.modules():                                       # resnet.py:154-159
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


class ASPP(nn.Module):
    """network/utils.py:332-361."""

    def __init__(self, in_channels, rates):
        super().__init__()
        def branch(k, d):
            return nn.Sequential(nn.Conv2d(in_channels, 256, k, padding=d if k == 3 else 0, dilation=d, bias=False),
                                 nn.BatchNorm2d(256), nn.ReLU(inplace=True))
        mods = [branch(1, 1)] + [branch(3, r) for r in rates]
        mods.append(nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(in_channels, 256, 1, bias=False),
                                  nn.BatchNorm2d(256), nn.ReLU(inplace=True)))
        self.convs = nn.ModuleList(mods)
        self.project = nn.Sequential(nn.Conv2d(5 * 256, 256, 1, bias=False), nn.BatchNorm2d(256),
                                     nn.ReLU(inplace=True), nn.Dropout(0.1))


class DeepLabHeadV3Plus(nn.Module):
    """network/utils.py:8-40."""

    def __init__(self, in_channels, low_level_channels, num_classes, aspp_dilate=(12, 24, 36)):
        super().__init__()
        self.project = nn.Sequential(nn.Conv2d(low_level_channels, 48, 1, bias=False), nn.BatchNorm2d(48),
                                     nn.ReLU(inplace=True))
        self.aspp = ASPP(in_channels, aspp_dilate)
        self.classifier = nn.Sequential(nn.Conv2d(304, 256, 3, padding=1, bias=False), nn.BatchNorm2d(256),
                                        nn.ReLU(inplace=True), nn.Conv2d(256, num_classes, 1))
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight)


class _ModelFn(torch.autograd.Function):
    """Whole-network autograd node: parameter gradients are written by the backward plan straight into the
    flat gradient buffer (p.grad become views of it); nothing is returned through autograd for them."""

    @staticmethod
    def forward(ctx, model, x, anchor):
        plan, logits, feats = model._engine.forward(x, model.compute_dtype, True)
        ctx.model, ctx.plan, ctx.nheads = model, plan, len(logits)
        ctx.set_materialize_grads(False)
        return tuple(logits) + tuple(feats)              # (logits of every head..., features of every head...)

    @staticmethod
    def backward(ctx, *grads):
        n = ctx.nheads
        ctx.model._engine.backward(ctx.plan, list(grads[:n]), list(grads[n:]))
        return None, None, None


class DeepLabV3_embedding(nn.Module):
    """network/_deeplab.py:28-43 + network/utils.py:56-118: forward(x) -> (logits, centers, features_out)."""

    def __init__(self, backbone, classifier):
        super().__init__()
        self.backbone = backbone
        self.classifier = classifier
        env = os.environ.get("DMLNET_DTYPE", "f32").lower()
        self.compute_dtype = torch.bfloat16 if env in ("bf16", "bfloat16") else torch.float32
        object.__setattr__(self, "_engine", Engine(self))
        self._anchor = None
        self._register_state_dict_hook(self._own_storage_hook)

    @staticmethod
    def _own_storage_hook(module, state_dict, prefix, local_metadata):
        """Checkpoints stay in the reference's format (contiguous OIHW fp32 tensors, one storage each,
        main_embedding.py:404-414) although the live parameters are views of one flat K-R-S-C buffer."""
        for k, v in list(state_dict.items()):
            if torch.is_tensor(v) and (not v.is_contiguous() or
                                       v.untyped_storage().nbytes() != v.numel() * v.element_size()):
                state_dict[k] = v.detach().clone(memory_format=torch.contiguous_format)
        return state_dict

    def set_compute_dtype(self, dtype, fp32_products=None):
        """torch.float32: the reference's arithmetic (network/utils.py:84-118 computes in fp32) -- exact fp32 MFMAs by default;
        `fp32_products="bf16x3"` keeps fp32 storage everywhere and computes the convolutions' products on the bf16 matrix
        cores through a three-term split of both operands (fp32-level error, DmlConvDesc.f32_split); `fp32_products="f16x2"`
        computes them on the fp16 matrix cores through a two-term split of the power-of-two-scaled operands (22 significand
        bits, the reference's own fp32-vs-fp64 level on the parity fixtures; three MFMAs per block instead of six; the
        weight gradients and the shapes its kernel does not take stay on the three-term split).  torch.bfloat16: bf16 storage
        of activations / compute weights, fp32 accumulation (the throughput mode)."""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be float32 or bfloat16")
        if fp32_products not in (None, "exact", "bf16x3", "f16x2"):
            raise ValueError("fp32_products must be 'exact', 'bf16x3' or 'f16x2'")
        self.compute_dtype = dtype
        if fp32_products is not None:
            self._engine.f32_split = {"exact": 0, "bf16x3": 1, "f16x2": 2}[fp32_products]
        return self

    def set_sync_batchnorm(self, enabled=True, group=None):
        """Batch statistics over every rank of `group` instead of per replica (the anomaly side of the reference
        trains with SynchronizedBatchNorm2d, anomaly/lib/nn/modules/batchnorm.py:56-139; the DeepLab drivers do not).
        Needs an initialised torch.distributed process group and equal per-rank batches; off by default."""
        self._engine.sync_bn, self._engine.sync_group = bool(enabled), group
        self._engine.plans.clear()
        return self

    @property
    def centers(self):
        return 3.0 * torch.eye(self.classifier.classifier[3].out_channels)

    def forward(self, x):
        eng = self._engine
        k = self.classifier.classifier[3].out_channels
        if self.training and torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            logits, feats = _ModelFn.apply(self, x, self._anchor)
        else:
            _, logits, feats = eng.forward(x, self.compute_dtype, self.training)
            logits, feats = logits[0], feats[0]
        return logits, eng.prototypes(k), feats


class DeepLabV3_embedding_self_distillation(DeepLabV3_embedding):
    """network/_deeplab.py:45 + network/utils.py:120-193 of the reference: one backbone, a base head with 16 prototypes
    and `cls_novel` further heads with 17, 18, ... (the incremental classes); forward(x) returns three LISTS
    (logits, centers, features_out), one entry per head.  The loss of main_self_distillation.py:447-507 reaches the
    last head only -- the backward plan then skips the other heads' segments."""

    base_classes = 16            # utils.py:131 hard-codes it
    cls_novel = 1                # utils.py:125

    def __init__(self, backbone, rates):
        nn.Module.__init__(self)
        self.backbone = backbone
        self.classifier_list = ["classifier"] + ["classifier_%d" % (i + 1) for i in range(self.cls_novel)]
        self.classifier = DeepLabHeadV3Plus(2048, 256, self.base_classes, rates)
        for i in range(self.cls_novel):
            setattr(self, self.classifier_list[i + 1], DeepLabHeadV3Plus(2048, 256, self.base_classes + i + 1, rates))
        env = os.environ.get("DMLNET_DTYPE", "f32").lower()
        self.compute_dtype = torch.bfloat16 if env in ("bf16", "bfloat16") else torch.float32
        object.__setattr__(self, "_engine", Engine(self))
        self._anchor = None
        self._register_state_dict_hook(self._own_storage_hook)

    def head_modules(self):
        return [getattr(self, n) for n in self.classifier_list]

    def forward(self, x):
        eng = self._engine
        n = len(self.classifier_list)
        if self.training and torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != x.device:
                self._anchor = torch.zeros(1, device=x.device, requires_grad=True)
            outs = _ModelFn.apply(self, x, self._anchor)
            logits, feats = list(outs[:n]), list(outs[n:])
        else:
            _, logits, feats = eng.forward(x, self.compute_dtype, self.training)
        centers = [eng.prototypes(h.classifier[3].out_channels) for h in self.head_modules()]
        return logits, centers, feats


def _segm_resnet(name, backbone_name, num_classes, output_stride, pretrained_backbone):
    if pretrained_backbone:
        # resnet.py:216 downloads ImageNet weights; load them yourself with load_state_dict instead
        import warnings
        warnings.warn("pretrained_backbone=True ignored: no download here; load a checkpoint with load_state_dict")
    if output_stride == 8:
        dilate, rates = (False, True, True), (12, 24, 36)
    else:
        dilate, rates = (False, False, True), (6, 12, 18)
    if backbone_name not in _DEPTHS:
        raise NotImplementedError(backbone_name)
    backbone = ResNetTrunk(_DEPTHS[backbone_name], dilate)
    if name == "deeplabv3plus_embedding_self_distillation":
        return DeepLabV3_embedding_self_distillation(backbone, rates)        # modeling.py:39-40: num_classes unused
    if name != "deeplabv3plus_embedding":
        raise NotImplementedError("%s: only the embedding DeepLabV3+ models are built on MI355X (BASELINE.json)" % name)
    return DeepLabV3_embedding(backbone, DeepLabHeadV3Plus(2048, 256, num_classes, rates))


def _load_model(arch_type, backbone, num_classes, output_stride, pretrained_backbone):
    if backbone.startswith("resnet"):
        return _segm_resnet(arch_type, backbone, num_classes, output_stride, pretrained_backbone)
    raise NotImplementedError(backbone)


def deeplabv3plus_embedding_resnet101(num_classes=21, output_stride=8, pretrained_backbone=True):
    """DeepLabV3+ with a ResNet-101 backbone and the prototype-distance head (modeling.py:140-148)."""
    return _load_model("deeplabv3plus_embedding", "resnet101", num_classes, output_stride, pretrained_backbone)


def deeplabv3plus_embedding_resnet50(num_classes=21, output_stride=8, pretrained_backbone=True):
    return _load_model("deeplabv3plus_embedding", "resnet50", num_classes, output_stride, pretrained_backbone)


def _out_of_scope(name):
    def factory(num_classes=21, output_stride=8, pretrained_backbone=True):
        raise NotImplementedError("%s is outside the MI355X hot path (SURVEY.md section 8); use "
                                  "deeplabv3plus_embedding_resnet101" % name)
    factory.__name__ = name
    return factory


deeplabv3_resnet50 = _out_of_scope("deeplabv3_resnet50")
deeplabv3plus_resnet50 = _out_of_scope("deeplabv3plus_resnet50")
deeplabv3_resnet101 = _out_of_scope("deeplabv3_resnet101")
deeplabv3plus_resnet101 = _out_of_scope("deeplabv3plus_resnet101")


def deeplabv3plus_embedding_self_distillation_resnet101(num_classes=21, output_stride=8, pretrained_backbone=True):
    """Shared backbone + base head (16) + incremental head(s) (17, ...), modeling.py:150-158 of the reference
    (num_classes is accepted and ignored there as well: the head widths are fixed in the model class)."""
    return _load_model("deeplabv3plus_embedding_self_distillation", "resnet101", num_classes, output_stride,
                       pretrained_backbone)

deeplabv3_mobilenet = _out_of_scope("deeplabv3_mobilenet")
deeplabv3plus_mobilenet = _out_of_scope("deeplabv3plus_mobilenet")


def convert_to_separable_conv(module):
    raise NotImplementedError("separable convolutions (network/utils.py:364-376) are outside the hot path")
