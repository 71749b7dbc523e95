"""`ModelBuilder` / `SegmentationModuleOOD` with the reference's names and signatures (anomaly/models/models.py:90-120,
122-234) for the one configuration the open-set evaluation uses (anomaly/config/*: encoder `resnet50dilated`, decoder
`ppm_deepsup_embedding`, 13 StreetHazards classes) -- SURVEY.md 8(f) rank 2.

The modules below are parameter containers with the reference's module tree, so its checkpoints
(`encoder_epoch_N.pth` / `decoder_epoch_N.pth`) load with the reference's own `load_state_dict(..., strict=False)`; the
forward runs as a static plan of gfx950 kernels (`dmlnet.engine_ppm`).  Inference only (`segSize` given): the training
branch of the reference's decoder is dead code and raises NotImplementedError here.
"""
from __future__ import annotations

import math
import os

import torch
import torch.nn as nn

from dmlnet.engine_ppm import PPMEngine


class SynchronizedBatchNorm2d(nn.BatchNorm2d):
    """Key-compatible with anomaly/lib/nn/modules/batchnorm.py:39-54 (eps 1e-5, momentum 0.001, three extra buffers)."""

    def __init__(self, num_features, eps=1e-5, momentum=0.001, affine=True):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine)
        self.register_buffer("_tmp_running_mean", torch.zeros(num_features))
        self.register_buffer("_tmp_running_var", torch.ones(num_features))
        self.register_buffer("_running_iter", torch.ones(1))


BatchNorm2d = SynchronizedBatchNorm2d


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, dilation=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=dilation, dilation=dilation, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride


class ResnetDilated(nn.Module):
    """Deep-stem ResNet (resnet.py:96-150) with the strides of layer3 / layer4 turned into dilations
    (models.py:285-328, dilate_scale 8: the block that had stride 2 keeps dilation d/2, the rest of the layer uses d)."""

    def __init__(self, depths=(3, 4, 6, 3), dilate_scale=8):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 3, stride=2, padding=1, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu1 = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv2d(64, 64, 3, padding=1, bias=False)
        self.bn2 = BatchNorm2d(64)
        self.relu2 = nn.ReLU(inplace=True)
        self.conv3 = nn.Conv2d(64, 128, 3, padding=1, bias=False)
        self.bn3 = BatchNorm2d(128)
        self.relu3 = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        dil = {8: (1, 1, 2, 4), 16: (1, 1, 1, 2)}[dilate_scale]
        inplanes = 128
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), depths)):
            stride = 1 if li == 0 else 2
            d = dil[li]
            first_d = 1
            if d > 1:                       # _nostride_dilate: stride-2 convs -> stride 1 (3x3: dilation d // 2)
                stride, first_d = 1, d // 2
            blocks = []
            for bi in range(n):
                ds = None
                if bi == 0 and (li > 0 or inplanes != planes * 4):
                    ds = nn.Sequential(nn.Conv2d(inplanes, planes * 4, 1, stride=stride, bias=False), BatchNorm2d(planes * 4))
                blocks.append(Bottleneck(inplanes, planes, stride if bi == 0 else 1, (first_d if bi == 0 else d), ds))
                inplanes = planes * 4
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))
        for m in self.modules():            # resnet.py:121-127
            if isinstance(m, nn.Conv2d):
                n = m.kernel_size[0] * m.kernel_size[1] * m.out_channels
                m.weight.data.normal_(0, math.sqrt(2.0 / n))
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.fill_(1)
                m.bias.data.zero_()


class PPMDeepsup_embedding(nn.Module):
    """models.py:586-618: four pooling branches, the 3x3 fusion conv, the 13-channel embedding conv; the deep-supervision
    head exists for checkpoint compatibility only."""

    def __init__(self, num_class=150, fc_dim=4096, use_softmax=False, pool_scales=(1, 2, 3, 6)):
        super().__init__()
        self.use_softmax = use_softmax
        self.ppm = nn.ModuleList([nn.Sequential(nn.AdaptiveAvgPool2d(s), nn.Conv2d(fc_dim, 512, 1, bias=False),
                                                BatchNorm2d(512), nn.ReLU(inplace=True)) for s in pool_scales])
        self.cbr_deepsup = nn.Sequential(nn.Conv2d(fc_dim // 2, fc_dim // 4, 3, padding=1, bias=False),
                                         BatchNorm2d(fc_dim // 4), nn.ReLU(inplace=True))
        self.conv_last = nn.Sequential(nn.Conv2d(fc_dim + len(pool_scales) * 512, 512, 3, padding=1, bias=False),
                                       BatchNorm2d(512), nn.ReLU(inplace=True), nn.Dropout2d(0.1),
                                       nn.Conv2d(512, num_class, 1))
        self.conv_last_deepsup = nn.Conv2d(fc_dim // 4, num_class, 1, 1, 0)
        self.dropout_deepsup = nn.Dropout2d(0.1)
        self.centers = 3.0 * torch.eye(13)          # models.py:613-617


class ModelBuilder:
    @staticmethod
    def weights_init(m):                    # models.py:124-133
        classname = m.__class__.__name__
        if classname.find("Conv") != -1:
            nn.init.kaiming_normal_(m.weight.data)
        elif classname.find("BatchNorm") != -1:
            m.weight.data.fill_(1.0)
            m.bias.data.fill_(1e-4)

    @staticmethod
    def build_encoder(arch="resnet50dilated", fc_dim=512, weights=""):
        arch = arch.lower()
        depths = {"resnet50dilated": (3, 4, 6, 3), "resnet101dilated": (3, 4, 23, 3)}.get(arch)
        if depths is None:
            raise NotImplementedError("encoder %r is outside the MI355X build (resnet50dilated / resnet101dilated only)" % arch)
        net = ResnetDilated(depths, dilate_scale=8)         # no ImageNet download here: random init unless `weights`
        if len(weights) > 0:                                # models.py:179-182
            net.load_state_dict(torch.load(weights, map_location=lambda storage, loc: storage), strict=False)
        return net

    @staticmethod
    def build_decoder(arch="ppm_deepsup", fc_dim=512, num_class=150, weights="", use_softmax=False):
        arch = arch.lower()
        if arch != "ppm_deepsup_embedding":
            raise NotImplementedError("decoder %r is outside the MI355X build (ppm_deepsup_embedding only)" % arch)
        net = PPMDeepsup_embedding(num_class=num_class, fc_dim=fc_dim, use_softmax=use_softmax)
        net.apply(ModelBuilder.weights_init)                # models.py:228-232
        if len(weights) > 0:
            net.load_state_dict(torch.load(weights, map_location=lambda storage, loc: storage), strict=False)
        return net


class _EnginePair:
    """what dmlnet.engine expects of a model: `.backbone`, `.head_modules()`"""

    def __init__(self, enc, dec):
        self.backbone, self.decoder = enc, dec

    def head_modules(self):
        return [self.decoder]


class SegmentationModuleOOD(nn.Module):
    """models.py:90-120.  forward(feed_dict, segSize=(H, W)) -> (pred [B,13,H,W], ft [B,13,H,W])."""

    def __init__(self, net_enc, net_dec, crit=None, deep_sup_scale=None):
        super().__init__()
        self.encoder = net_enc
        self.decoder = net_dec
        self.crit = crit
        self.deep_sup_scale = deep_sup_scale
        object.__setattr__(self, "_engine", PPMEngine(_EnginePair(net_enc, net_dec)))
        self._dtype = torch.bfloat16 if os.environ.get("DMLNET_DTYPE", "").lower() in ("bf16", "bfloat16") else torch.float32

    def set_compute_dtype(self, dtype, fp32_products=None):
        """as network.modeling's: torch.float32 (fp32_products "exact" | "bf16x3" | "f16x2") or torch.bfloat16"""
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype is float32 or bfloat16")
        if fp32_products not in (None, "exact", "bf16x3", "f16x2"):
            raise ValueError("fp32_products must be 'exact', 'bf16x3' or 'f16x2'")
        self._dtype = dtype
        if fp32_products is not None:
            self._engine.f32_split = {"exact": 0, "bf16x3": 1, "f16x2": 2}[fp32_products]
        return self

    def _accumulate(self, img, segSize, scores, feats, alpha):
        if self.training or not self.decoder.use_softmax:
            raise NotImplementedError("only the inference branch (eval(), decoder built with use_softmax=True, segSize given) "
                                      "runs on the MI355X path")
        with torch.no_grad():
            return self._engine.infer(img, segSize, self._dtype, scores, feats, alpha)

    def forward(self, feed_dict, *, segSize=None, compute_loss=True):
        if segSize is None:
            raise NotImplementedError("training branch of the pyramid-pooling embedding decoder (dead in the reference, "
                                      "SURVEY.md F11)")
        return self._accumulate(feed_dict["img_data"], segSize, None, None, 1.0)


def evaluate_multiscale(segmentation_module, img_resized_list, segSize):
    """The multi-scale mean of eval_ood_traditional.py:190-210: scores = sum_i pred_i / n, ft1 = sum_i ft_i / n over the
    resized copies of one image; each scale's upsample writes its share straight into the two accumulators, and the
    scales' forward passes overlap on separate streams."""
    m = segmentation_module
    if m.training or not m.decoder.use_softmax:
        raise NotImplementedError("only the inference branch (eval(), decoder built with use_softmax=True) runs on the MI355X path")
    with torch.no_grad():
        return m._engine.infer_multiscale(list(img_resized_list), segSize, m._dtype)
