"""CPU restatement of the anomaly sub-project's open-set segmentation model -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(models/ -> dmlnet.engine_ppm -> libdmlnet_hip.so) never does.

What it restates (inference branch only):
  anomaly/models/resnet.py:60-93,96-166        deep-stem ResNet-50 (three 3x3 stem convs, Bottleneck with the stride on conv2)
  anomaly/models/models.py:285-346             ResnetDilated(dilate_scale=8): layer3 / layer4 strides -> dilations 2 / 4
  anomaly/models/models.py:586-668             PPMDeepsup_embedding: pool scales (1,2,3,6), 3x3 fusion conv, 13-channel
                                               embedding, -||f - 3 e_k||^2 at 1/8 resolution, THEN bilinear to segSize
  anomaly/eval_ood_traditional.py:190-210      multi-scale mean of the upsampled scores / features
  anomaly/lib/nn/modules/batchnorm.py:56-61    SynchronizedBatchNorm2d in eval() = F.batch_norm with running statistics
Parity pin: tests/golden/g14_ppm.npz, minted from the reference's own classes by tests/tools/mint_golden_ppm.py (weights are
regenerated on both sides from tests/helpers.synth_state_dict; the fixture holds inputs and outputs only).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class _BN(nn.BatchNorm2d):
    def __init__(self, c):
        super().__init__(c, eps=1e-5, momentum=0.001)
        for name, t in (("_tmp_running_mean", torch.zeros(c)), ("_tmp_running_var", torch.ones(c)), ("_running_iter", torch.ones(1))):
            self.register_buffer(name, t)


def _cb(cin, cout, k, stride=1, dil=1):
    return [nn.Conv2d(cin, cout, k, stride=stride, padding=dil * (k // 2), dilation=dil, bias=False), _BN(cout)]


class _Block(nn.Module):
    def __init__(self, cin, planes, stride, dil, down):
        super().__init__()
        self.conv1, self.bn1 = _cb(cin, planes, 1)
        self.conv2, self.bn2 = _cb(planes, planes, 3, stride, dil)
        self.conv3, self.bn3 = _cb(planes, planes * 4, 1)
        self.downsample = nn.Sequential(*_cb(cin, planes * 4, 1, stride)) if down else None

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        return F.relu(y + (x if self.downsample is None else self.downsample(x)))


class EncoderRef(nn.Module):
    def __init__(self, depths=(3, 4, 6, 3)):
        super().__init__()
        self.conv1, self.bn1 = _cb(3, 64, 3, 2)
        self.conv2, self.bn2 = _cb(64, 64, 3)
        self.conv3, self.bn3 = _cb(64, 128, 3)
        cin = 128
        # (stride of the first block, dilation of the first block, dilation of the others) after _nostride_dilate
        cfg = [(1, 1, 1), (2, 1, 1), (1, 1, 2), (1, 2, 4)]
        for li, (planes, n) in enumerate(zip((64, 128, 256, 512), depths)):
            s, d0, d = cfg[li]
            blocks = [_Block(cin, planes, s, d0, True)] + [_Block(planes * 4, planes, 1, d, False) for _ in range(n - 1)]
            cin = planes * 4
            setattr(self, "layer%d" % (li + 1), nn.Sequential(*blocks))

    def forward(self, x):
        x = F.relu(self.bn1(self.conv1(x)))
        x = F.relu(self.bn2(self.conv2(x)))
        x = F.relu(self.bn3(self.conv3(x)))
        x = F.max_pool2d(x, 3, 2, 1)
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))


class DecoderRef(nn.Module):
    def __init__(self, num_class=13, fc_dim=2048, scales=(1, 2, 3, 6)):
        super().__init__()
        self.scales = scales
        self.ppm = nn.ModuleList([nn.Sequential(nn.Identity(), *_cb(fc_dim, 512, 1)) for _ in scales])
        self.cbr_deepsup = nn.Sequential(*_cb(fc_dim // 2, fc_dim // 4, 3))
        self.conv_last = nn.Sequential(*_cb(fc_dim + 512 * len(scales), 512, 3), nn.Identity(), nn.Identity(),
                                       nn.Conv2d(512, num_class, 1))
        self.conv_last_deepsup = nn.Conv2d(fc_dim // 4, num_class, 1)

    def forward(self, conv5, seg_size):
        h, w = conv5.shape[2:]
        parts = [conv5]
        for s, br in zip(self.scales, self.ppm):
            p = F.relu(br[2](br[1](F.adaptive_avg_pool2d(conv5, s))))
            parts.append(F.interpolate(p, (h, w), mode="bilinear", align_corners=False))
        x = torch.cat(parts, 1)
        x = F.relu(self.conv_last[1](self.conv_last[0](x)))
        emb = self.conv_last[4](x)                                            # [B, 13, h, w]
        centers = 3.0 * torch.eye(13, dtype=emb.dtype)
        d = -((emb.permute(0, 2, 3, 1).unsqueeze(3) - centers) ** 2).sum(-1).permute(0, 3, 1, 2)
        return (F.interpolate(d, size=seg_size, mode="bilinear", align_corners=False),
                F.interpolate(emb, size=seg_size, mode="bilinear", align_corners=False))


class SegmentationModuleOODRef(nn.Module):
    def __init__(self, depths=(3, 4, 6, 3), num_class=13):
        super().__init__()
        self.encoder = EncoderRef(depths)
        self.decoder = DecoderRef(num_class)

    def forward(self, img, seg_size):
        return self.decoder(self.encoder(img), seg_size)


def evaluate_multiscale(model, imgs, seg_size):
    n = len(imgs)
    scores = ft = 0
    for img in imgs:
        s, f = model(img, seg_size)
        scores = scores + s / n
        ft = ft + f / n
    return scores, ft
