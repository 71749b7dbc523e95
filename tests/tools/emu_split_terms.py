"""Authoring-container experiment (CPU, oracle only): would a TWO-term bf16 split of the convolution operands
(x = hi + mid, three MFMAs instead of the six of the three-term split) pass the unchanged G5 fixture bars?

Every conv of the pinned oracle is replaced by an autograd function whose three GEMMs (forward, data gradient, weight
gradient) see operands truncated to `terms` bf16 terms; everything else stays fp32.  The result is compared with the
reference-minted g5_full_train fixture exactly as tests/test_gpu_model.py::test_g5_full_train_step_matches_reference does.

    python tests/tools/emu_split_terms.py 2      (or 3, or h2: two fp16 terms of the power-of-two-scaled tensor)
    python tests/tools/emu_split_terms.py h2 g5l (the conditioned 128 x 128 fixture of tests/tools/mint_golden_large.py)
"""
import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import helpers as H  # noqa: E402
from oracle import dmlnet_ref as R  # noqa: E402

MODE = sys.argv[1] if len(sys.argv) > 1 else "2"
TERMS = int(MODE) if MODE.isdigit() else 2


def trunc(v, terms=TERMS):
    if MODE == "h2":
        # two fp16 terms of the tensor scaled by a power of two that puts its largest magnitude below 2^15
        amax = v.abs().max().item()
        s = 2.0 ** (14 - int(np.ceil(np.log2(amax)))) if amax > 0 else 1.0
        xs = v * s
        hi = xs.to(torch.float16).to(torch.float32)
        lo = (xs - hi).to(torch.float16).to(torch.float32)
        return (hi + lo) / s
    out = torch.zeros_like(v)
    r = v.clone()
    for _ in range(terms):
        t = r.to(torch.bfloat16).to(torch.float32)
        out = out + t
        r = r - t
    return out


class SplitConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, stride, pad, dil):
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, pad, dil, b is not None)
        return F.conv2d(trunc(x), trunc(w), b, stride, pad, dil)

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        stride, pad, dil, has_b = ctx.cfg
        g = trunc(gy)
        gx = torch.nn.grad.conv2d_input(x.shape, trunc(w), g, stride, pad, dil)
        gw = torch.nn.grad.conv2d_weight(trunc(x), w.shape, g, stride, pad, dil)
        gb = gy.sum((0, 2, 3)) if has_b else None
        return gx, gw, gb, None, None, None


def patch(model):
    for m in model.modules():
        if isinstance(m, nn.Conv2d):
            m.forward = (lambda mod: lambda x: SplitConv.apply(x, mod.weight, mod.bias, mod.stride, mod.padding, mod.dilation))(m)


def main_large():
    """the conditioned fixture: same checks as tests/test_gpu_model.py::test_g5l_full_train_step_matches_reference"""
    torch.manual_seed(0)
    g = H.load_golden("g5l_full_train")
    m = R.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    m.load_state_dict(H.conditioned_state_dict(H.shapes_of(m), 1, g["beta_idx"], g["beta_val"]))
    m.train()
    m.classifier.aspp.project[3].eval()
    R.set_bn_momentum(m.backbone, 0.01)
    patch(m)
    img = H.synth_tensor(5, "g5l.img", (2, 3, 128, 128))
    lab = H.synth_labels(5, "g5l.lab", (2, 128, 128), 16, 255, ignore_rows=5)
    lg, ctr, ft = m(img)
    loss = R.ce_over_n(lg, lab, 255)
    loss.backward()
    T = torch.from_numpy
    print("mode", MODE, "fixture g5l")
    print("logits rel err vs fp32 ref %.3e (bar 1e-3)" % H.rel_err(lg[:, :, ::4, ::4], T(g["logits_sub"])))
    print("loss rel %.3e" % (abs(loss.item() - float(g["loss"])) / abs(float(g["loss"]))))
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    rels = []
    for (k, gr), cs in zip(grads.items(), g["grad_checksums"]):
        got = H.checksum(gr)
        rels.append((np.abs(got[1:] - cs[1:]) / (np.abs(cs[1:]) + 1e-300)).max())
    rels = np.array(rels)
    print("gradient checksums: worst rel %.3e median %.3e, %d of %d beyond 2e-3" % (rels.max(), np.median(rels), (rels > 2e-3).sum(), len(rels)))
    for key in [str(k) for k in g["grad_keep"]]:
        ref = T(g["grad__" + key.replace(".", "_")])
        got = grads[key].detach()
        got = got if got.numel() < 70000 else got.contiguous().flatten()[::61]
        print("  grad %-40s rel %.3e (bar 2e-3)" % (key, H.rel_err(got.reshape(ref.shape), ref)))


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "g5l":
        return main_large()
    torch.manual_seed(0)
    g = H.load_golden("g5_full_train")
    m = R.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=1))
    m.train()
    m.classifier.aspp.project[3].eval()
    R.set_bn_momentum(m.backbone, 0.01)
    patch(m)
    img = H.synth_tensor(5, "g5.img", (2, 3, 64, 64))
    lab = H.synth_labels(5, "g5.lab", (2, 64, 64), 16, 255, ignore_rows=3)
    lg, ctr, ft = m(img)
    loss = R.ce_over_n(lg, lab, 255)
    loss.backward()
    T = torch.from_numpy
    print("mode", MODE)
    print("logits rel err vs fp32 ref %.3e (bar 1e-3)" % H.rel_err(lg, T(g["logits"])))
    print("loss rel %.3e" % (abs(loss.item() - float(g["loss"])) / abs(float(g["loss"]))))
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    worst, nbad = 0.0, 0
    for (k, gr), cs in zip(grads.items(), g["grad_checksums"]):
        got = H.checksum(gr)
        rel = np.abs(got[1:] - cs[1:]) / (np.abs(cs[1:]) + 1e-300)
        worst = max(worst, rel.max())
        nbad += int(not np.allclose(got[1:], cs[1:], rtol=2e-3))
    print("gradient checksums: worst rel %.3e, %d of %d beyond 2e-3" % (worst, nbad, len(grads)))
    for key in ("backbone.bn1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer2.0.downsample.0.weight",
                "backbone.layer3.5.bn2.bias", "classifier.project.0.weight", "classifier.aspp.project.1.weight",
                "classifier.classifier.3.weight", "classifier.classifier.3.bias", "backbone.conv1.weight"):
        ref = T(g["grad__" + key.replace(".", "_")])
        got = grads[key].detach()
        got = got if got.numel() < 70000 else got.contiguous().flatten()[::16]
        print("  grad %-40s rel %.3e (bar 2e-3)" % (key, H.rel_err(got.reshape(ref.shape), ref)))


if __name__ == "__main__":
    main()
