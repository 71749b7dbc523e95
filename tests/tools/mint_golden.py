#!/usr/bin/env python3
"""Mint the golden vectors in tests/golden/ from the REAL reference (authoring container only).

Imports ``/root/reference`` with the shims of SURVEY.md Appendix B (fake torchvision, identity
``.cuda()``), runs the reference's own modules on seeded inputs, cross-checks the CPU oracle
(``oracle/dmlnet_ref.py``) against them, and writes small ``.npz`` fixtures.  Nothing from the
reference travels: fixtures hold inputs/seeds and expected outputs only.

    python tests/tools/mint_golden.py            # regenerate everything (~2 min on 8 cores)
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True
import helpers as H  # noqa: E402
from oracle import dmlnet_ref as O  # noqa: E402

REF = "/root/reference"
OUT = os.path.join(ROOT, "tests", "golden")


def install_shims():
    tv, tvm, tvmu = (types.ModuleType(n) for n in ("torchvision", "torchvision.models",
                                                   "torchvision.models.utils"))
    tvmu.load_state_dict_from_url = torch.hub.load_state_dict_from_url
    tv.models, tvm.utils = tvm, tvmu
    sys.modules.update({"torchvision": tv, "torchvision.models": tvm, "torchvision.models.utils": tvmu})
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def save(name, **arrays):
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = np.asarray(v)
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **conv)
    print("  wrote %s (%.1f KB)" % (path, os.path.getsize(path) / 1024))


def assert_close(a, b, tol, what):
    err = H.max_abs(a, b)
    scale = float(b.detach().abs().max()) + 1e-12
    print("  oracle vs reference: %-34s max|d|=%.3e (scale %.3e)" % (what, err, scale))
    assert err <= tol * max(1.0, scale), what


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.set_num_threads(8)
    install_shims()
    sys.path.insert(0, os.path.join(REF, "DeepLabV3Plus-Pytorch"))
    import network as R  # the reference package
    from network import utils as RU
    from network.backbone import resnet as RR
    ref_loss = load_by_path("ref_loss", os.path.join(REF, "DeepLabV3Plus-Pytorch/utils/loss.py"))
    ref_sched = load_by_path("ref_sched", os.path.join(REF, "DeepLabV3Plus-Pytorch/utils/scheduler.py"))

    # ------------------------------------------------------------------ G1 distance head
    print("G1 distance head")
    x = H.synth_tensor(11, "g1.x", (2, 16, 12, 20), scale=2.0)
    stub = RU._SimpleSegmentationModel_embedding(nn.Identity(), nn.Identity())
    lg, ctr, ft = stub(x)
    olg, octr, oft = O.distance_head(x)
    assert_close(olg, lg, 1e-6, "logits")
    assert_close(oft, ft, 0, "features_out")
    assert_close(octr, ctr, 0, "centers")
    # general K x C prototypes (novel-class prototypes are arbitrary vectors, test_embedding.py:254-257)
    protos = H.synth_tensor(11, "g1.protos", (5, 16))
    f64 = ft.double().unsqueeze(3) - protos.double()
    lg_gen = -(f64 ** 2).sum(-1).permute(0, 3, 1, 2).float()
    assert_close(O.distance_head(x, protos)[0], lg_gen, 1e-6, "logits (general prototypes)")
    save("g1_distance_head", x=x, logits=lg, features=ft, centers=ctr, argmax=lg.argmax(1),
         protos=protos, logits_general=lg_gen)

    # ------------------------------------------------------------------ G2 losses
    print("G2 losses")
    sys.path.insert(0, os.path.join(REF, "anomaly"))
    with contextlib.redirect_stdout(io.StringIO()):
        import models as amodels          # /root/reference/anomaly/models (prints its 13x13 centers)

    class Enc(nn.Module):
        def forward(self, x, return_feature_maps=False):
            return x

    class Dec(nn.Module):
        def forward(self, x, segSize=None):
            return x

    logit = (H.synth_tensor(12, "g2.logit", (3, 13, 20, 24), scale=3.0)).requires_grad_(True)
    label = H.synth_labels(12, "g2.label", (3, 20, 24), 13, -1, ignore_frac=0.1)
    sm = amodels.SegmentationModule(Enc(), Dec(), nn.CrossEntropyLoss(ignore_index=-1))
    loss, acc = sm({"img_data": logit, "seg_label": label})
    loss.backward()
    g_ref = logit.grad.clone()
    lo = logit.detach().clone().requires_grad_(True)
    oloss = O.dml_loss(lo, label, alpha=0.01, ignore_index=-1)
    oloss.backward()
    assert_close(oloss, loss.detach().reshape(()), 1e-6, "DML loss")
    assert_close(lo.grad, g_ref, 1e-6, "DML dL/dlogit")
    assert_close(O.dml_loss_loop(lo.detach(), label), loss.detach().reshape(()), 1e-6, "DML loss (loop form)")
    assert_close(O.pixel_acc(lo.detach(), label), acc, 1e-7, "pixel acc")
    # DeepLab-side live loss: CE / n with ignore 255 (utils/loss.py:34-42)
    logit2 = H.synth_tensor(12, "g2.logit2", (2, 16, 12, 20), scale=3.0).requires_grad_(True)
    label2 = H.synth_labels(12, "g2.label2", (2, 12, 20), 16, 255, ignore_rows=2)
    crit = ref_loss.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)
    l2 = crit(logit2, label2, None)
    l2.backward()
    lo2 = logit2.detach().clone().requires_grad_(True)
    ol2 = O.ce_over_n(lo2, label2, 255)
    ol2.backward()
    assert_close(ol2, l2.detach(), 1e-6, "CE/n loss")
    assert_close(lo2.grad, logit2.grad, 1e-6, "CE/n grad")
    save("g2_losses", logit=logit.detach(), label=label, loss=loss.detach().reshape(()), acc=acc,
         grad=g_ref, logit2=logit2.detach(), label2=label2, loss2=l2.detach(), grad2=logit2.grad)

    # ------------------------------------------------------------------ G3 head only
    print("G3 DeepLabHeadV3Plus (train-mode BN, dropout off)")
    head = RU.DeepLabHeadV3Plus(2048, 256, 16, [6, 12, 18])
    shapes = OrderedDict(("classifier." + k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = H.synth_state_dict(shapes, seed=3)
    head.load_state_dict(OrderedDict((k[len("classifier."):], v) for k, v in sd.items()))
    head.train()
    head.aspp.project[3].eval()                       # F14: dropout off
    low = H.synth_tensor(3, "g3.low", (2, 256, 16, 16)).requires_grad_(True)
    out = H.synth_tensor(3, "g3.out", (2, 2048, 4, 4)).requires_grad_(True)
    wgt = H.synth_tensor(3, "g3.wgt", (2, 16, 16, 16))
    y = head({"low_level": low, "out": out})
    (y * wgt).sum().backward()
    ohead = O._Head(16, (6, 12, 18))
    ohead.load_state_dict(OrderedDict((k[len("classifier."):], v) for k, v in sd.items()))
    ohead.train()
    ohead.aspp.project[3].eval()
    lo_, out_ = low.detach().clone().requires_grad_(True), out.detach().clone().requires_grad_(True)
    oy = ohead({"low_level": lo_, "out": out_})
    (oy * wgt).sum().backward()
    assert_close(oy, y.detach(), 1e-5, "head output")
    assert_close(lo_.grad, low.grad, 1e-5, "head d/dlow")
    assert_close(out_.grad, out.grad, 1e-5, "head d/dout")
    pg = {k: p.grad for k, p in head.named_parameters()}
    opg = {k: p.grad for k, p in ohead.named_parameters()}
    for k in pg:
        assert H.max_abs(opg[k], pg[k]) <= 1e-4 * (1 + float(pg[k].abs().max())), k
    grad_sums = np.stack([H.checksum(pg[k]) for k in pg])
    keep = ["project.0.weight", "project.1.weight", "aspp.convs.0.0.weight", "aspp.convs.4.1.weight",
            "aspp.convs.4.2.bias", "aspp.project.1.weight", "classifier.1.bias", "classifier.3.weight",
            "classifier.3.bias"]
    extra = {"grad__" + k.replace(".", "_"): pg[k] for k in keep}
    # a strided sample of the three big dilated-conv / decoder grads
    extra["grad_sample__aspp_convs_2_0_weight"] = pg["aspp.convs.2.0.weight"][::8, ::64]
    extra["grad_sample__classifier_0_weight"] = pg["classifier.0.weight"][::8, ::8]
    bufs = dict(head.named_buffers())
    save("g3_head", y=y.detach(), dlow=low.grad, dout=out.grad, grad_names=np.array(list(pg.keys())),
         grad_checksums=grad_sums, rm_project=bufs["project.1.running_mean"],
         rv_project=bufs["project.1.running_var"], rm_pool=bufs["aspp.convs.4.2.running_mean"],
         rv_pool=bufs["aspp.convs.4.2.running_var"], rv_cls=bufs["classifier.1.running_var"], **extra)

    # ------------------------------------------------------------------ G4 bottlenecks
    print("G4 Bottleneck blocks")
    g4 = {}
    cases = {"s1": dict(inplanes=64, planes=16, stride=1, dilation=1, ds=False),
             "s2": dict(inplanes=32, planes=16, stride=2, dilation=1, ds=True),
             "d2": dict(inplanes=64, planes=16, stride=1, dilation=2, ds=False)}
    for name, c in cases.items():
        ds = None
        if c["ds"]:
            ds = nn.Sequential(RR.conv1x1(c["inplanes"], c["planes"] * 4, c["stride"]),
                               nn.BatchNorm2d(c["planes"] * 4))
        blk = RR.Bottleneck(c["inplanes"], c["planes"], c["stride"], ds, dilation=c["dilation"])
        shapes = OrderedDict(("backbone.blk." + k, tuple(v.shape)) for k, v in blk.state_dict().items())
        sd = H.synth_state_dict(shapes, seed=4)
        strip = OrderedDict((k[len("backbone.blk."):], v) for k, v in sd.items())
        blk.load_state_dict(strip)
        blk.train()
        xin = H.synth_tensor(4, "g4.x." + name, (2, c["inplanes"], 8, 8)).requires_grad_(True)
        yb = blk(xin)
        wb = H.synth_tensor(4, "g4.w." + name, tuple(yb.shape))
        (yb * wb).sum().backward()
        ob = O._Bottleneck(dict(inplanes=c["inplanes"], planes=c["planes"], stride=c["stride"],
                                dilation=c["dilation"], downsample=c["ds"]))
        ob.load_state_dict(strip)
        ob.train()
        xo = xin.detach().clone().requires_grad_(True)
        oyb = ob(xo)
        (oyb * wb).sum().backward()
        assert_close(oyb, yb.detach(), 1e-5, "bottleneck %s out" % name)
        assert_close(xo.grad, xin.grad, 1e-5, "bottleneck %s dx" % name)
        g4[name + "_y"] = yb.detach()
        g4[name + "_dx"] = xin.grad
        for k, p in blk.named_parameters():
            g4[name + "_grad__" + k.replace(".", "_")] = p.grad
        for k, b in blk.named_buffers():
            if "num_batches" not in k:
                g4[name + "_buf__" + k.replace(".", "_")] = b
    save("g4_bottleneck", **g4)

    # ------------------------------------------------------------------ G5 full model
    print("G5 full model (train-mode BN, dropout off) 2x3x64x64")
    ref = R.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    orc = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    assert list(ref.state_dict().keys()) == list(orc.state_dict().keys()), "state_dict keys differ"
    assert len(ref.state_dict()) == 674
    shapes = H.shapes_of(ref)
    assert shapes == H.shapes_of(orc)
    sd = H.synth_state_dict(shapes, seed=1)
    ref.load_state_dict(sd)
    orc.load_state_dict(sd)
    for m in (ref, orc):
        m.train()
        m.classifier.aspp.project[3].eval()
        O.set_bn_momentum(m.backbone, 0.01)
    img = H.synth_tensor(5, "g5.img", (2, 3, 64, 64))
    lab = H.synth_labels(5, "g5.lab", (2, 64, 64), 16, 255, ignore_rows=3)
    crit = ref_loss.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)
    # fp64 run of the same reference modules = "exact" arithmetic; tells how much of 1e-3 the reference's own
    # fp32 rounding already uses up on this input
    ref.double()
    lg64, _, ft64 = ref(img.double())
    crit(lg64, lab, ft64).backward()
    g64 = OrderedDict((k, p.grad.clone()) for k, p in ref.named_parameters())
    ref.float()
    ref.load_state_dict(sd)
    for p in ref.parameters():
        p.grad = None
    for m_ in ref.modules():
        if isinstance(m_, nn.BatchNorm2d):
            m_.reset_running_stats()
    ref.load_state_dict(sd)
    lg, ctr, ft = ref(img)
    lg64 = lg64.detach()
    ref_noise = float((lg.detach().double() - lg64).abs().max() / lg64.abs().max())
    print("  reference fp32 vs fp64 on this input: logits %.3e (relative to max |logit|)" % ref_noise)
    loss = crit(lg, lab, ft)
    loss.backward()
    olg, octr, oft = orc(img)
    oloss = O.ce_over_n(olg, lab, 255)
    oloss.backward()
    assert_close(olg, lg.detach(), 1e-4, "full logits")
    assert_close(oft, ft.detach(), 1e-4, "full features_out")
    assert_close(oloss, loss.detach(), 1e-5, "full loss")
    rg = OrderedDict((k, p.grad) for k, p in ref.named_parameters())
    og = OrderedDict((k, p.grad) for k, p in orc.named_parameters())
    worst = max(H.max_abs(og[k], rg[k]) / (float(rg[k].abs().max()) + 1e-12) for k in rg)
    print("  oracle vs reference: worst relative param-grad error %.3e" % worst)
    assert worst < 2e-3
    gnoise = np.array([float((rg[k].double() - g64[k]).abs().max() / (g64[k].abs().max() + 1e-30)) for k in rg])
    print("  reference fp32 vs fp64 param grads (max-norm): median %.2e max %.2e" % (np.median(gnoise), gnoise.max()))
    keep = ["backbone.conv1.weight", "backbone.bn1.weight", "backbone.layer1.0.conv1.weight",
            "backbone.layer2.0.downsample.0.weight", "backbone.layer3.5.bn2.bias",
            "backbone.layer4.2.conv3.weight", "classifier.project.0.weight",
            "classifier.aspp.project.1.weight", "classifier.classifier.3.weight",
            "classifier.classifier.3.bias"]
    extra = {"grad__" + k.replace(".", "_"): (rg[k] if rg[k].numel() < 70000 else rg[k].flatten()[::16])
             for k in keep}
    rb = dict(ref.named_buffers())
    save("g5_full_train", logits=lg.detach(), logits64=lg64.float(), ref_noise=ref_noise, grad_noise=gnoise,
         loss=loss.detach(),
         grad_names=np.array(list(rg.keys())), grad_checksums=np.stack([H.checksum(g) for g in rg.values()]),
         rm_stem=rb["backbone.bn1.running_mean"], rv_stem=rb["backbone.bn1.running_var"],
         rv_l4=rb["backbone.layer4.2.bn3.running_var"], rv_head=rb["classifier.classifier.1.running_var"],
         **extra)

    # ---- G8 trajectory: SGD(2 groups) + PolyLR on the same model / batch (main_embedding.py:385-392,458-507)
    print("G8 6-step SGD/PolyLR trajectory on the full model (2x3x64x64)")
    img = H.synth_tensor(5, "g8.img", (2, 3, 64, 64))
    lab = H.synth_labels(5, "g8.lab", (2, 64, 64), 16, 255, ignore_rows=3)
    ref.load_state_dict(sd)
    for p in ref.parameters():
        p.grad = None
    lr, total = 0.0002, 20
    opt = torch.optim.SGD([{"params": ref.backbone.parameters(), "lr": 0.1 * lr},
                           {"params": ref.classifier.parameters(), "lr": lr}],
                          lr=lr, momentum=0.9, weight_decay=1e-4)
    sched = ref_sched.PolyLR(opt, total, power=0.9)
    losses, lrs = [], []
    for it in range(6):
        opt.zero_grad()
        lg, _, ft = ref(img)
        ls = crit(lg, lab, ft)
        ls.backward()
        opt.step()
        sched.step()
        losses.append(float(ls))
        lrs.append([g["lr"] for g in opt.param_groups])
    # oracle trajectory
    orc.load_state_dict(sd)
    for p in orc.parameters():
        p.grad = None
    oopt = O.make_optimizer(orc, lr=lr, weight_decay=1e-4)
    olosses = [float(O.train_step(orc, oopt, img, lab, it, total, [0.1 * lr, lr],
                                  lambda a, b: O.ce_over_n(a, b, 255))) for it in range(6)]
    print("  ref losses   ", ["%.6f" % v for v in losses])
    print("  oracle losses", ["%.6f" % v for v in olosses])
    assert np.allclose(losses, olosses, rtol=2e-4)
    assert np.allclose(lrs[-1], [g["lr"] for g in oopt.param_groups], rtol=1e-6)
    fin = ref.state_dict()
    save("g8_trajectory", losses=np.array(losses), lrs=np.array(lrs), lr=lr, total_itrs=total,
         w_stem=fin["backbone.conv1.weight"], w_last=fin["classifier.classifier.3.weight"],
         b_last=fin["classifier.classifier.3.bias"], rm_stem=fin["backbone.bn1.running_mean"],
         w_checksums=np.stack([H.checksum(v.float()) for v in fin.values()]))

    # ------------------------------------------------------------------ G5b eval forward, config #1
    print("G5b eval forward 1x3x256x256 (BASELINE config #1) with calibrated running stats")
    ref.load_state_dict(sd)
    ref.train()
    ref.classifier.aspp.project[3].eval()
    for m in ref.modules():
        if isinstance(m, nn.BatchNorm2d):
            m.momentum = 1.0
    cal = H.synth_tensor(5, "g5b.calib", (2, 3, 128, 128))
    with torch.no_grad():
        ref(cal)
    ref.eval()
    stats = torch.cat([v.flatten() for k, v in ref.state_dict().items()
                       if k.endswith("running_mean") or k.endswith("running_var")])
    img1 = H.synth_tensor(5, "g5b.img", (1, 3, 256, 256))
    with torch.no_grad():
        lg, ctr, ft = ref(img1)
    orc.load_state_dict(ref.state_dict())
    orc.eval()
    with torch.no_grad():
        olg, _, oft = orc(img1)
    assert_close(olg, lg, 1e-4, "eval logits")
    print("  eval logits range [%.3f, %.3f]" % (float(lg.min()), float(lg.max())))
    proto_car = H.synth_tensor(5, "g5b.proto", (16,), scale=0.5)
    save("g5b_full_eval", bn_stats=stats, logits_sub=lg[:, :, ::4, ::4], logits_checksum=H.checksum(lg),
         feats_sub=ft[:, ::4, ::4, :], argmax=lg.argmax(1).to(torch.uint8), centers=ctr)

    # ------------------------------------------------------------------ G6 bilinear
    print("G6 bilinear (align_corners=False)")
    a = H.synth_tensor(6, "g6.a", (1, 3, 5, 7))
    b = H.synth_tensor(6, "g6.b", (2, 4, 1, 1))
    c = H.synth_tensor(6, "g6.c", (1, 2, 5, 7))
    a.requires_grad_(True)
    c.requires_grad_(True)
    ua = F.interpolate(a, size=(20, 28), mode="bilinear", align_corners=False)
    wa = H.synth_tensor(6, "g6.wa", (1, 3, 20, 28))
    (ua * wa).sum().backward()
    uc = F.interpolate(c, size=(13, 17), mode="bilinear", align_corners=False)
    wc = H.synth_tensor(6, "g6.wc", (1, 2, 13, 17))
    (uc * wc).sum().backward()
    save("g6_bilinear", a=a.detach(), ua=ua.detach(), wa=wa, da=a.grad, b=b,
         ub=F.interpolate(b, size=(6, 5), mode="bilinear", align_corners=False),
         c=c.detach(), uc=uc.detach(), wc=wc, dc=c.grad)

    # ------------------------------------------------------------------ G7 scoring (harness numpy)
    print("G7 dissum / MSP / novel-prototype relabel (harness arithmetic restated from the drivers)")
    lg7 = -(H.synth_tensor(7, "g7.lg", (1, 16, 9, 11)) ** 2 + 0.05) * 20
    ft7 = H.synth_tensor(7, "g7.ft", (1, 9, 11, 16), scale=0.2)
    lg7[0, :, 0, 0] = -100.0                              # force a clipped pixel (sum 1600 > 1000)
    out = lg7.squeeze().numpy()
    # test_embedding.py:349-350,365
    dsm = -np.sum(out, axis=0)
    dsm[dsm > 1000] = 1000
    dsm_n = (dsm - np.min(dsm)) / (np.max(dsm) - np.min(dsm))
    # anomaly/eval_ood_traditional.py:301-305
    ds2 = -np.sum(out, axis=0)
    ds2[ds2 >= 400] = 400
    ds2_n = (ds2 - np.min(ds2)) / (np.max(ds2) - np.min(ds2))
    assert np.array_equal(O.dissum_score(out, 1000, False), dsm_n)
    assert np.array_equal(O.dissum_score(out, 400, True), ds2_n)
    # test_embedding.py:339-341
    preds = lg7.max(dim=1)[1].numpy()
    msp = 1 - F.softmax(lg7, dim=1).max(dim=1)[0].numpy()
    # test_embedding.py:254-257,428-445
    shots = [H.synth_tensor(7, "g7.shot%d" % i, (16,), scale=0.3).numpy().astype(np.float64).tolist()
             for i in range(5)]
    proto = np.zeros((16,))
    for i in range(5):
        proto += np.array(shots[i])
    proto /= 5
    feats = ft7.view(1, 9 * 11, 16).squeeze().numpy()
    dcar = -np.sum((feats - proto) ** 2, axis=1).reshape(9, 11)
    rel = preds.copy()
    rel[0][np.logical_and(dcar > -1.5, dcar > lg7.max(dim=1)[0].squeeze().numpy())] = 16
    assert np.array_equal(O.novel_relabel(preds[0], out, ft7[0].numpy(), O.mean_prototype(shots)), rel[0])
    assert (rel[0] == 16).any() and (rel[0] != 16).any()
    save("g7_scoring", logits=lg7, feats=ft7, dissum_deeplab=dsm_n, dissum_anomaly=ds2_n, preds=preds,
         msp=msp, shots=np.array(shots), proto=proto, dcar=dcar, relabel=rel)
    print("all golden vectors minted; oracle agrees with the reference on every one")


if __name__ == "__main__":
    main()
