"""CPU: the oracle (oracle/dmlnet_ref.py) against the golden vectors minted from the real reference.

These are the pins of SURVEY.md §8(c) G1-G8.  Tolerances: the oracle ran bit-identical to the
reference in the authoring container; 1e-5 relative leaves room for a different CPU / thread count.
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as H
from oracle import dmlnet_ref as O

T = torch.from_numpy


def close(a, b, rtol=1e-5):
    assert H.max_abs(a, b) <= rtol * (1.0 + float(b.abs().max())), (H.max_abs(a, b), float(b.abs().max()))


def test_g1_distance_head():
    g = H.load_golden("g1_distance_head")
    lg, ctr, ft = O.distance_head(T(g["x"]))
    close(lg, T(g["logits"]))
    assert torch.equal(ft, T(g["features"]))
    assert torch.equal(ctr, T(g["centers"]))
    assert torch.equal(lg.argmax(1), T(g["argmax"]))
    close(O.distance_head(T(g["x"]), T(g["protos"]))[0], T(g["logits_general"]))
    # F5: with 3*I prototypes the logit has the closed form -|f|^2 + 6 f_k - 9
    f = T(g["features"])
    closed = (-(f ** 2).sum(-1, keepdim=True) + 6 * f - 9).permute(0, 3, 1, 2)
    close(closed, T(g["logits"]), 1e-5)


def test_g2_losses():
    g = H.load_golden("g2_losses")
    lo = T(g["logit"]).requires_grad_(True)
    loss = O.dml_loss(lo, T(g["label"]), alpha=0.01, ignore_index=-1)
    loss.backward()
    close(loss, T(g["loss"]), 1e-6)
    close(lo.grad, T(g["grad"]), 1e-6)
    close(O.dml_loss_loop(lo.detach(), T(g["label"])), T(g["loss"]), 1e-6)
    close(O.pixel_acc(lo.detach(), T(g["label"])), T(g["acc"]), 1e-6)
    lo2 = T(g["logit2"]).requires_grad_(True)
    l2 = O.ce_over_n(lo2, T(g["label2"]), 255)
    l2.backward()
    close(l2, T(g["loss2"]), 1e-6)
    close(lo2.grad, T(g["grad2"]), 1e-6)


def _head(seed=3):
    head = O._Head(16, (6, 12, 18))
    shapes = OrderedDict(("classifier." + k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = H.synth_state_dict(shapes, seed=seed)
    head.load_state_dict(OrderedDict((k[len("classifier."):], v) for k, v in sd.items()))
    head.train()
    head.aspp.project[3].eval()
    return head


def test_g3_head():
    g = H.load_golden("g3_head")
    head = _head()
    low = H.synth_tensor(3, "g3.low", (2, 256, 16, 16)).requires_grad_(True)
    out = H.synth_tensor(3, "g3.out", (2, 2048, 4, 4)).requires_grad_(True)
    wgt = H.synth_tensor(3, "g3.wgt", (2, 16, 16, 16))
    y = head({"low_level": low, "out": out})
    (y * wgt).sum().backward()
    close(y, T(g["y"]))
    close(low.grad, T(g["dlow"]))
    close(out.grad, T(g["dout"]))
    pg = dict((k, p.grad) for k, p in head.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    for n, cs in zip(names, g["grad_checksums"]):
        assert np.allclose(H.checksum(pg[n]), cs, rtol=1e-4, atol=1e-6), n
    close(pg["classifier.3.weight"], T(g["grad__classifier_3_weight"]), 1e-4)
    close(pg["aspp.convs.2.0.weight"][::8, ::64], T(g["grad_sample__aspp_convs_2_0_weight"]), 1e-4)
    bufs = dict(head.named_buffers())
    close(bufs["project.1.running_var"], T(g["rv_project"]))
    close(bufs["aspp.convs.4.2.running_mean"], T(g["rm_pool"]))


@pytest.mark.parametrize("name,cfg", [
    ("s1", dict(inplanes=64, planes=16, stride=1, dilation=1, downsample=False)),
    ("s2", dict(inplanes=32, planes=16, stride=2, dilation=1, downsample=True)),
    ("d2", dict(inplanes=64, planes=16, stride=1, dilation=2, downsample=False)),
])
def test_g4_bottleneck(name, cfg):
    g = H.load_golden("g4_bottleneck")
    blk = O._Bottleneck(cfg)
    shapes = OrderedDict(("backbone.blk." + k, tuple(v.shape)) for k, v in blk.state_dict().items())
    sd = H.synth_state_dict(shapes, seed=4)
    blk.load_state_dict(OrderedDict((k[len("backbone.blk."):], v) for k, v in sd.items()))
    blk.train()
    x = H.synth_tensor(4, "g4.x." + name, (2, cfg["inplanes"], 8, 8)).requires_grad_(True)
    y = blk(x)
    w = H.synth_tensor(4, "g4.w." + name, tuple(y.shape))
    (y * w).sum().backward()
    close(y, T(g[name + "_y"]))
    close(x.grad, T(g[name + "_dx"]))
    for k, p in blk.named_parameters():
        close(p.grad, T(g[name + "_grad__" + k.replace(".", "_")]), 1e-4)
    for k, b in blk.named_buffers():
        if "num_batches" not in k:
            close(b, T(g[name + "_buf__" + k.replace(".", "_")]))


def full_model(train=True):
    m = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=1))
    if train:
        m.train()
        m.classifier.aspp.project[3].eval()
        O.set_bn_momentum(m.backbone, 0.01)
    return m


def test_state_dict_contract():
    m = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    sd = m.state_dict()
    assert len(sd) == 674
    assert sum(p.numel() for p in m.parameters()) == 58752688
    assert "backbone.layer3.22.conv3.weight" in sd and "classifier.aspp.convs.4.2.running_var" in sd
    assert "classifier.classifier.3.bias" in sd and not any("centers" in k for k in sd)
    names = [str(n) for n in H.load_golden("g5_full_train")["grad_names"]]
    assert names == [k for k, _ in m.named_parameters()]


def test_g5_full_train_and_g8_trajectory():
    g = H.load_golden("g5_full_train")
    m = full_model()
    img = H.synth_tensor(5, "g5.img", (2, 3, 64, 64))
    lab = H.synth_labels(5, "g5.lab", (2, 64, 64), 16, 255, ignore_rows=3)
    lg, ctr, ft = m(img)
    loss = O.ce_over_n(lg, lab, 255)
    loss.backward()
    close(lg, T(g["logits"]), 1e-4)
    close(lg, T(g["logits64"]), 1e-4)
    close(loss, T(g["loss"]), 1e-5)
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    for (k, gr), cs in zip(grads.items(), g["grad_checksums"]):
        assert np.allclose(H.checksum(gr)[1:], cs[1:], rtol=1e-3), k
    close(grads["classifier.classifier.3.bias"], T(g["grad__classifier_classifier_3_bias"]), 1e-4)
    close(grads["backbone.conv1.weight"], T(g["grad__backbone_conv1_weight"]), 1e-3)
    close(dict(m.named_buffers())["backbone.bn1.running_var"], T(g["rv_stem"]))
    # G8: 6 SGD steps with two LR groups + PolyLR
    t = H.load_golden("g8_trajectory")
    m = full_model()
    img = H.synth_tensor(5, "g8.img", (2, 3, 64, 64))
    lab = H.synth_labels(5, "g8.lab", (2, 64, 64), 16, 255, ignore_rows=3)
    lr, total = float(t["lr"]), int(t["total_itrs"])
    opt = O.make_optimizer(m, lr=lr, weight_decay=1e-4)
    losses = [float(O.train_step(m, opt, img, lab, it, total, [0.1 * lr, lr],
                                 lambda a, b: O.ce_over_n(a, b, 255))) for it in range(6)]
    assert np.allclose(losses, t["losses"], rtol=1e-3), (losses, t["losses"])
    assert np.allclose([gp["lr"] for gp in opt.param_groups], t["lrs"][-1], rtol=1e-6)
    close(m.state_dict()["classifier.classifier.3.bias"], T(t["b_last"]), 1e-3)
    close(m.state_dict()["backbone.conv1.weight"], T(t["w_stem"]), 1e-3)


def test_nonsquare_conditioning_record_g13n():
    """tests/golden/g13n_nonsquare.npz (tests/tools/mint_golden_nonsquare.py) carries the moved BatchNorm betas of the non-square
    2 x 3 x 128 x 192 live-oracle check and the proof numbers the mint script asserted on the reference."""
    g = H.load_golden("g13n_nonsquare")
    assert int(g["seed"]) == 11 and tuple(int(v) for v in g["shape"]) == (2, 3, 128, 192)
    assert float(g["relu_margin"]) >= 64.0 and float(g["relu_margin_over_noise"]) >= 6.0 and int(g["relu_elems"]) > 16_000_000
    assert (g["relu_margins"] >= np.maximum(64.0, 6.0 * g["relu_fp32_noise"])).all() and int(g["beta_moved"]) == g["beta_idx"].size
    m = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    sd = H.conditioned_state_dict(H.shapes_of(m), 11, g["beta_idx"], g["beta_val"])
    plain = H.synth_state_dict(H.shapes_of(m), seed=11)
    moved = [k for k in sd if not torch.equal(sd[k], plain[k])]
    assert moved and all(k.endswith(".bias") and k[:-4] + "running_mean" in sd for k in moved)       # only BatchNorm betas
    assert max(float((sd[k] - plain[k]).abs().max()) for k in moved) <= float(g["beta_max_delta"]) + 1e-12
    m.load_state_dict(sd)


@pytest.mark.parametrize("os_", [8, 16])
def test_variant_conditioning_records_g14v(os_):
    """tests/golden/g14v_os<OS>.npz (tests/tools/mint_golden_variants.py): moved BatchNorm betas of the factory-variant checks and the
    proof numbers asserted on the reference; they load into every embedding width of that output stride."""
    g = H.load_golden("g14v_os%d" % os_)
    assert int(g["seed"]) == 21 and tuple(int(v) for v in g["shape"]) == (2, 3, 64, 80)
    assert float(g["relu_margin"]) >= 64.0 and float(g["relu_margin_over_noise"]) >= 6.0
    assert (g["relu_margins"] >= np.maximum(64.0, 6.0 * g["relu_fp32_noise"])).all() and int(g["beta_moved"]) == g["beta_idx"].size
    for K in (13, 32):
        m = O.deeplabv3plus_embedding_resnet101(num_classes=K, output_stride=os_)
        sd = H.conditioned_state_dict(H.shapes_of(m), 21, g["beta_idx"], g["beta_val"])
        plain = H.synth_state_dict(H.shapes_of(m), seed=21)
        moved = [k for k in sd if not torch.equal(sd[k], plain[k])]
        assert moved and all(k.endswith(".bias") and k[:-4] + "running_mean" in sd for k in moved)
        m.load_state_dict(sd)


def test_large_conditioned_fixtures_g5l_g8l_g12l():
    """The well-conditioned 128 x 128 fixtures (tests/tools/mint_golden_large.py: every ReLU input of the network at least
    64 x eps32 x sum|terms| -- and 6 x the reference's own fp32-vs-fp64 noise -- away from zero): the oracle reproduces them, and the
    proof numbers the mint script stored are the ones it asserts."""
    g = H.load_golden("g5l_full_train")
    assert float(g["relu_margin"]) >= 64.0 and float(g["relu_margin_over_noise"]) >= 6.0 and int(g["relu_elems"]) > 11_000_000
    assert (g["relu_margins"] >= np.maximum(64.0, 6.0 * g["relu_fp32_noise"])).all() and int(g["beta_moved"]) == g["beta_idx"].size
    m = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    sd = H.conditioned_state_dict(H.shapes_of(m), 1, g["beta_idx"], g["beta_val"])
    plain = H.synth_state_dict(H.shapes_of(m), seed=1)
    moved = [k for k in sd if not torch.equal(sd[k], plain[k])]
    assert moved and all(k.endswith(".bias") and k[:-4] + "running_mean" in sd for k in moved)       # only BatchNorm betas
    assert max(float((sd[k] - plain[k]).abs().max()) for k in moved) <= float(g["beta_max_delta"]) + 1e-12
    m.load_state_dict(sd)
    m.train()
    m.classifier.aspp.project[3].eval()
    O.set_bn_momentum(m.backbone, 0.01)
    img = H.synth_tensor(5, "g5l.img", (2, 3, 128, 128))
    lab = H.synth_labels(5, "g5l.lab", (2, 128, 128), 16, 255, ignore_rows=5)
    lg, ctr, ft = m(img)
    loss = O.ce_over_n(lg, lab, 255)
    loss.backward()
    close(lg[:, :, ::4, ::4], T(g["logits_sub"]), 1e-4)
    close(lg[:, :, ::4, ::4], T(g["logits64_sub"]), 1e-4)
    close(loss, T(g["loss"]), 1e-5)
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    assert [str(n) for n in g["grad_names"]] == list(grads.keys())
    for (k, gr), cs in zip(grads.items(), g["grad_checksums"]):
        assert np.allclose(H.checksum(gr)[1:], cs[1:], rtol=3e-4), k
    # G8L: six SGD / PolyLR steps from the same weights, bars stored in the fixture (8 x the reference's own run-to-run deviation)
    t = H.load_golden("g8l_trajectory")
    assert np.array_equal(t["beta_idx"], g["beta_idx"]) and np.array_equal(t["beta_val"], g["beta_val"])
    m.load_state_dict(sd)
    for p in m.parameters():
        p.grad = None
    lr, total = float(t["lr"]), int(t["total_itrs"])
    opt = O.make_optimizer(m, lr=lr, weight_decay=1e-4)
    losses = np.array([float(O.train_step(m, opt, img, lab, it, total, [0.1 * lr, lr], lambda a, b: O.ce_over_n(a, b, 255)))
                       for it in range(6)])
    assert (np.abs(losses - t["losses"]) <= t["bars"] * np.abs(t["losses"])).all(), (losses, t["losses"])
    assert np.allclose([gp["lr"] for gp in opt.param_groups], t["lrs"][-1], rtol=1e-6)
    fin = m.state_dict()
    for i, (k, bar) in enumerate(zip([str(k) for k in t["wkeys"]], t["wbars"])):
        got = fin[k].float() if fin[k].numel() < 70000 else fin[k].float().flatten()[::61]
        ref = T(t["w_%d" % i])
        assert float((got.reshape(ref.shape) - ref).abs().max()) <= bar * float(ref.abs().max()), k
    # G12L: the two-head model
    g2 = H.load_golden("g12l_multihead")
    assert float(g2["relu_margin"]) >= 64.0 and float(g2["relu_margin_over_noise"]) >= 6.0
    m2 = O.deeplabv3plus_embedding_self_distillation_resnet101(output_stride=16)
    sd2 = H.conditioned_state_dict(H.shapes_of(m2), 12, g2["beta_idx"], g2["beta_val"])
    assert len(sd2) == int(g2["n_keys"]) and list(sd2.keys())[-4:] == [str(k) for k in g2["keys"]]
    m2.load_state_dict(sd2)
    m2.train()
    m2.classifier.aspp.project[3].eval()
    m2.classifier_1.aspp.project[3].eval()
    img2 = H.synth_tensor(12, "g12l.img", (2, 3, 128, 128))
    lab2 = H.synth_labels(12, "g12l.lab", (2, 128, 128), 17, 255, ignore_rows=5)
    logits, centers, feats = m2(img2)
    loss2 = O.ce_over_n(logits[-1], lab2, 255)
    loss2.backward()
    close(logits[0][:, :, ::4, ::4], T(g2["logits0_sub"]), 1e-5)
    close(logits[1][:, :, ::4, ::4], T(g2["logits1_sub"]), 1e-5)
    close(loss2, T(g2["loss"]), 1e-6)
    live = OrderedDict((k, p.grad) for k, p in m2.named_parameters() if p.grad is not None)
    assert [str(n) for n in g2["grad_names"]] == list(live.keys())
    for (k, gr), cs in zip(live.items(), g2["grad_checksums"]):
        assert np.allclose(H.checksum(gr)[1:], cs[1:], rtol=3e-4), k


def load_bn_stats(model, flat):
    off = 0
    sd = model.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            n = v.numel()
            v.copy_(flat[off:off + n].view_as(v))
            off += n
    assert off == flat.numel()


def test_g5b_eval_config1():
    """BASELINE config #1: 1x3x256x256 eval forward on CPU."""
    g = H.load_golden("g5b_full_eval")
    m = full_model(train=False)
    load_bn_stats(m, T(g["bn_stats"]))
    m.eval()
    with torch.no_grad():
        lg, ctr, ft = m(H.synth_tensor(5, "g5b.img", (1, 3, 256, 256)))
    assert lg.shape == (1, 16, 256, 256) and ft.shape == (1, 256, 256, 16) and ctr.shape == (16, 16)
    close(lg[:, :, ::4, ::4], T(g["logits_sub"]), 1e-4)
    close(ft[:, ::4, ::4, :], T(g["feats_sub"]), 1e-4)
    assert np.allclose(H.checksum(lg), g["logits_checksum"], rtol=1e-4)
    agree = (lg.argmax(1).to(torch.uint8) == T(g["argmax"])).float().mean().item()
    assert agree > 0.999


def test_g6_bilinear():
    g = H.load_golden("g6_bilinear")
    close(O.bilinear(T(g["a"]), (20, 28)), T(g["ua"]), 1e-6)
    close(O.bilinear(T(g["b"]), (6, 5)), T(g["ub"]), 1e-6)
    c = T(g["c"]).requires_grad_(True)
    uc = O.bilinear(c, (13, 17))
    (uc * T(g["wc"])).sum().backward()
    close(uc, T(g["uc"]), 1e-6)
    close(c.grad, T(g["dc"]), 1e-5)
    # Appendix A: exact x4 weights cycle through 0.625, 0.875, 0.125, 0.375
    ramp = torch.arange(4.0).view(1, 1, 1, 4)
    up = O.bilinear(ramp, (1, 16)).flatten()
    assert torch.allclose(up[2:6], torch.tensor([0.125, 0.375, 0.625, 0.875]))


def test_g7_scoring():
    g = H.load_golden("g7_scoring")
    out = g["logits"][0]
    assert np.array_equal(O.dissum_score(out, 1000, False), g["dissum_deeplab"])
    assert np.array_equal(O.dissum_score(out, 400, True), g["dissum_anomaly"])
    assert np.allclose(O.msp_score(T(g["logits"])).numpy(), g["msp"], atol=1e-6)
    proto = O.mean_prototype(g["shots"].tolist())
    assert np.allclose(proto, g["proto"])
    rel = O.novel_relabel(g["preds"][0], out, g["feats"][0], proto)
    assert np.array_equal(rel, g["relabel"][0])


def test_poly_lr():
    assert O.poly_lr(0.1, 0, 100) == pytest.approx(0.1)
    assert O.poly_lr(0.1, 50, 100) == pytest.approx(0.1 * 0.5 ** 0.9)
    assert O.poly_lr(0.1, 100, 100) == pytest.approx(1e-6)       # utils/scheduler.py:10 min_lr floor


def test_stage_plan_matches_reference_dilations():
    plan = O.stage_plan(16)
    l4 = [c for c in plan if c["stage"] == 4]
    assert [c["dilation"] for c in l4] == [1, 2, 2] and [c["stride"] for c in l4] == [1, 1, 1]
    plan8 = O.stage_plan(8)
    l3 = [c for c in plan8 if c["stage"] == 3]
    assert l3[0]["dilation"] == 1 and l3[1]["dilation"] == 2
    assert [c["dilation"] for c in plan8 if c["stage"] == 4][:2] == [2, 4]
    assert len(plan) == 33


def test_g12_multihead_self_distillation_model():
    """shared backbone + 16- and 17-prototype heads (utils.py:120-193), loss on the last head only"""
    g = H.load_golden("g12_multihead")
    m = O.deeplabv3plus_embedding_self_distillation_resnet101(output_stride=16)
    sd = H.synth_state_dict(H.shapes_of(m), seed=12)
    assert len(sd) == int(g["n_keys"]) and list(sd.keys())[-4:] == [str(k) for k in g["keys"]]
    m.load_state_dict(sd)
    m.train()
    m.classifier.aspp.project[3].eval()
    m.classifier_1.aspp.project[3].eval()
    img = H.synth_tensor(12, "g12.img", (2, 3, 64, 64))
    lab = H.synth_labels(12, "g12.lab", (2, 64, 64), 17, 255, ignore_rows=3)
    logits, centers, feats = m(img)
    assert [tuple(l.shape) for l in logits] == [(2, 16, 64, 64), (2, 17, 64, 64)] and centers[1].shape == (17, 17)
    loss = O.ce_over_n(logits[-1], lab, 255)
    loss.backward()
    close(logits[0][:, :, ::4, ::4], T(g["logits0_sub"]), 1e-5)
    close(logits[1][:, :, ::4, ::4], T(g["logits1_sub"]), 1e-5)
    close(feats[1][:, ::4, ::4, :], T(g["feats1_sub"]), 1e-5)
    close(loss, T(g["loss"]), 1e-6)
    grads = dict((k, p.grad) for k, p in m.named_parameters())
    for i, k in enumerate(str(k) for k in g["grad_keys"]):
        assert np.allclose(H.checksum(grads[k])[1:], g["grad_%d_checksum" % i][1:], rtol=1e-4), k
    for k in (str(k) for k in g["untouched"]):
        assert grads[k] is None


def test_bf16_storage_emulation_is_the_same_model_when_rounding_is_off(monkeypatch):
    """oracle/bf16_emu.py only inserts roundings: with the rounding replaced by the identity the patched model must
    reproduce the pinned oracle (logits, loss, every gradient, running statistics) -- so what the bf16 parity tests
    compare against is the reference's graph, not a second restatement of it."""
    from oracle import bf16_emu
    img = H.synth_tensor(5, "g5.img", (2, 3, 64, 64))
    lab = H.synth_labels(5, "g5.lab", (2, 64, 64), 16, 255, ignore_rows=3)

    def run(patch):
        o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
        o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=1))
        o.train()
        o.classifier.aspp.project[3].eval()
        O.set_bn_momentum(o.backbone, 0.01)
        if patch:
            bf16_emu.emulate_bf16_storage(o)
        lg, _, _ = o(img)
        loss = O.dml_loss(lg, lab, alpha=0.01, ignore_index=255)
        loss.backward()
        return o, lg.detach(), loss.detach()

    ref, lg_ref, loss_ref = run(False)
    monkeypatch.setattr(bf16_emu, "q", lambda t: t)
    emu, lg_emu, loss_emu = run(True)
    close(lg_emu, lg_ref, 2e-5)
    close(loss_emu, loss_ref, 1e-5)
    for (k, p), (_, r) in zip(emu.named_parameters(), ref.named_parameters()):
        assert H.max_abs(p.grad, r.grad) <= 2e-4 * (float(r.grad.abs().max()) + 1e-12), k
    for (k, b), (_, r) in zip(emu.named_buffers(), ref.named_buffers()):
        close(b.float(), r.float(), 1e-5)
    monkeypatch.undo()
    # with the rounding on the result moves by a bf16-sized amount, not more
    _, lg_q, loss_q = run(True)
    d = H.rel_err(lg_q, lg_ref)
    assert 1e-4 < d < 0.15, d          # measured 6.5e-2: ~100 layers amplify the 2^-9 roundings ~16x
    assert abs(float(loss_q) - float(loss_ref)) < 5e-2 * abs(float(loss_ref))
    assert torch.equal(bf16_emu.q(torch.tensor([1.0 + 2 ** -9, 1.0 + 3 * 2 ** -9])), torch.tensor([1.0, 1.0 + 2 ** -7]))
