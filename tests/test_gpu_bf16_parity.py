"""GPU: the bf16 throughput mode -- the kernels bench.py times (conv_igemm_dma_kernel, conv_wgrad_big_kernel, the fused
BN-backward epilogue, the bit-mask BN kernels) -- against the oracle run with bf16 storage emulated at the points the
HIP plan rounds (oracle/bf16_emu.py): logits, loss and EVERY parameter gradient at model level.

Why not 1e-3: the reference's arithmetic is fp32 (network/utils.py:84-118 of the reference); in bf16 storage mode a single
rounding is 2^-9 = 2e-3 relative, and the HIP path and the emulation differ in fp32 summation order, so individual bf16
values flip by one ulp and the flips propagate through ~100 layers.  The bars below are 2-3x what was measured on
MI355X (printed by the tests) -- an order of magnitude inside what a mis-scaled gradient on any layer class would give
(a wrong factor on one layer moves that tensor's relative error to O(1)) -- and the fp32 mode keeps the 1e-3 bar
(test_gpu_model.py, test_fp32_768_against_oracle below).
"""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _build_hip(dtype, seed, train=True):
    import network
    import utils
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=seed))
    m.cuda()
    m.set_compute_dtype(dtype)
    if train:
        m.train()
        m.classifier.aspp.project[3].eval()
        utils.set_bn_momentum(m.backbone, 0.01)
    else:
        m.eval()
    return m


def _build_oracle(seed, emulate, train=True, dtype=torch.float32):
    from oracle import bf16_emu
    from oracle import dmlnet_ref as O
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=seed))
    o = o.to(dtype)
    if train:
        o.train()
        o.classifier.aspp.project[3].eval()
        O.set_bn_momentum(o.backbone, 0.01)
    else:
        o.eval()
    if emulate:
        bf16_emu.emulate_bf16_storage(o)
    return o


def _grad_errors(m, o):
    """per parameter tensor: max-norm error and 1 - cosine, relative to the oracle's gradient"""
    emax, ecos, names = [], [], []
    for (k, p), (k2, q) in zip(m.named_parameters(), o.named_parameters()):
        assert k == k2
        a, b = p.grad.detach().double().cpu().flatten(), q.grad.detach().double().flatten()
        sc = b.abs().max().item() + 1e-30
        emax.append((a - b).abs().max().item() / sc)
        ecos.append(1.0 - (a @ b).item() / (a.norm().item() * b.norm().item() + 1e-30))
        names.append(k)
    return np.array(emax), np.array(ecos), names


def _train_step_compare(shape, seed, tag, bars):
    import utils
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    img = H.synth_tensor(seed, tag + ".img", shape)
    lab = H.synth_labels(seed, tag + ".lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
    m = _build_hip(torch.bfloat16, seed)
    lg, _, ft = m(img.cuda())
    loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
    loss.backward()
    torch.cuda.synchronize()
    o = _build_oracle(seed, emulate=True)
    olg, _, oft = o(img)
    oloss = O.dml_loss(olg, lab, alpha=0.01, ignore_index=255)
    oloss.backward()
    # how far bf16 storage itself moves the result (emulated oracle vs the plain fp32 oracle): context for the bars
    o32 = _build_oracle(seed, emulate=False)
    with torch.no_grad():
        ref32, _, _ = o32(img)
    e_lg = H.rel_err(lg, olg)
    e_ft = H.rel_err(ft, oft)
    e_loss = abs(loss.item() - oloss.item()) / abs(oloss.item())
    emax, ecos, names = _grad_errors(m, o)
    worst = int(np.argmax(emax))
    print("%s: logits rel %.2e (bf16 storage vs fp32 oracle: %.2e), features %.2e, loss %.2e | grads max-norm: median "
          "%.2e p95 %.2e max %.2e (%s) | 1-cos: median %.2e p95 %.2e max %.2e"
          % (tag, e_lg, H.rel_err(olg, ref32), e_ft, e_loss, np.median(emax), np.percentile(emax, 95), emax.max(),
             names[worst], np.median(ecos), np.percentile(ecos, 95), ecos.max()))
    assert torch.isfinite(lg).all()
    assert e_lg <= bars["logits"] and e_ft <= bars["logits"], (e_lg, e_ft)
    assert e_loss <= bars["loss"], e_loss
    assert np.median(emax) <= bars["g_med"], np.median(emax)
    assert np.percentile(emax, 95) <= bars["g_p95"], np.percentile(emax, 95)
    assert emax.max() <= bars["g_max"], (emax.max(), names[worst])
    assert ecos.max() <= bars["cos_max"], (ecos.max(), names[int(np.argmax(ecos))])
    # running statistics come from the fp32 accumulators on both sides
    bufs, obufs = dict(m.named_buffers()), dict(o.named_buffers())
    for k in ("backbone.bn1.running_var", "backbone.layer3.11.bn2.running_mean", "backbone.layer4.2.bn3.running_var",
              "classifier.classifier.1.running_var"):
        assert H.rel_err(bufs[k], obufs[k]) <= bars["logits"], k


BARS_SMALL = dict(logits=2e-2, loss=5e-3, g_med=2e-2, g_p95=6e-2, g_max=0.25, cos_max=3e-2)
BARS_768 = dict(logits=2e-2, loss=5e-3, g_med=2e-2, g_p95=6e-2, g_max=0.25, cos_max=3e-2)


def test_bf16_train_step_g5_sized_against_emulated_oracle():
    _train_step_compare((2, 3, 64, 64), 5, "bf16.g5", BARS_SMALL)


def test_bf16_train_step_nonsquare_against_emulated_oracle():
    _train_step_compare((3, 3, 96, 128), 9, "bf16.fresh", BARS_SMALL)


def test_bf16_train_step_768_bs2_against_emulated_oracle():
    """the benchmark's crop size: large-map-only code paths (two-stage BN finalize, conv_wgrad_big_kernel, the
    192 x 192 layers, 31-bit offset fast path)"""
    _train_step_compare((2, 3, 768, 768), 77, "bf16.768", BARS_768)


def test_fp32_768_bs2_against_oracle():
    """fp32 mode at the benchmark's crop size against the fp32 oracle: logits / loss at 1e-3, every gradient checksum"""
    import utils
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    shape, seed = (2, 3, 768, 768), 77
    img = H.synth_tensor(seed, "bf16.768.img", shape)
    lab = H.synth_labels(seed, "bf16.768.lab", (2, 768, 768), 16, 255, ignore_frac=0.05)
    m = _build_hip(torch.float32, seed)
    lg, _, ft = m(img.cuda())
    loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
    loss.backward()
    torch.cuda.synchronize()
    o = _build_oracle(seed, emulate=False)
    olg, _, oft = o(img)
    oloss = O.dml_loss(olg, lab, alpha=0.01, ignore_index=255)
    oloss.backward()
    e_lg, e_ft = H.rel_err(lg, olg), H.rel_err(ft, oft)
    e_loss = abs(loss.item() - oloss.item()) / abs(oloss.item())
    emax, ecos, names = _grad_errors(m, o)
    cs_bad = []
    for (k, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        a, b = H.checksum(p.grad), H.checksum(q.grad)
        if not np.allclose(a[1:], b[1:], rtol=5e-3):
            cs_bad.append((k, a, b))
    print("fp32.768: logits %.2e features %.2e loss %.2e | grads max-norm median %.2e p95 %.2e max %.2e (%s); "
          "checksum mismatches %d" % (e_lg, e_ft, e_loss, np.median(emax), np.percentile(emax, 95), emax.max(),
                                       names[int(np.argmax(emax))], len(cs_bad)))
    assert e_lg <= 1e-3 and e_ft <= 1e-3 and e_loss <= 1e-3
    # the fp32 oracle itself sits a few 1e-3 from an fp64 evaluation on single tensors at this size (ReLU sign flips);
    # a wiring / scaling error is O(1)
    assert np.median(emax) <= 1e-3 and np.percentile(emax, 95) <= 5e-3 and emax.max() <= 5e-2
    assert len(cs_bad) <= 3, cs_bad[:3]


def test_config2_forward_only_768_bs8_bf16():
    """BASELINE configs[1]: forward-only 768 x 768, 8 images, bf16, no grad.  Eval-mode plan (BN + residual + ReLU in the
    conv epilogues) at the full size: properties of the head at bs = 8, and parity with the emulated oracle at bs = 2
    (the same plan code, a size the CPU finishes in seconds)."""
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    m = _build_hip(torch.bfloat16, 1, train=False)
    g = torch.Generator().manual_seed(808)
    img = torch.randn(8, 3, 768, 768, generator=g)
    with torch.no_grad():
        lg, ctr, ft = m(img.cuda())
    assert lg.shape == (8, 16, 768, 768) and ft.shape == (8, 768, 768, 16) and not lg.requires_grad
    assert torch.isfinite(lg).all()
    closed = (-(ft * ft).sum(-1, keepdim=True) + 6 * ft - 9).permute(0, 3, 1, 2)      # F5: 3*I prototypes
    assert (lg - closed).abs().max().item() <= 1e-4 * lg.abs().max().item()
    assert (lg.argmax(1) == ft.argmax(-1)).float().mean().item() > 0.9999
    # images are independent in eval mode: the first two of the batch equal a batch of two
    with torch.no_grad():
        lg2, _, _ = m(img[:2].cuda())
    assert H.rel_err(lg2, lg[:2]) <= 1e-6
    o = _build_oracle(1, emulate=True, train=False)
    with torch.no_grad():
        olg, _, oft = o(img[:2])
    e = H.rel_err(lg2, olg)
    print("config2 bf16 eval forward vs emulated oracle: logits rel %.2e" % e)
    assert e <= 2e-2
    agree = (lg2.argmax(1).cpu() == olg.argmax(1)).float().mean().item()
    assert agree > 0.98, agree
