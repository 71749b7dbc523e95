"""GPU: the bf16 throughput mode -- the kernels bench.py times (conv_igemm_dma_kernel, conv_wgrad_big_kernel, the fused
BN-backward epilogue, the bit-mask BN kernels) -- gated at model level, plus the fp32 mode at the benchmark's crop size.

The reference's arithmetic is fp32 (network/utils.py:84-118 there) and the fp32 mode keeps the 1e-3 bar.  bf16 storage
cannot: one rounding is 2^-9, and a train-mode BatchNorm network amplifies perturbations so strongly that the bf16-storage
ORACLE itself (oracle/bf16_emu.py) moves its gradients by 1 - cos = 0.08 under a 1e-3 relative change of the image
(tests/tools/bf16_noise.py; on a 2 x 64 x 64 input the gradients are pure noise, 1 - cos = 0.8).  Two gates replace an
element-wise comparison:
  1. locally exact: after a bf16 train step at 768 x 768 every one of the 112 conv + BN units is recomputed from the
     plan's own stored tensors and must match to one bf16 ulp (forward conv, statistics, BN apply, BN backward, gamma /
     beta / weight / data gradients) -- every kernel of the timed configuration on its actual inputs;
  2. statistically equivalent end to end: logits, loss and for EVERY parameter tensor the direction and the norm of the
     gradient are as close to the fp32 oracle as the emulated bf16-storage oracle's are.
"""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu


def _build_hip(dtype, seed, train=True, fp32_products=None):
    import network
    import utils
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=seed))
    m.cuda()
    m.set_compute_dtype(dtype, fp32_products=fp32_products)
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


def _oracle_grads(seed, shape, tag, emulate):
    from oracle import dmlnet_ref as O
    img = H.synth_tensor(seed, tag + ".img", shape)
    lab = H.synth_labels(seed, tag + ".lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
    o = _build_oracle(seed, emulate=emulate)
    lg, _, ft = o(img)
    loss = O.dml_loss(lg, lab, alpha=0.01, ignore_index=255)
    loss.backward()
    return lg.detach(), float(loss), {k: p.grad.detach().double() for k, p in o.named_parameters()}, o


def _vs_truth(grads, truth):
    """per tensor: 1 - cosine to the fp32 oracle's gradient and the ratio of the norms"""
    one_minus_cos, ratio = [], []
    for k, t in truth.items():
        a, b = grads[k].flatten(), t.flatten()
        na, nb = a.norm().item(), b.norm().item()
        one_minus_cos.append(1.0 - (a @ b).item() / (na * nb + 1e-30))
        ratio.append(na / (nb + 1e-30))
    return np.array(one_minus_cos), np.array(ratio)


def _statistical_equivalence(shape, seeds, tag):
    """End to end, bf16 mode.  A deep BatchNorm network in train mode amplifies perturbations: on these inputs a 1e-3
    relative change of the IMAGE moves the bf16-storage oracle's own gradients by 1 - cos = 0.08 (4 x 256^2; tools:
    tests/tools/bf16_noise.py), so two correct bf16 implementations cannot agree element by element.  What a correct
    one must do is sit exactly as far from the fp32 truth as the emulated bf16 oracle does -- logits, loss and, for EVERY
    parameter tensor, direction (cosine) and magnitude (norm ratio) of the gradient.  A mis-scaled or mis-wired
    gradient on any layer class shows as a norm ratio of 2 / 0.5 or a cosine near 0 on those tensors.

    One draw of that comparison is itself a random variable: the per-tensor 1 - cos of BOTH implementations moves by
    +-20 % from input to input (the ASPP image-pooling BatchNorm over B samples amplifies whatever reaches it), so the
    ratio hip / emulation of a single seed scatters by +-0.14.  The gate therefore averages over several seeds (weights
    and inputs); measured with tests/tools/bf16_noise_seeds.py (profiles/r03_bf16_noise_seeds.txt): median x1.05 +- 0.05
    over 10 seeds at 4 x 256^2, x1.04 +- 0.07 over 4 seeds at 2 x 768^2.  Bars: 1.3x the emulation on median and p95
    (round 2: 2.5x), 1.5x on the maximum over the 338 tensors (2x), norm-ratio spread 1.3x."""
    import utils
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    stat = {"hip": [], "emu": []}
    for seed in seeds:
        img = H.synth_tensor(seed, tag + ".img", shape)
        lab = H.synth_labels(seed, tag + ".lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
        m = _build_hip(torch.bfloat16, seed)
        lg, _, ft = m(img.cuda())
        loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
        loss.backward()
        torch.cuda.synchronize()
        g_hip = {k: p.grad.detach().double().cpu() for k, p in m.named_parameters()}
        t_lg, t_loss, g_true, o32 = _oracle_grads(seed, shape, tag, emulate=False)
        e_lg, e_loss, g_emu, oemu = _oracle_grads(seed, shape, tag, emulate=True)
        d_hip, d_emu = H.rel_err(lg, t_lg), H.rel_err(e_lg, t_lg)
        l_hip, l_emu = abs(loss.item() - t_loss) / abs(t_loss), abs(e_loss - t_loss) / abs(t_loss)
        c_hip, r_hip = _vs_truth(g_hip, g_true)
        c_emu, r_emu = _vs_truth(g_emu, g_true)
        names = list(g_true.keys())
        print("%s seed %d vs fp32 oracle | logits: hip %.2e emu %.2e | loss: hip %.2e emu %.2e | 1-cos median/p95/max: hip "
              "%.2e %.2e %.2e (%s) emu %.2e %.2e %.2e | norm ratio median [min,max]: hip %.3f [%.3f (%s), %.3f (%s)] emu %.3f "
              "[%.3f, %.3f]"
              % (tag, seed, d_hip, d_emu, l_hip, l_emu, np.median(c_hip), np.percentile(c_hip, 95), c_hip.max(),
                 names[int(np.argmax(c_hip))], np.median(c_emu), np.percentile(c_emu, 95), c_emu.max(), np.median(r_hip),
                 r_hip.min(), names[int(np.argmin(r_hip))], r_hip.max(), names[int(np.argmax(r_hip))], np.median(r_emu),
                 r_emu.min(), r_emu.max()))
        assert torch.isfinite(lg).all()
        stat["hip"].append((d_hip, l_hip, np.median(c_hip), np.percentile(c_hip, 95), c_hip.max(), np.median(r_hip),
                            r_hip.max() - r_hip.min()))
        stat["emu"].append((d_emu, l_emu, np.median(c_emu), np.percentile(c_emu, 95), c_emu.max(), np.median(r_emu),
                            r_emu.max() - r_emu.min()))
        # per seed: nothing mis-wired or mis-scaled (a wrong gradient is O(1) on these), running statistics equivalent
        assert c_hip.max() <= 2.0 * c_emu.max() + 2e-2, names[int(np.argmax(c_hip))]
        # norm ratio of every tensor, per seed, relative to what the emulation shows on the same seed (measured on eight seeds:
        # spread hip / emulation <= 1.5, largest deviation from 1 <= 1.3x the emulation's + 0.1): a gradient mis-scaled by 1.4x on a
        # single tensor deviates by 0.4 and fails this on most seeds (bars 0.2-0.55)
        dev_hip, dev_emu = max(1 - r_hip.min(), r_hip.max() - 1), max(1 - r_emu.min(), r_emu.max() - 1)
        assert r_hip.max() - r_hip.min() <= 2.0 * (r_emu.max() - r_emu.min()) + 0.05, (r_hip.min(), r_hip.max(), r_emu.min(), r_emu.max())
        assert dev_hip <= 2.0 * dev_emu + 0.05, (dev_hip, dev_emu)
        bufs, obufs = dict(m.named_buffers()), dict(oemu.named_buffers())
        for k in ("backbone.bn1.running_var", "backbone.layer3.11.bn2.running_mean", "backbone.layer4.2.bn3.running_var",
                  "classifier.classifier.1.running_var"):
            assert H.rel_err(bufs[k], obufs[k]) <= max(2.0 * H.rel_err(dict(o32.named_buffers())[k], obufs[k]), 1e-3), k
        del m
        torch.cuda.empty_cache()
    hip, emu = np.array(stat["hip"]).mean(0), np.array(stat["emu"]).mean(0)
    print("%s mean over %d seeds | hip / emulation: logits %.2f loss %.2f | 1-cos median x%.3f p95 x%.3f max x%.3f | norm "
          "median %.3f vs %.3f, spread %.3f vs %.3f" % (tag, len(seeds), hip[0] / emu[0], hip[1] / emu[1], hip[2] / emu[2],
                                                          hip[3] / emu[3], hip[4] / emu[4], hip[5], emu[5], hip[6], emu[6]))
    assert hip[0] <= 1.5 * emu[0] + 1e-3, (hip[0], emu[0])
    assert hip[1] <= 3.0 * emu[1] + 2e-3, (hip[1], emu[1])
    # direction: the distribution over the 338 tensors must be the emulation's
    assert hip[2] <= 1.3 * emu[2] + 2e-3, (hip[2], emu[2])
    assert hip[3] <= 1.3 * emu[3] + 5e-3, (hip[3], emu[3])
    assert hip[4] <= 1.5 * emu[4] + 1e-2, (hip[4], emu[4])
    # magnitude: the bulk as well centred as the emulation's, the spread over the tensors no wider than 1.3x
    assert abs(hip[5] - 1) <= abs(emu[5] - 1) + 0.1, (hip[5], emu[5])
    assert hip[6] <= 1.3 * emu[6] + 0.05, (hip[6], emu[6])


def test_bf16_end_to_end_4x256_statistically_equivalent_to_emulated_oracle():
    _statistical_equivalence((4, 3, 256, 256), (9, 10, 11, 12, 13), "bf16.256")


def test_bf16_end_to_end_768_bs2_statistically_equivalent_to_emulated_oracle():
    _statistical_equivalence((2, 3, 768, 768), (77, 78, 79), "bf16.768")


def _act(a):
    """plan Act (possibly a channel slice of a concat buffer) -> [M, C] float32 on the CPU"""
    off = (a.ptr - a.t.data_ptr()) // a.es
    flat = a.t.view(-1)
    idx = off + torch.arange(a.M, device=flat.device).unsqueeze(1) * a.ld + torch.arange(a.C, device=flat.device).unsqueeze(0)
    return flat[idx].float().cpu()


def _nchw(a2d, B, Hh, Ww):
    return a2d.view(B, Hh, Ww, -1).permute(0, 3, 1, 2).contiguous()


@pytest.mark.parametrize("shape,seed,stage32", [((2, 3, 768, 768), 77, 0), ((3, 3, 96, 160), 9, 0), ((3, 3, 96, 160), 9, 1)])
def test_bf16_plan_every_unit_is_locally_exact(shape, seed, stage32, monkeypatch):
    """The tight gate for the kernels bench.py times.  One bf16 train step; then for EVERY conv + BN (+ residual + ReLU)
    unit of the plan, from the plan's OWN stored tensors: the unit's outputs are recomputed with torch on the CPU in
    fp32 / fp64 and compared at bf16 resolution -- forward conv (conv_igemm_dma_kernel), batch statistics from the fp32
    accumulators, BN apply; backward BN (fused reduce in the data-gradient epilogue or stand-alone), gamma / beta
    gradients, weight gradient (conv_wgrad_big_kernel / conv_wgrad_kernel + split-K fold), data gradient.  Teacher
    forcing removes the chaos of the end-to-end comparison: every kernel is held to 1 bf16 ulp on its actual inputs at
    the benchmark's map sizes.

    Gradients with SEVERAL producers (d(out): the pooled term + four ASPP data gradients; the input of a block with a
    downsample branch: two data gradients, d(low) three; every other block input: conv1's data gradient + the masked
    identity-branch gradient) are checked as sums: the stored gradient against the fp32 sum of all its consumers'
    recomputed contributions -- one bf16 rounding where the plan rounds once (the fused identity add; fp32 staging,
    `stage32` = DML_GRAD_STAGE32=1, DmlConvDesc.acc32), one per producer otherwise."""
    import torch.nn.functional as F
    import utils
    monkeypatch.setenv("DML_GRAD_STAGE32", str(stage32))
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    img = H.synth_tensor(seed, "unit.img", shape)
    lab = H.synth_labels(seed, "unit.lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
    m = _build_hip(torch.bfloat16, seed)
    lg, _, ft = m(img.cuda())
    utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft).backward()
    torch.cuda.synchronize()
    plan = next(p for k, p in m._engine.plans.items() if k[4])
    names = {id(mod): n for n, mod in m.named_modules()}
    from collections import Counter
    consumers = Counter(id(u.x.root) for u in plan.units)
    for u in plan.units:
        if u.res is not None:
            consumers[id(u.res.root)] += 1          # the identity branch adds its gradient to that buffer too
    ULP = 2.0 ** -8                      # round to nearest: the error is at most half an ulp = 2^-9 .. 2^-8 of the value
    worst = {}

    def note(kind, name, err, bar):
        if err > worst.get(kind, (0, ""))[0]:
            worst[kind] = (err, name)
        assert err <= bar, "%s of %s: %.3e > %.1e" % (kind, name, err, bar)

    def relmax(got, ref):
        return (got.double() - ref.double()).abs().max().item() / (ref.double().abs().max().item() + 1e-30)

    n_fused = 0
    gsum, nprod, n_ident, root_of = {}, Counter(), Counter(), {}      # per gradient buffer: fp32 sum of its consumers' contributions

    def contribute(root, g, identity=False):
        k = id(root)
        root_of[k] = root
        gsum[k] = g.double() if k not in gsum else gsum[k] + g.double()
        nprod[k] += 1
        n_ident[k] += 1 if identity else 0

    for u in plan.units:
        name = names[id(u.conv)]
        conv, bn = u.conv, u.bn
        cin = conv.in_channels
        x = _nchw(_act(u.x), u.x.B, u.x.H, u.x.W)[:, :cin]
        wq = conv.weight.detach().float().cpu()
        if u.dtype == torch.bfloat16:                  # (the ASPP image-pooling unit lives in fp32 storage: fp32 weights)
            wq = wq.to(torch.bfloat16).float()
        wq = wq.contiguous().requires_grad_(True)
        has_grad = u.x is u.x.root and u.x.root.grad is not None
        xr = x.clone().requires_grad_(has_grad)
        y_ref = F.conv2d(xr, wq, None, conv.stride, conv.padding, conv.dilation)
        y = _nchw(_act(u.y), u.y.B, u.y.H, u.y.W)
        note("conv fwd (stored y vs fp32 conv of the stored operands)", name, relmax(y, y_ref.detach()), 1.05 * ULP)
        mean, var = y_ref.detach().mean((0, 2, 3)), y_ref.detach().var((0, 2, 3), unbiased=False)
        M0 = y_ref.numel() // y_ref.shape[1]          # samples per channel (the image-pooling unit: the batch size)
        note("batch mean", name, (u.mean.cpu() - mean).abs().max().item() / (var.sqrt().max().item() + 1e-30), 1e-4)
        note("batch invstd", name, relmax(u.invstd.cpu(), torch.rsqrt(var + bn.eps)), 1e-4 if M0 >= 64 else 1e-3)
        gam, bet = bn.weight.detach().float().cpu(), bn.bias.detach().float().cpu()
        sh = (1, -1, 1, 1)
        mu, inv = u.mean.float().cpu(), u.invstd.float().cpu()
        z_ref = (y - mu.view(sh)) * (gam * inv).view(sh) + bet.view(sh)
        if u.res is not None:
            z_ref = z_ref + _nchw(_act(u.res), u.res.B, u.res.H, u.res.W)
        if u.relu:
            z_ref = z_ref.clamp_min(0)
        z = _nchw(_act(u.z), u.z.B, u.z.H, u.z.W)
        note("bn apply (z)", name, relmax(z, z_ref), 1.05 * ULP)
        # ---- backward, from the stored dz / y / z
        dz = _nchw(_act(u.dz), u.z.B, u.z.H, u.z.W).double()
        g = dz * (z > 0) if u.relu else dz
        if getattr(u, "up", None) is not None:
            # the downsample branch reads the BLOCK output's gradient and applies that output's ReLU mask itself (Plan.block_bwd)
            g = dz * (_nchw(_act(u.up.z), u.up.z.B, u.up.z.H, u.up.z.W) > 0)
        xhat = (y.double() - mu.double().view(sh)) * inv.double().view(sh)
        M = y.numel() // y.shape[1]
        dbeta, dgamma = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
        dy_ref = (gam.double() * inv.double()).view(sh) * (g - dbeta.view(sh) / M - xhat * dgamma.view(sh) / M)
        dy = _nchw(_act(u.dy), u.y.B, u.y.H, u.y.W)
        note("bn backward (dy)", name, relmax(dy, dy_ref), 1.5 * ULP)
        note("d gamma", name, relmax(bn.weight.grad.cpu(), dgamma), 2e-3)
        note("d beta", name, relmax(bn.bias.grad.cpu(), dbeta), 2e-3)
        # ---- weight / data gradient from the stored x and dy
        y_ref.backward(dy)
        note("weight gradient", name, relmax(conv.weight.grad.cpu(), wq.grad), 2e-3)
        if xr.requires_grad and consumers[id(u.x.root)] == 1:
            gx = _nchw(_act(u.x.root.grad), u.x.B, u.x.H, u.x.W)[:, :cin]
            note("data gradient", name, relmax(gx, xr.grad), 1.05 * ULP)
        elif xr.requires_grad:
            contribute(u.x.root, xr.grad)
        if u.res is not None and u.res is u.res.root and u.res.root.grad is not None and consumers[id(u.res.root)] > 1:
            contribute(u.res.root, g.float(), identity=True)      # the identity branch hands the masked block-output gradient on
    # the image-pooling branch reads `out` through the global average pool: its input gradient / HW reaches every pixel
    rec = plan.heads[0]
    out_root = rec.branches[0].x.root
    gp = _act(rec.pooled.root.grad)                                            # [B, C]
    contribute(out_root, (gp / (out_root.H * out_root.W)).view(out_root.B, -1, 1, 1).expand(-1, -1, out_root.H, out_root.W))
    n_multi = 0
    for k, tot in gsum.items():
        root = root_of[k]
        stored = _nchw(_act(root.grad), root.B, root.H, root.W)[:, :tot.shape[1]]
        # the identity branch's term is a masked copy of a stored bf16 tensor: it adds no rounding of its own
        roundings = 1 if root.g32 is not None else max(1, nprod[k] - n_ident[k])
        nm = next((names[id(uu.conv)] for uu in plan.units if uu.x is root), "?")
        note("gradient with several producers, %d rounding%s" % (roundings, "s" if roundings > 1 else ""),
             "%d producers, input of %s" % (nprod[k], nm), relmax(stored, tot), 1.05 * ULP * roundings)
        n_multi += 1
    assert n_multi >= 33                      # every block input + d(out) + d(low) (= layer2.0's input)
    dsc_fused = sum(1 for fn, args in plan.bwd if fn is plan.lib.dml_conv_igemm and args[0]._obj.bnr_partials)
    print("units %d, data gradients with the fused BN-backward sums %d; worst: %s"
          % (len(plan.units), dsc_fused, "; ".join("%s %.2e (%s)" % (k, v[0], v[1]) for k, v in worst.items())))
    assert len(plan.units) == 113 - 1          # 112 conv+BN units (the final 1x1 conv has no BN)
    assert dsc_fused >= 60                      # the timed configuration: most reduces run inside the data gradients
    n_acc32 = sum(1 for fn, args in plan.bwd if fn is plan.lib.dml_conv_igemm and args[0]._obj.acc32)
    assert n_acc32 == (5 if stage32 else 0), n_acc32       # d(out) + the four blocks with a downsample branch (low = layer2.0's)


@pytest.mark.parametrize("products", ["exact", "bf16x3", "f16x2"])
def test_fp32_768_bs2_against_oracle(products):
    """fp32 mode at the benchmark's crop size against the fp32 oracle: logits / loss at 1e-3, every gradient checksum -- with the
    exact fp32 MFMA, with the convolutions' products on the bf16 matrix cores (three-term split) and on the fp16 matrix cores
    (two-term split of the scaled operands), same bars"""
    import utils
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    shape, seed = (2, 3, 768, 768), 77
    img = H.synth_tensor(seed, "bf16.768.img", shape)
    lab = H.synth_labels(seed, "bf16.768.lab", (2, 768, 768), 16, 255, ignore_frac=0.05)
    m = _build_hip(torch.float32, seed, fp32_products=products)
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
    print("fp32.768 (%s): logits %.2e features %.2e loss %.2e | grads max-norm median %.2e p95 %.2e max %.2e (%s); "
          "checksum mismatches %d" % (products, e_lg, e_ft, e_loss, np.median(emax), np.percentile(emax, 95), emax.max(),
                                       names[int(np.argmax(emax))], len(cs_bad)))
    assert e_lg <= 1e-3 and e_ft <= 1e-3 and e_loss <= 1e-3
    # 1.2 M pixels per image pair: fp32 summation order alone moves single gradient entries by a few 1e-3 of the tensor's
    # maximum (measured on MI355X: median 3.9e-3, p95 1.7e-2, max 6.5e-2 -- the fp32 oracle is as far from an fp64
    # evaluation, test_gpu_model.py), while the checksums (sum, sum |.|, sum of squares of every tensor) agree to 5e-3:
    # a wiring / scaling error is O(1) on both
    assert np.median(emax) <= 1e-2 and np.percentile(emax, 95) <= 5e-2 and emax.max() <= 0.2
    assert np.median(ecos) <= 1e-4 and ecos.max() <= 1e-2, (np.median(ecos), ecos.max())
    assert not cs_bad, cs_bad[:3]


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
