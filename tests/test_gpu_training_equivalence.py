"""GPU: does TRAINING in the bf16 throughput mode track training in the reference's fp32 arithmetic?

Every other bf16 gate is one forward / backward.  Here the reference's loop (main_embedding.py:458-507: zero_grad ->
forward -> loss -> backward -> SGD with the backbone at 0.1 x lr -> PolyLR, BatchNorm momentum 0.01 on the backbone,
running statistics updated every step) runs for 30 steps on a cycling sequence of batches, on the HIP path in fp32 and
in bf16 from the same weights, and -- as well on the pinned fp32 oracle and on the bf16-storage
emulation of it (oracle/bf16_emu.py; about a minute of CPU time on the GPU box).  The fp32 plan must follow the oracle; the bf16 plan
must stay as close to the fp32 plan as the emulation stays to the oracle (loss curve, weight drift per stage, running
variances), and keep doing so for all 30 steps.

Second test: the HIP path on REFERENCE-INITIALISED weights (torch.manual_seed(1), kaiming fan_out backbone / fan_in head,
BN gamma = 1: resnet.py:154-159, network/utils.py:36-40) instead of the conditioned fixture weights.  With gamma = 1
everywhere the reference's OWN fp32 gradients are 5-20 % from its fp64 evaluation, so the bars
are relative to that gap: the HIP fp32 path must be as close to the fp64 oracle as the fp32 oracle is.
"""
import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu

STAGES = ("stem", "layer1", "layer2", "layer3", "layer4", "head")


def _stage(name):
    if name.startswith("backbone.layer"):
        return name.split(".")[1]
    return "stem" if name.startswith("backbone.") else "head"


def _batches(n, shape, seed):
    """n synthetic batches; labels are blocky (32 x 32 tiles of one class, 5 % ignored) so that there is something to fit"""
    out = []
    B, _, Hh, Ww = shape
    for i in range(n):
        img = H.synth_tensor(seed, "train.img.%d" % i, shape)
        g = H.rng_for(seed, "train.lab.%d" % i)
        coarse = g.integers(0, 16, size=(B, (Hh + 31) // 32, (Ww + 31) // 32))
        lab = np.repeat(np.repeat(coarse, 32, 1), 32, 2)[:, :Hh, :Ww].astype(np.int64)
        lab[g.random(lab.shape) < 0.05] = 255
        out.append((img, torch.from_numpy(lab)))
    return out


def _hip_run(dtype, sd, batches, steps, lr, total, snap_at):
    import network
    import utils
    from dmlnet.optim import FusedSGD
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.load_state_dict(sd)
    m.cuda()
    m.set_compute_dtype(dtype)
    m.train()
    m.classifier.aspp.project[3].eval()
    utils.set_bn_momentum(m.backbone, 0.01)
    opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.1 * lr}, {"params": m.classifier.parameters(), "lr": lr}],
                   lr=lr, momentum=0.9, weight_decay=1e-4).bind(m)
    sched = utils.PolyLR(opt, total, power=0.9)
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=(dtype == torch.bfloat16))
    dev = [(i.cuda(), l.cuda()) for i, l in batches]
    losses, snaps = [], {}
    for it in range(steps):
        img, lab = dev[it % len(dev)]
        opt.zero_grad()
        lg, _, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
        if it + 1 in snap_at:
            snaps[it + 1] = {k: v.detach().float().cpu().clone() for k, v in m.state_dict().items()}
    return np.array(losses), snaps


def _oracle_run(emulate, sd, batches, steps, lr, total, snap_at=()):
    from oracle import bf16_emu
    from oracle import dmlnet_ref as O
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.load_state_dict(sd)
    o.train()
    o.classifier.aspp.project[3].eval()
    O.set_bn_momentum(o.backbone, 0.01)
    if emulate:
        bf16_emu.emulate_bf16_storage(o)
    opt = O.make_optimizer(o, lr=lr)
    base = [g["lr"] for g in opt.param_groups]
    losses, snaps = [], {}
    for it in range(steps):
        img, lab = batches[it % len(batches)]
        losses.append(float(O.train_step(o, opt, img, lab, it, total, base,
                                         lambda lg, y: O.dml_loss(lg, y, alpha=0.01, ignore_index=255))))
        if it + 1 in snap_at:
            snaps[it + 1] = {k: v.detach().float().clone() for k, v in o.state_dict().items()}
    return np.array(losses), snaps


def _drift(snap, sd0, param_names):
    """per stage: the flattened weight change since the start"""
    out = {}
    for s in STAGES:
        out[s] = torch.cat([(snap[k] - sd0[k]).flatten().double() for k in param_names if _stage(k) == s])
    return out


def _cmp_drift(a, b):
    """(norm ratio, 1 - cos) per stage of two drift dicts"""
    res = {}
    for s in STAGES:
        na, nb = a[s].norm().item(), b[s].norm().item()
        res[s] = (na / (nb + 1e-30), 1.0 - (a[s] @ b[s]).item() / (na * nb + 1e-30))
    return res


POOL_BN = "classifier.aspp.convs.4.2"       # BatchNorm over B samples per channel (image-pooling branch): its own row


def _cmp_rv(a, b):
    """per stage: worst relative difference of a BatchNorm running_var vector"""
    res = {s: 0.0 for s in STAGES + ("pool-bn",)}
    for k in a:
        if k.endswith("running_var"):
            s = "pool-bn" if k.startswith(POOL_BN) else _stage(k)
            res[s] = max(res[s], H.rel_err(a[k], b[k]))
    return res


def test_bf16_training_tracks_fp32_over_30_steps():
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    import network
    shape, steps, n_or, lr, total = (4, 3, 256, 256), 30, 30, 0.01, 60
    m0 = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    sd0 = H.synth_state_dict(H.shapes_of(m0), seed=21)
    pnames = [k for k, _ in m0.named_parameters()]
    del m0
    batches = _batches(5, shape, 21)
    import time
    l32, s32 = _hip_run(torch.float32, sd0, batches, steps, lr, total, (6, steps))
    l16, s16 = _hip_run(torch.bfloat16, sd0, batches, steps, lr, total, (6, steps))
    t0 = time.time()
    lo, so = _oracle_run(False, sd0, batches, n_or, lr, total, (6, steps))
    le, se = _oracle_run(True, sd0, batches, n_or, lr, total, (6, steps))
    print("oracle + emulation, %d steps each: %.0f s on %d threads" % (n_or, time.time() - t0, torch.get_num_threads()))
    d_hip = np.abs(l16 - l32) / np.abs(l32)
    d_emu = np.abs(le - lo) / np.abs(lo)
    d_32 = np.abs(l32[:n_or] - lo) / np.abs(lo)
    print("step  loss: fp32 oracle  hip fp32   emulation   hip bf16  |  hip32-oracle  emu-oracle  hip16-hip32")
    for i in range(steps):
        if i < n_or:
            print("%3d   %11.6f %10.6f %11.6f %10.6f  |  %10.2e %10.2e %10.2e" % (i, lo[i], l32[i], le[i], l16[i], d_32[i], d_emu[i], d_hip[i]))
        else:
            print("%3d   %11s %10.6f %11s %10.6f  |  %10s %10s %10.2e" % (i, "", l32[i], "", l16[i], "", "", d_hip[i]))
    checks = []          # (condition, message): everything is printed before anything is asserted

    # (1) the fp32 plan follows the pinned oracle (the dynamics amplify rounding: the bar widens with the step, cf. G8;
    # late in the run two fp32 implementations are 1-4 % apart in the loss of a step)
    for i, tol in enumerate((1e-5, 1e-4, 3e-4, 1e-3, 3e-3, 1e-2)):
        checks.append((d_32[i] <= tol, "fp32 step %d: %.6f vs %.6f" % (i, l32[i], lo[i])))
    checks.append((d_32.max() <= 8e-2, "fp32 plan vs oracle over the run: %.3e" % d_32.max()))
    # (2) the bf16 plan is as close to the fp32 plan as the emulation is to the oracle: step by step over the first six;
    # over the whole run, where the trajectories of ANY two implementations drift apart, no further than twice what the
    # emulation (or the other fp32 implementation) is from the oracle; both runs train, to the same loss
    ref_mean, ref_max = max(d_emu.mean(), d_32.mean()), max(d_emu.max(), d_32.max())
    print("loss deviation over %d steps, mean / max: hip fp32 vs oracle %.2e / %.2e, emulation vs oracle %.2e / %.2e, hip bf16 "
          "vs hip fp32 %.2e / %.2e" % (steps, d_32.mean(), d_32.max(), d_emu.mean(), d_emu.max(), d_hip.mean(), d_hip.max()))
    checks.append(((d_hip[:6] <= 2.0 * d_emu[:6].max() + 2e-3).all(), "first six steps: %s vs %s" % (d_hip[:6], d_emu[:6])))
    checks.append((d_hip.mean() <= 2.0 * ref_mean + 5e-3, "mean loss deviation %.3e vs %.3e" % (d_hip.mean(), ref_mean)))
    checks.append((d_hip.max() <= 2.5 * ref_max + 2e-2, "max loss deviation %.3e vs %.3e" % (d_hip.max(), ref_max)))
    checks.append((l32[-5:].mean() < 0.9 * l32[:5].mean() and l16[-5:].mean() < 0.9 * l16[:5].mean(), "both runs train"))
    # (the scale of "equal" at step 30 is how far two fp32 implementations of the same step -- this plan in fp32 and the oracle --
    # or the emulation and the oracle have drifted apart by then: one draw each of a chaotic trajectory.  The HIP runs are
    # bitwise reproducible, test_gpu_model.py::test_train_steps_are_bitwise_reproducible, so the draw is a fixed one.)
    # The yardstick is taken over the LAST TEN steps, not at the final ones alone: where two trajectories happen to cross at the end
    # (round 4: the two fp32 implementations 0.4 % apart over steps 25-29, 2.5 % over steps 20-29) a five-step window makes the bar
    # a function of that coincidence.
    ref_final = max(abs(le[-5:].mean() - lo[-5:].mean()), abs(l32[-5:].mean() - lo[-5:].mean()))
    ref_late = max(d_emu[-10:].mean(), d_32[-10:].mean())
    checks.append((abs(l16[-5:].mean() - l32[-5:].mean()) <= 2.0 * max(ref_final, ref_late * l32[-5:].mean()) + 3e-2 * l32[-5:].mean(),
                   "final loss level %.4f vs %.4f (fp32 implementations apart by %.4f at the end, %.2e relative over the last ten steps)"
                   % (l16[-5:].mean(), l32[-5:].mean(), ref_final, ref_late)))
    # (3) weight drift per stage and running variances, after 6 and after 30 steps: hip bf16 vs hip fp32 against
    # emulation vs oracle at the same step
    sd0f = {k: v.float() for k, v in sd0.items()}
    print("stage   after  weight drift (norm ratio, 1-cos): hip32/oracle    emu/oracle   hip16/hip32 | running_var rel diff: "
          "hip32/oracle emu/oracle hip16/hip32")
    for at in (6, steps):
        dr_hip = _cmp_drift(_drift(s16[at], sd0f, pnames), _drift(s32[at], sd0f, pnames))
        dr_emu = _cmp_drift(_drift(se[at], sd0f, pnames), _drift(so[at], sd0f, pnames))
        dr_32 = _cmp_drift(_drift(s32[at], sd0f, pnames), _drift(so[at], sd0f, pnames))
        rv_hip, rv_emu, rv_32 = _cmp_rv(s16[at], s32[at]), _cmp_rv(se[at], so[at]), _cmp_rv(s32[at], so[at])
        for s in STAGES + ("pool-bn",):
            if s == "pool-bn":
                print("%-7s %3d %80s | %.2e %.2e %.2e" % (s, at, "", rv_32[s], rv_emu[s], rv_hip[s]))
                continue
            print("%-7s %3d %31s %6.3f %.2e   %6.3f %.2e   %6.3f %.2e | %.2e %.2e %.2e"
                  % (s, at, "", dr_32[s][0], dr_32[s][1], dr_emu[s][0], dr_emu[s][1], dr_hip[s][0], dr_hip[s][1], rv_32[s],
                     rv_emu[s], rv_hip[s]))
        for s in STAGES:
            if at == 6:
                checks.append((dr_32[s][1] <= 1e-2 and abs(dr_32[s][0] - 1) <= 2e-2, "fp32 plan vs oracle, drift %s %s" % (s, dr_32[s])))
            ref_c, ref_n = max(dr_emu[s][1], dr_32[s][1]), max(abs(dr_emu[s][0] - 1), abs(dr_32[s][0] - 1))
            checks.append((dr_hip[s][1] <= 1.5 * ref_c + 1e-2, "direction of the drift @%d %s: %.3e vs %.3e" % (at, s, dr_hip[s][1], ref_c)))
            checks.append((abs(dr_hip[s][0] - 1) <= 2.0 * ref_n + 3e-2, "size of the drift @%d %s: %.3f vs %.3f" % (at, s, dr_hip[s][0], ref_n)))
            checks.append((rv_hip[s] <= 2.0 * max(rv_emu[s], rv_32[s]) + 1e-2, "running_var @%d %s: %.3e vs %.3e" % (at, s, rv_hip[s], rv_emu[s])))
        checks.append((rv_hip["pool-bn"] <= 3.0 * max(rv_emu["pool-bn"], rv_32["pool-bn"]) + 5e-2,
                       "pooled-branch running_var @%d: %.3e vs %.3e" % (at, rv_hip["pool-bn"], rv_emu["pool-bn"])))
    bad = [msg for ok, msg in checks if not ok]
    assert not bad, bad


def test_reference_initialised_weights_fp32_within_the_fp32_vs_fp64_gap():
    """torch.manual_seed(1) + the reference's init (the bench's weights): HIP fp32 step vs the fp64 oracle, bars set by how
    far the fp32 ORACLE is from the same fp64 evaluation."""
    import network
    import utils
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    torch.manual_seed(1)
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    gam = [v for k, v in sd.items() if k.endswith("bn3.weight")]
    assert all(float((g - 1).abs().max()) == 0 for g in gam)          # gamma = 1 everywhere: NOT the conditioned fixture weights
    w = sd["backbone.layer3.5.conv2.weight"]
    assert abs(float(w.std()) - (2.0 / (256 * 9)) ** 0.5) < 0.05 * (2.0 / (256 * 9)) ** 0.5      # kaiming fan_out (resnet.py:156)
    shape = (2, 3, 128, 128)
    img = H.synth_tensor(31, "refinit.img", shape)
    lab = H.synth_labels(31, "refinit.lab", (2, 128, 128), 16, 255, ignore_frac=0.05)
    m.cuda().train()
    m.set_compute_dtype(torch.float32)
    m.classifier.aspp.project[3].eval()
    lg, _, ft = m(img.cuda())
    loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
    loss.backward()
    torch.cuda.synchronize()
    res = {}
    for name, dt in (("fp32", torch.float32), ("fp64", torch.float64)):
        o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
        o.load_state_dict(sd)
        o = o.to(dt)
        o.train()
        o.classifier.aspp.project[3].eval()
        olg, _, _ = o(img.to(dt))
        ol = O.dml_loss(olg, lab, alpha=0.01, ignore_index=255)
        ol.backward()
        res[name] = (olg.detach().double(), float(ol), [p.grad.detach().double().flatten() for p in o.parameters()])
    t_lg, t_loss, t_g = res["fp64"]
    o_lg, o_loss, o_g = res["fp32"]
    h_g = [p.grad.detach().double().cpu().flatten() for p in m.parameters()]

    def gerr(gs):
        e, c = [], []
        for a, b in zip(gs, t_g):
            e.append((a - b).abs().max().item() / (b.abs().max().item() + 1e-30))
            c.append(1.0 - (a @ b).item() / (a.norm().item() * b.norm().item() + 1e-30))
        return np.array(e), np.array(c)

    eh, ch = gerr(h_g)
    eo, co = gerr(o_g)
    lh, lo_ = H.rel_err(lg, t_lg), H.rel_err(o_lg, t_lg)
    print("reference-initialised weights, vs fp64 oracle | logits: hip %.2e oracle-fp32 %.2e | loss: hip %.2e oracle-fp32 %.2e | "
          "grad max-norm err median/p95/max: hip %.2e %.2e %.2e oracle-fp32 %.2e %.2e %.2e | 1-cos median/max: hip %.2e %.2e "
          "oracle-fp32 %.2e %.2e" % (lh, lo_, abs(loss.item() - t_loss) / abs(t_loss), abs(o_loss - t_loss) / abs(t_loss),
                                     np.median(eh), np.percentile(eh, 95), eh.max(), np.median(eo), np.percentile(eo, 95), eo.max(),
                                     np.median(ch), ch.max(), np.median(co), co.max()))
    assert torch.isfinite(lg).all()
    assert lh <= 3.0 * lo_ + 1e-4
    assert abs(loss.item() - t_loss) <= 3.0 * abs(o_loss - t_loss) + 1e-4 * abs(t_loss)
    assert np.median(eh) <= 2.0 * np.median(eo) + 1e-3
    assert np.percentile(eh, 95) <= 2.0 * np.percentile(eo, 95) + 5e-3
    assert eh.max() <= 2.0 * eo.max() + 2e-2
    assert np.median(ch) <= 2.0 * np.median(co) + 1e-5 and ch.max() <= 2.0 * co.max() + 1e-3
