#!/usr/bin/env python3
"""Mint the WELL-CONDITIONED full-model fixtures (authoring container only; imports the real reference):

    g5l_full_train   one train step of deeplabv3plus_embedding_resnet101, 2 x 3 x 128 x 128
    g8l_trajectory   six SGD / PolyLR steps of the same model and batch shape
    g12l_multihead   one train step of the self-distillation (two-head) model, loss on the last head

Why they exist.  The 64 x 64 fixtures (g5 / g8 / g12) normalise layer3 / layer4 / ASPP over 2 x 4 x 4 = 32 samples, and nothing in
them keeps a pre-ReLU value away from zero: an element within rounding of the threshold has its ReLU mask -- hence one element of the
backward, hence percent of that channel's dbeta and per mille of everything behind it -- decided by the summation order of the conv
in front of it.  Two correct fp32-accurate implementations can therefore disagree on such a fixture beyond its own bars (round 4: the
two-plane mode crossed one edge per fixture after a K-order change).  A fixture must not sit on a knife edge.  These do not:

  * 128 x 128 inputs: every BatchNorm of the backbone / ASPP branches / decoder sees >= 128 samples;
  * CONDITIONING PROVED, not assumed.  The reference model runs in fp64 with hooks on its own nn.Conv2d / nn.BatchNorm2d / nn.ReLU
    modules; for every ReLU input z = sum_k BN_k(conv_k(x_k)) (+ identity) the script forms the magnitude sum
          A = sum_k [ |gamma| invstd ( conv(|x|, |w|) + |mean| ) + |beta| ]  +  |identity|
    -- the "sum of |terms|" any rounding-error bound of that value is proportional to -- and REQUIRES
          |z| >= FACT * eps32 * A      (FACT = 64, eps32 = 2^-23)
    for EVERY element of EVERY ReLU input of the network.  A random draw never satisfies that (~1e-4 of 1e7 elements fall inside), so
    the weights are conditioned: walking the ReLUs in forward order (one pass: everything upstream of a ReLU is final when it is
    reached), each channel with an element inside SAFETY * band gets its BatchNorm beta moved by the smallest |delta| (a few 1e-4) that
    puts all of the channel's samples outside.  The moved betas travel in the fixture (sparse: index into the concatenation of all
    BatchNorm biases, fp32 value); everything else is helpers.synth_state_dict as before.
  * The local bound is not the whole story: the INPUT of a deep layer already differs between two implementations (propagated
    rounding), and the block-output ReLUs (bn3 + identity: a small-gamma branch on top of a long residual sum) see that at 100-200 x
    eps32 * A between the reference's own fp32 and fp64 runs, where a plain conv -> BN -> ReLU unit sees 2-7.  So the script measures
    that noise per ReLU (|z32 - z64| / (eps32 A), reference fp32 against reference fp64) and conditions a second time with the
    requirement  |z| >= max(FACT, NOISE_TUNE x measured noise of this ReLU) * eps32 * A.
  * The proof is re-run on the FINAL fp32 weights without any modification, in fp64 and in fp32: every ReLU must have
    margin >= max(FACT, NOISE_REQ x its own measured fp32-vs-fp64 noise), and every ReLU mask of the fp32 run must equal the fp64 one.
    Smallest margin, smallest margin / noise ratio and the noise itself are printed and stored.
  * g8l: the weights of steps 1..5 cannot be conditioned (they are the optimizer's), so the trajectory's bars are DERIVED from the
    reference itself and stored in the fixture, one rule for every arithmetic mode: at step t an implementation may deviate from the
    reference's fp32 run by 8 x the largest deviation, over steps <= t, between the reference's own runs (fp32 with 8 threads, fp32
    with 1 thread, fp64), floor 1e-5.

    python tests/tools/mint_golden_large.py            (~10 min on 8 cores)
"""
from __future__ import annotations

import os
import sys
from collections import OrderedDict

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "tools")]
sys.dont_write_bytecode = True
import helpers as H  # noqa: E402
import mint_golden as MG  # noqa: E402
from oracle import dmlnet_ref as O  # noqa: E402

EPS32 = 2.0 ** -23
FACT = 64.0            # required margin: |z| >= FACT * eps32 * sum|terms| ...
NOISE_TUNE = 8.0       # ... and >= NOISE_TUNE x the ReLU's measured fp32-vs-fp64 noise when conditioning,
NOISE_REQ = 6.0        # >= NOISE_REQ x the noise measured on the final weights in the proof
SAFETY = 2.0           # the conditioning pass clears SAFETY x the band, so that the re-check on the rounded weights holds with room


# ---------------------------------------------------------------------------------------------------
# hooks: magnitude sums of every ReLU input, on the reference's own modules
# ---------------------------------------------------------------------------------------------------
class ReluAudit:
    """mode "tune": move BatchNorm betas so that no ReLU input lies within SAFETY * FACT * eps32 * A of zero (fp64 pass);
    mode "record": store z and A of every ReLU input;  mode "compare": against a recorded pass (same order)."""

    def __init__(self, model: nn.Module, mode: str, recorded=None, req=None):
        self.model, self.mode, self.recorded = model, mode, recorded
        self.req = req                 # tune: required margin per ReLU call (None: FACT everywhere)
        self.margins, self.noise = [], []
        self.last_abs = None
        self.pending = []              # BatchNorm outputs since the last ReLU: (module, out, A)
        self.records = []              # per ReLU call: (z fp32 copy as float64 -> stored as float64 numpy, A)
        self.moved = OrderedDict()     # BatchNorm module -> number of channels moved
        self.max_delta = 0.0
        self.min_margin = float("inf")
        self.n_elems = 0
        self.k = 0
        self.mask_diff = 0
        self.noise_use = 0.0
        self.handles = []
        for m in model.modules():
            if isinstance(m, nn.Conv2d):
                self.handles.append(m.register_forward_hook(self._conv))
            elif isinstance(m, nn.BatchNorm2d):
                self.handles.append(m.register_forward_hook(self._bn))
            elif isinstance(m, nn.ReLU):
                self.handles.append(m.register_forward_pre_hook(self._relu))

    def close(self):
        for h in self.handles:
            h.remove()

    def _conv(self, mod, inp, out):
        x = inp[0].detach()
        self.last_abs = F.conv2d(x.abs().double(), mod.weight.detach().abs().double(), None, mod.stride, mod.padding, mod.dilation)
        if mod.bias is not None:
            self.last_abs = self.last_abs + mod.bias.detach().abs().double().view(1, -1, 1, 1)

    def _bn(self, mod, inp, out):
        x = inp[0].detach().double()
        assert mod.training and self.last_abs is not None and self.last_abs.shape == x.shape
        mean = x.mean((0, 2, 3), keepdim=True)
        var = x.var((0, 2, 3), unbiased=False, keepdim=True)
        invstd = (var + mod.eps).rsqrt()
        g = mod.weight.detach().double().abs().view(1, -1, 1, 1)
        b = mod.bias.detach().double().abs().view(1, -1, 1, 1)
        A = g * invstd * (self.last_abs + mean.abs()) + b
        self.pending.append((mod, out.detach().double(), A))
        self.last_abs = None

    def _relu(self, mod, inp):
        z = inp[0]
        if not self.pending:
            return None
        zd = z.detach().double()
        bn_sum = sum(o for _, o, _ in self.pending)
        A = sum(a for _, _, a in self.pending) + (zd - bn_sum).abs()
        last_bn = self.pending[-1][0]
        self.pending = []
        unit = EPS32 * A
        self.n_elems += zd.numel()
        if self.mode == "tune":
            band = SAFETY * (FACT if self.req is None else self.req[self.k]) * unit
            bad = (zd.abs() < band).flatten(2).any(2).any(0).nonzero().flatten().tolist()
            if bad:
                zf, bf = zd.transpose(0, 1).flatten(1).numpy(), band.transpose(0, 1).flatten(1).numpy()
                delta = torch.zeros(zd.shape[1], dtype=torch.float64)
                for c in bad:
                    delta[c] = self._clear(zf[c], bf[c])
                self.max_delta = max(self.max_delta, float(delta.abs().max()))
                self.moved[last_bn] = self.moved.get(last_bn, 0) + len(bad)
                with torch.no_grad():
                    last_bn.bias.add_(delta.to(last_bn.bias.dtype))
                    z = z + delta.to(z.dtype).view(1, -1, 1, 1)
                zd = z.detach().double()
            self.min_margin = min(self.min_margin, float((zd.abs() / unit).min()))
            self.k += 1
            return (z,)
        if self.mode == "record":
            self.margins.append(float((zd.abs() / unit).min()))
            self.min_margin = min(self.min_margin, self.margins[-1])
            self.records.append((zd.numpy().copy(), unit.numpy().copy()))
        else:
            z64, unit64 = self.recorded[self.k]
            self.mask_diff += int(((zd.numpy() > 0) != (z64 > 0)).sum())
            self.noise.append(float((np.abs(zd.numpy() - z64) / unit64).max()))
            self.noise_use = max(self.noise_use, self.noise[-1])
        self.k += 1
        return None

    @staticmethod
    def _clear(z, band):
        """smallest |delta| with |z_i + delta| > band_i for every sample of the channel"""
        q = 0.5 * float(np.median(band))
        for lo in range(0, 20000, 500):
            ks = np.arange(lo + 1, lo + 501, dtype=np.float64)
            cand = np.stack([ks * q, -ks * q], 1).reshape(-1)               # +q, -q, +2q, -2q, ...
            reach = np.abs(cand).max() + band.max()
            near = np.abs(z) < reach
            zz, bb = z[near], band[near]
            ok = (np.abs(zz[None, :] + cand[:, None]) > bb[None, :]).all(1)
            if ok.any():
                return float(cand[int(np.argmax(ok))])
        raise RuntimeError("no beta shift clears this channel")


def bn_bias_keys(sd):
    return [k for k in sd if k.endswith(".bias") and (k[:-4] + "running_mean") in sd]


def condition(model_ctor, shapes, seed, img, prep):
    """-> (state_dict with conditioned BatchNorm betas (fp32), sparse (index, value) of the moved ones, proof numbers)"""
    sd = H.synth_state_dict(shapes, seed=seed)
    keys = bn_bias_keys(sd)
    base = torch.cat([sd[k].flatten() for k in keys]).clone()
    m = model_ctor()

    def tune(req):
        rounds = 0
        while True:
            rounds += 1
            m.load_state_dict(sd)
            prep(m)
            m.double()
            with torch.no_grad():
                aud = ReluAudit(m, "tune", req=req)
                m(img.double())
                aud.close()
            moved = sum(aud.moved.values())
            print("  conditioning pass %d: %d ReLU-input elements, %d channels moved in %d BatchNorm layers, largest |delta beta| %.2e"
                  % (rounds, aud.n_elems, moved, len(aud.moved), aud.max_delta))
            new = m.state_dict()
            for k in keys:
                sd[k] = new[k].float().clone()
            if moved == 0:
                return
            assert rounds < 8

    def measure():
        """fp64 margins and fp32-vs-fp64 noise per ReLU on the current fp32 weights, nothing modified"""
        m.load_state_dict(sd)
        prep(m)
        m.double()
        with torch.no_grad():
            rec = ReluAudit(m, "record")
            m(img.double())
            rec.close()
        m.float()
        m.load_state_dict(sd)
        prep(m)
        with torch.no_grad():
            cmp_ = ReluAudit(m, "compare", rec.records)
            m(img)
            cmp_.close()
        assert cmp_.k == len(rec.records)
        return rec, cmp_

    tune(None)
    rec, cmp_ = measure()
    print("  after the local-bound pass: smallest margin %.0f, fp32-vs-fp64 noise per ReLU: median %.1f, max %.1f (x eps32 * sum|terms|)"
          % (rec.min_margin, float(np.median(cmp_.noise)), cmp_.noise_use))
    tune([max(FACT, NOISE_TUNE * n) for n in cmp_.noise])
    rec, cmp_ = measure()
    cur = torch.cat([sd[k].flatten() for k in keys])
    idx = (cur != base).nonzero().flatten()
    ratio = min(mg / max(nz, 1e-9) for mg, nz in zip(rec.margins, cmp_.noise))
    print("  PROOF: %d ReLU-input elements in %d ReLU calls; smallest |z| / (eps32 * sum|terms|) = %.1f (required >= %.0f); the "
          "reference's fp32 run: %d masks differ from fp64; its noise |z32 - z64| per ReLU: median %.1f max %.1f x eps32 * sum|terms|; "
          "smallest margin / noise over the ReLUs %.1f (required >= %.0f)"
          % (rec.n_elems, len(rec.records), rec.min_margin, FACT, cmp_.mask_diff, float(np.median(cmp_.noise)), cmp_.noise_use,
             ratio, NOISE_REQ))
    assert rec.min_margin >= FACT and cmp_.mask_diff == 0 and ratio >= NOISE_REQ
    proof = dict(relu_elems=rec.n_elems, relu_margin=rec.min_margin, relu_margins=np.array(rec.margins),
                 relu_fp32_noise=np.array(cmp_.noise), relu_margin_over_noise=ratio, relu_fact=FACT, relu_noise_req=NOISE_REQ,
                 beta_moved=int(idx.numel()), beta_max_delta=float((cur - base).abs().max()))
    return sd, (idx.to(torch.int32), cur[idx].clone()), proof


def checksums(grads):
    return np.stack([H.checksum(g) for g in grads.values()])


def main():
    torch.set_num_threads(8)
    MG.install_shims()
    sys.path.insert(0, os.path.join(MG.REF, "DeepLabV3Plus-Pytorch"))
    import network as R  # the reference package
    ref_loss = MG.load_by_path("ref_loss", os.path.join(MG.REF, "DeepLabV3Plus-Pytorch/utils/loss.py"))
    ref_sched = MG.load_by_path("ref_sched", os.path.join(MG.REF, "DeepLabV3Plus-Pytorch/utils/scheduler.py"))
    crit = ref_loss.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)

    def prep_single(m):
        m.train()
        m.classifier.aspp.project[3].eval()
        O.set_bn_momentum(m.backbone, 0.01)

    # ------------------------------------------------------------------ G5L
    print("G5L full model, one train step, 2x3x128x128, conditioned weights")
    ctor = lambda: R.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)  # noqa: E731
    ref = ctor()
    shapes = H.shapes_of(ref)
    img = H.synth_tensor(5, "g5l.img", (2, 3, 128, 128))
    lab = H.synth_labels(5, "g5l.lab", (2, 128, 128), 16, 255, ignore_rows=5)
    sd, (bidx, bval), proof = condition(ctor, shapes, 1, img, prep_single)
    assert H.conditioned_state_dict(shapes, 1, bidx.numpy(), bval.numpy()).keys() == sd.keys()
    chk = H.conditioned_state_dict(shapes, 1, bidx.numpy(), bval.numpy())
    assert all(torch.equal(chk[k], sd[k]) for k in sd), "helpers.conditioned_state_dict does not reproduce the conditioned weights"
    # fp64 run (exact arithmetic) for the noise figures, then the fp32 run that is the fixture
    ref.load_state_dict(sd)
    prep_single(ref)
    ref.double()
    lg64, _, ft64 = ref(img.double())
    crit(lg64, lab, ft64).backward()
    g64 = OrderedDict((k, p.grad.clone()) for k, p in ref.named_parameters())
    ref = ctor()
    ref.load_state_dict(sd)
    prep_single(ref)
    lg, ctr, ft = ref(img)
    loss = crit(lg, lab, ft)
    loss.backward()
    rg = OrderedDict((k, p.grad) for k, p in ref.named_parameters())
    lg64 = lg64.detach()
    ref_noise = float((lg.detach().double() - lg64).abs().max() / lg64.abs().max())
    gnoise = np.array([float((rg[k].double() - g64[k]).abs().max() / (g64[k].abs().max() + 1e-30)) for k in rg])
    cs32, cs64 = checksums(rg), checksums(g64)
    csn = np.abs(cs32[:, 1:] - cs64[:, 1:]) / np.abs(cs64[:, 1:])
    print("  reference fp32 vs fp64: logits %.2e; parameter gradients max-norm median %.2e max %.2e; checksums worst %.2e"
          % (ref_noise, np.median(gnoise), gnoise.max(), csn.max()))
    orc = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    orc.load_state_dict(sd)
    prep_single(orc)
    olg, octr, oft = orc(img)
    oloss = O.ce_over_n(olg, lab, 255)
    oloss.backward()
    MG.assert_close(olg, lg.detach(), 1e-4, "full logits")
    MG.assert_close(oft, ft.detach(), 1e-4, "full features_out")
    MG.assert_close(oloss, loss.detach(), 1e-5, "full loss")
    og = OrderedDict((k, p.grad) for k, p in orc.named_parameters())
    worst = max(H.max_abs(og[k], rg[k]) / (float(rg[k].abs().max()) + 1e-12) for k in rg)
    print("  oracle vs reference: worst relative param-grad error %.3e" % worst)
    assert worst < 2e-3
    keep = ["backbone.conv1.weight", "backbone.bn1.weight", "backbone.layer1.0.conv1.weight",
            "backbone.layer2.0.downsample.0.weight", "backbone.layer3.5.bn2.bias", "backbone.layer3.22.conv2.weight",
            "backbone.layer4.2.conv3.weight", "classifier.project.0.weight", "classifier.aspp.convs.2.0.weight",
            "classifier.aspp.project.1.weight", "classifier.classifier.0.weight", "classifier.classifier.3.weight",
            "classifier.classifier.3.bias"]
    extra = {"grad__" + k.replace(".", "_"): (rg[k] if rg[k].numel() < 70000 else rg[k].flatten()[::61]) for k in keep}
    rb = dict(ref.named_buffers())
    MG.save("g5l_full_train", beta_idx=bidx, beta_val=bval, logits_sub=lg.detach()[:, :, ::4, ::4],
            logits64_sub=lg64.float()[:, :, ::4, ::4], logits_checksum=H.checksum(lg), ref_noise=ref_noise, grad_noise=gnoise,
            loss=loss.detach(), grad_names=np.array(list(rg.keys())), grad_checksums=cs32, grad_keep=np.array(keep),
            rm_stem=rb["backbone.bn1.running_mean"], rv_stem=rb["backbone.bn1.running_var"],
            rv_l3=rb["backbone.layer3.11.bn2.running_var"], rv_l4=rb["backbone.layer4.2.bn3.running_var"],
            rv_head=rb["classifier.classifier.1.running_var"], **proof, **extra)

    # ------------------------------------------------------------------ G8L trajectory
    print("G8L six SGD / PolyLR steps, 2x3x128x128, from the conditioned weights")
    lr, total = 0.0002, 20

    def trajectory(dtype, threads):
        torch.set_num_threads(threads)
        m = ctor()
        m.load_state_dict(sd)
        prep_single(m)
        m.to(dtype)
        opt = torch.optim.SGD([{"params": m.backbone.parameters(), "lr": 0.1 * lr},
                               {"params": m.classifier.parameters(), "lr": lr}], lr=lr, momentum=0.9, weight_decay=1e-4)
        sched = ref_sched.PolyLR(opt, total, power=0.9)
        losses, lrs = [], []
        for it in range(6):
            opt.zero_grad()
            a, _, f = m(img.to(dtype))
            ls = crit(a, lab, f)
            ls.backward()
            opt.step()
            sched.step()
            losses.append(float(ls))
            lrs.append([g["lr"] for g in opt.param_groups])
        torch.set_num_threads(8)
        return np.array(losses), np.array(lrs), OrderedDict((k, v.detach().double().clone()) for k, v in m.state_dict().items())

    l8, lrs, fin8 = trajectory(torch.float32, 8)
    l1, _, fin1 = trajectory(torch.float32, 1)
    ld, _, find = trajectory(torch.float64, 8)
    dev = np.maximum(np.abs(l1 - l8) / np.abs(l8), np.abs(ld - l8) / np.abs(l8))
    bars = np.maximum(1e-5, 8.0 * np.maximum.accumulate(dev))
    print("  reference losses (fp32, 8 threads)", ["%.6f" % v for v in l8])
    print("  1 thread vs 8 threads             ", ["%.1e" % v for v in np.abs(l1 - l8) / np.abs(l8)])
    print("  fp64 vs fp32                      ", ["%.1e" % v for v in np.abs(ld - l8) / np.abs(l8)])
    print("  bars stored (8 x running max, floor 1e-5)", ["%.1e" % v for v in bars])
    oorc = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    oorc.load_state_dict(sd)
    prep_single(oorc)
    oopt = O.make_optimizer(oorc, lr=lr, weight_decay=1e-4)
    olosses = np.array([float(O.train_step(oorc, oopt, img, lab, it, total, [0.1 * lr, lr],
                                            lambda a, b: O.ce_over_n(a, b, 255))) for it in range(6)])
    print("  oracle losses                     ", ["%.6f" % v for v in olosses])
    assert (np.abs(olosses - l8) <= bars * np.abs(l8)).all()

    def wdev(k):
        sc = find[k].abs().max() + 1e-30
        return float(max((fin1[k] - fin8[k]).abs().max(), (find[k] - fin8[k]).abs().max()) / sc)
    wkeys = ["classifier.classifier.3.bias", "backbone.conv1.weight", "backbone.bn1.running_mean",
             "backbone.layer3.11.bn2.weight", "classifier.aspp.project.1.bias"]
    wbars = np.array([max(1e-5, 8.0 * wdev(k)) for k in wkeys])
    print("  final-tensor bars", dict(zip(wkeys, ["%.1e" % v for v in wbars])))
    MG.save("g8l_trajectory", beta_idx=bidx, beta_val=bval, losses=l8, losses_1thread=l1, losses_fp64=ld, bars=bars, lrs=lrs, lr=lr,
            total_itrs=total, wkeys=np.array(wkeys), wbars=wbars,
            **{"w_%d" % i: (fin8[k].float() if fin8[k].numel() < 70000 else fin8[k].float().flatten()[::61]) for i, k in enumerate(wkeys)})

    # ------------------------------------------------------------------ G12L two-head model
    print("G12L self-distillation model, one train step, 2x3x128x128, conditioned weights, loss on the last head")
    ctor2 = lambda: R.deeplabv3plus_embedding_self_distillation_resnet101(num_classes=16, output_stride=16,  # noqa: E731
                                                                          pretrained_backbone=False)

    def prep_multi(m):
        m.train()
        m.classifier.aspp.project[3].eval()
        m.classifier_1.aspp.project[3].eval()

    class Last(nn.Module):            # the conditioning / proof passes only need a forward through every ReLU
        def __init__(self):
            super().__init__()
            self.m = ctor2()

        def forward(self, x):
            return self.m(x)

        def state_dict(self, *a, **k):
            return self.m.state_dict(*a, **k)

        def load_state_dict(self, s, *a, **k):
            return self.m.load_state_dict(s, *a, **k)

    ref2 = ctor2()
    shapes2 = H.shapes_of(ref2)
    img2 = H.synth_tensor(12, "g12l.img", (2, 3, 128, 128))
    lab2 = H.synth_labels(12, "g12l.lab", (2, 128, 128), 17, 255, ignore_rows=5)
    sd2, (bidx2, bval2), proof2 = condition(Last, shapes2, 12, img2, lambda w: prep_multi(w.m))
    chk = H.conditioned_state_dict(shapes2, 12, bidx2.numpy(), bval2.numpy())
    assert all(torch.equal(chk[k], sd2[k]) for k in sd2)
    orc2 = O.deeplabv3plus_embedding_self_distillation_resnet101(output_stride=16)
    res = {}
    for name, m in (("ref", ref2), ("orc", orc2)):
        m.load_state_dict(sd2)
        prep_multi(m)
        logits, centers, feats = m(img2)
        ls = O.ce_over_n(logits[-1], lab2, 255)               # utils/loss.py:34-42 with alpha = 0, on the last head
        ls.backward()
        res[name] = (logits, centers, feats, ls, OrderedDict((k, p.grad) for k, p in m.named_parameters()))
    (lg2, ctr2, ft2, loss2, g2), (olg2, octr2, oft2, oloss2, og2) = res["ref"], res["orc"]
    for h in range(2):
        MG.assert_close(olg2[h], lg2[h].detach(), 1e-6, "logits head %d" % h)
        MG.assert_close(oft2[h], ft2[h].detach(), 1e-6, "features head %d" % h)
    assert abs(float(loss2) - float(oloss2)) < 1e-6
    none_ref = sorted(k for k, v in g2.items() if v is None)
    assert none_ref == sorted(k for k, v in og2.items() if v is None) and none_ref and all(k.startswith("classifier.") for k in none_ref)
    live = OrderedDict((k, v) for k, v in g2.items() if v is not None)
    keys2 = ["backbone.conv1.weight", "backbone.layer3.5.conv2.weight", "backbone.layer3.10.bn1.bias", "backbone.layer4.2.bn3.weight",
             "classifier_1.aspp.convs.1.0.weight", "classifier_1.classifier.0.weight", "classifier_1.classifier.3.weight",
             "classifier_1.classifier.3.bias"]
    save = dict(beta_idx=bidx2, beta_val=bval2, loss=float(loss2), n_keys=len(sd2), keys=np.array(list(sd2.keys())[-4:]),
                logits0_sub=lg2[0][:, :, ::4, ::4], logits1_sub=lg2[1][:, :, ::4, ::4], feats1_sub=ft2[1][:, ::4, ::4, :],
                logits0_checksum=H.checksum(lg2[0]), logits1_checksum=H.checksum(lg2[1]),
                grad_names=np.array(list(live.keys())), grad_checksums=checksums(live),
                grad_keys=np.array(keys2), untouched=np.array(none_ref[:3]), **proof2)
    for i, k in enumerate(keys2):
        save["grad_%d" % i] = g2[k] if g2[k].numel() <= 70000 else g2[k].reshape(-1)[::97]
    MG.save("g12l_multihead", **save)
    print("large fixtures minted; every ReLU input of each has the stated margin")


if __name__ == "__main__":
    main()
