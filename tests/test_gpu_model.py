"""GPU: the drop-in `network` / `utils` API on the HIP path against the golden vectors minted from the real
reference (G5, G5b, G8) and against the CPU oracle on fresh seeded inputs.  Tolerance from BASELINE.json:
1e-3 (relative to the tensor's max magnitude) in fp32 mode.
"""
from collections import OrderedDict

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
T = torch.from_numpy
TOL = 1e-3


def relclose(got, ref, tol, what=""):
    err = (got.detach().double().cpu() - ref.detach().double().cpu()).abs().max().item()
    scale = ref.detach().double().abs().max().item() + 1e-12
    assert err <= tol * scale, "%s: max|d|=%.3e scale=%.3e rel=%.3e > %.1e" % (what, err, scale, err / scale, tol)


def build(dtype=torch.float32, train=True, seed=1, num_classes=16, output_stride=16, fp32_products=None, state=None):
    import network
    import utils
    m = network.deeplabv3plus_embedding_resnet101(num_classes=num_classes, output_stride=output_stride,
                                                  pretrained_backbone=False)
    m.load_state_dict(state(H.shapes_of(m)) if state is not None else H.synth_state_dict(H.shapes_of(m), seed=seed))
    m.cuda()
    m.set_compute_dtype(dtype, fp32_products=fp32_products)
    if train:
        m.train()
        m.classifier.aspp.project[3].eval()          # F14: dropout off for parity
        utils.set_bn_momentum(m.backbone, 0.01)
    else:
        m.eval()
    return m


def g5_inputs():
    img = H.synth_tensor(5, "g5.img", (2, 3, 64, 64)).cuda()
    lab = H.synth_labels(5, "g5.lab", (2, 64, 64), 16, 255, ignore_rows=3).cuda()
    return img, lab


def g8_inputs():
    img = H.synth_tensor(5, "g8.img", (2, 3, 64, 64)).cuda()
    lab = H.synth_labels(5, "g8.lab", (2, 64, 64), 16, 255, ignore_rows=3).cuda()
    return img, lab


def g5l_inputs():
    img = H.synth_tensor(5, "g5l.img", (2, 3, 128, 128)).cuda()
    lab = H.synth_labels(5, "g5l.lab", (2, 128, 128), 16, 255, ignore_rows=5).cuda()
    return img, lab


def conditioned(fixture, seed):
    g = H.load_golden(fixture)
    return lambda shapes: H.conditioned_state_dict(shapes, seed, g["beta_idx"], g["beta_val"])


@pytest.mark.parametrize("products", ["exact", "bf16x3", "f16x2"])
def test_g5l_full_train_step_matches_reference(products):
    """THE full-model train-step gate of every fp32 arithmetic mode -- exact fp32 MFMA, three-term bf16 split, two-plane fp16 split (the
    bench headline) -- one set of bars, no mode-dependent branch: the reference-minted 2 x 3 x 128 x 128 fixture whose mint script
    proves that no ReLU input of the network lies within 64 x eps32 x sum|terms| (nor within 6 x the reference's own fp32-vs-fp64
    noise) of zero, so no summation order can flip a mask (tests/tools/mint_golden_large.py).  Logits, loss, running statistics at
    1e-3; all 338 parameter-gradient checksums and the sampled gradients at 1e-3 as well (the 64 x 64 fixture needed 2e-3;
    measured here: 5e-5 .. 8e-5 in every mode, the reference's own fp32-vs-fp64 checksum noise is 3e-5)."""
    import utils
    g = H.load_golden("g5l_full_train")
    m = build(fp32_products=products, state=conditioned("g5l_full_train", 1))
    img, lab = g5l_inputs()
    lg, ctr, ft = m(img)
    assert lg.shape == (2, 16, 128, 128) and ft.shape == (2, 128, 128, 16) and ctr.shape == (16, 16)
    loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(lg, lab, ft)
    loss.backward()
    relclose(lg[:, :, ::4, ::4], T(g["logits_sub"]), TOL, "logits vs fp32 reference")
    relclose(lg[:, :, ::4, ::4], T(g["logits64_sub"]), TOL, "logits vs fp64 reference")
    assert np.allclose(H.checksum(lg), g["logits_checksum"], rtol=TOL)
    assert abs(loss.item() - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    assert [str(n) for n in g["grad_names"]] == list(grads.keys())
    rels = np.array([(np.abs(H.checksum(gr)[1:] - cs[1:]) / np.abs(cs[1:])).max() for gr, cs in zip(grads.values(), g["grad_checksums"])])
    worst = list(grads.keys())[int(rels.argmax())]
    print("g5l %s: gradient checksums worst %.2e (%s) median %.2e; the reference's own fp32 vs fp64: %.2e"
          % (products, rels.max(), worst, np.median(rels), float(np.median(g["grad_noise"]))))
    assert rels.max() <= TOL, "gradient checksums: %d of %d tensors beyond 1e-3, worst %s %.3e" % ((rels > TOL).sum(), len(rels), worst, rels.max())
    for key in [str(k) for k in g["grad_keep"]]:
        ref = T(g["grad__" + key.replace(".", "_")])
        got = grads[key].detach().cpu()
        got = got if got.numel() < 70000 else got.contiguous().flatten()[::61]
        relclose(got.reshape(ref.shape), ref, TOL, "grad " + key)
    bufs = dict(m.named_buffers())
    relclose(bufs["backbone.bn1.running_mean"], T(g["rm_stem"]), TOL, "running_mean stem")
    relclose(bufs["backbone.bn1.running_var"], T(g["rv_stem"]), TOL, "running_var stem")
    relclose(bufs["backbone.layer3.11.bn2.running_var"], T(g["rv_l3"]), TOL, "running_var layer3")
    relclose(bufs["backbone.layer4.2.bn3.running_var"], T(g["rv_l4"]), TOL, "running_var layer4")
    relclose(bufs["classifier.classifier.1.running_var"], T(g["rv_head"]), TOL, "running_var head")
    assert int(bufs["backbone.bn1.num_batches_tracked"]) == 1


def test_g5_full_train_step_matches_reference():
    """The round-1 fixture (2 x 3 x 64 x 64, reference-minted, weights NOT conditioned): kept for the exact-fp32 mode, whose
    summation order it pins.  Its layer3 / layer4 / ASPP BatchNorms see 32 samples and single pre-ReLU values lie within rounding
    of zero, so it is no gate for a mode with another rounding pattern -- those are held to the conditioned 128 x 128 fixture
    above at the same bars."""
    import utils
    g = H.load_golden("g5_full_train")
    m = build(fp32_products="exact")
    img, lab = g5_inputs()
    lg, ctr, ft = m(img)
    assert lg.shape == (2, 16, 64, 64) and ft.shape == (2, 64, 64, 16) and ctr.shape == (16, 16)
    loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(lg, lab, ft)
    loss.backward()
    # On these (well-conditioned, helpers.synth_state_dict) weights the reference's own fp32 run sits 3.6e-6
    # (logits) / 1.5e-4 (worst parameter gradient) away from its fp64 evaluation, so 1e-3 is a real bar.
    relclose(lg, T(g["logits"]), TOL, "logits vs fp32 reference")
    relclose(lg, T(g["logits64"]), TOL, "logits vs fp64 reference")
    assert abs(loss.item() - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    assert names == list(grads.keys())
    bad = []
    for (k, gr), cs in zip(grads.items(), g["grad_checksums"]):
        got = H.checksum(gr)
        if not np.allclose(got[1:], cs[1:], rtol=2e-3):
            bad.append((k, got, cs))
    assert not bad, "gradient checksums differ for %d tensors, first: %r" % (len(bad), bad[:3])
    for key in ("backbone.bn1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer2.0.downsample.0.weight",
                "backbone.layer3.5.bn2.bias", "classifier.project.0.weight", "classifier.aspp.project.1.weight",
                "classifier.classifier.3.weight", "classifier.classifier.3.bias", "backbone.conv1.weight"):
        ref = T(g["grad__" + key.replace(".", "_")])
        got = grads[key].detach().cpu()
        got = got if got.numel() < 70000 else got.contiguous().flatten()[::16]
        relclose(got.reshape(ref.shape), ref, 2 * TOL, "grad " + key)
    bufs = dict(m.named_buffers())
    relclose(bufs["backbone.bn1.running_mean"], T(g["rm_stem"]), TOL, "running_mean stem")
    relclose(bufs["backbone.bn1.running_var"], T(g["rv_stem"]), TOL, "running_var stem")
    relclose(bufs["backbone.layer4.2.bn3.running_var"], T(g["rv_l4"]), TOL, "running_var layer4")
    relclose(bufs["classifier.classifier.1.running_var"], T(g["rv_head"]), TOL, "running_var head")
    assert int(bufs["backbone.bn1.num_batches_tracked"]) == 1


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16], ids=["f32", "bf16"])
def test_train_steps_are_bitwise_reproducible(dtype):
    """Three SGD steps twice from the same state: every parameter, buffer and loss must come out bit for bit the same -- all
    reductions of the step (weight-gradient slabs, K-split tails, BN partials, bias gradient, loss) are fixed-order sums, no
    floating-point atomics.  (The 30-step equivalence gate and the trajectory fixtures rely on it: the dynamics amplify a
    last-bit difference to 1e-2 within 20 steps.)"""
    import utils
    from dmlnet.optim import FusedSGD
    img = H.synth_tensor(11, "rep.img", (4, 3, 96, 128)).cuda()
    lab = H.synth_labels(11, "rep.lab", (4, 96, 128), 16, 255, ignore_rows=3).cuda()
    runs = []
    for rep in range(2):
        m = build(dtype=dtype, seed=3)
        opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.01}, {"params": m.classifier.parameters(), "lr": 0.1}],
                       lr=0.1, momentum=0.9, weight_decay=1e-4).bind(m)
        crit = utils.CrossEntropyLoss(ignore_index=255)
        losses = []
        for it in range(3):
            opt.zero_grad()
            lg, _, ft = m(img)
            loss = crit(lg, lab, ft)
            loss.backward()
            opt.step()
            losses.append(loss.item())
        torch.cuda.synchronize()
        runs.append((losses, {k: v.detach().clone() for k, v in m.state_dict().items()}))
    assert runs[0][0] == runs[1][0], (runs[0][0], runs[1][0])
    diff = [k for k, v in runs[0][1].items() if not torch.equal(v, runs[1][1][k])]
    assert not diff, "%d of %d tensors differ between two identical runs, e.g. %s" % (len(diff), len(runs[0][1]), diff[:5])


@pytest.mark.parametrize("products", ["exact", "bf16x3", "f16x2"])
def test_g8l_sgd_polylr_trajectory(products):
    """Six SGD (two LR groups, momentum, weight decay) + PolyLR steps from the conditioned weights at 2 x 3 x 128 x 128 against the
    reference's trajectory.  The bars are the FIXTURE's, one rule for every mode: at step t, 8 x the largest deviation (steps <= t)
    between the reference's own fp32 / 8 threads, fp32 / 1 thread and fp64 runs, floor 1e-5 (tests/tools/mint_golden_large.py) --
    the weights of steps 1.. are the optimizer's and cannot be conditioned, so the reference's own run-to-run spread is the yardstick."""
    import utils
    from dmlnet.optim import FusedSGD
    t = H.load_golden("g8l_trajectory")
    m = build(fp32_products=products, state=conditioned("g8l_trajectory", 1))
    img, lab = g5l_inputs()
    lr, total = float(t["lr"]), int(t["total_itrs"])
    opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.1 * lr},
                    {"params": m.classifier.parameters(), "lr": lr}], lr=lr, momentum=0.9, weight_decay=1e-4).bind(m)
    sched = utils.PolyLR(opt, total, power=0.9)
    crit = utils.CrossEntropyLoss(ignore_index=255)
    losses = []
    for it in range(6):
        opt.zero_grad()
        lg, _, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
    rel = [abs(a - b) / abs(b) for a, b in zip(losses, t["losses"])]
    print("g8l %s: relative loss differences" % products, ["%.1e" % v for v in rel], "bars", ["%.1e" % v for v in t["bars"]])
    for it, (r, bar) in enumerate(zip(rel, t["bars"])):
        assert r <= bar, "step %d: %.7f vs %.7f (rel %.2e > %.1e)" % (it, losses[it], t["losses"][it], r, bar)
    assert np.allclose([g_["lr"] for g_ in opt.param_groups], t["lrs"][-1], rtol=1e-6)
    sd = m.state_dict()
    for i, (k, bar) in enumerate(zip([str(k) for k in t["wkeys"]], t["wbars"])):
        got = sd[k].float().cpu()
        got = got if got.numel() < 70000 else got.contiguous().flatten()[::61]
        ref = T(t["w_%d" % i])
        relclose(got.reshape(ref.shape), ref, float(bar), "%s after 6 steps" % k)


def test_g8_sgd_polylr_trajectory():
    """the round-1 trajectory fixture (2 x 3 x 64 x 64, unconditioned weights): exact-fp32 mode only, see test_g5_full_train_step_matches_reference"""
    import utils
    from dmlnet.optim import FusedSGD
    t = H.load_golden("g8_trajectory")
    m = build(fp32_products="exact")
    img, lab = g8_inputs()
    lr, total = float(t["lr"]), int(t["total_itrs"])
    opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.1 * lr},
                    {"params": m.classifier.parameters(), "lr": lr}], lr=lr, momentum=0.9, weight_decay=1e-4).bind(m)
    sched = utils.PolyLR(opt, total, power=0.9)
    crit = utils.CrossEntropyLoss(ignore_index=255)
    losses = []
    for it in range(6):
        opt.zero_grad()
        lg, _, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        opt.step()
        sched.step()
        losses.append(loss.item())
    # the reference's own trajectory drifts with the CPU thread count (1 vs 8 threads: 1e-5 at step 1, 1.3e-3 at
    # step 5: the dynamics of this 32-sample fixture are chaotic), so the bar widens with the step index
    bars = (1e-5, 1e-4, 2e-4, 1e-3, 3e-3, 1e-2)
    print("g8: relative loss differences", ["%.1e" % (abs(a - b) / abs(b)) for a, b in zip(losses, t["losses"])])
    for it, (a, b, tol) in enumerate(zip(losses, t["losses"], bars)):
        assert abs(a - b) <= tol * abs(b), "step %d: %.6f vs %.6f" % (it, a, b)
    assert np.allclose([g_["lr"] for g_ in opt.param_groups], t["lrs"][-1], rtol=1e-6)
    sd = m.state_dict()
    relclose(sd["classifier.classifier.3.bias"], T(t["b_last"]), 5e-3, "final bias after 6 steps")
    relclose(sd["backbone.conv1.weight"], T(t["w_stem"]), 2e-3, "stem weight after 6 steps")
    relclose(sd["backbone.bn1.running_mean"], T(t["rm_stem"]), 2e-3, "stem running mean after 6 steps")


@pytest.mark.parametrize("products", ["exact", "f16x2"])
def test_g5b_eval_forward_config1(products):
    """BASELINE config #1 shape (1x3x256x256 eval forward) on the HIP path; also in the two-plane mode, whose eval plan splits every
    activation with dml_h2_split (running-statistics BatchNorm has no bound) and runs BN + residual + ReLU in the conv epilogues."""
    g = H.load_golden("g5b_full_eval")
    m = build(train=False, fp32_products=products)
    flat, off = T(g["bn_stats"]), 0
    sd = m.state_dict()
    for k, v in sd.items():
        if k.endswith("running_mean") or k.endswith("running_var"):
            sd[k] = flat[off:off + v.numel()].view_as(v).clone()
            off += v.numel()
    m.load_state_dict(sd)
    with torch.no_grad():
        lg, ctr, ft = m(H.synth_tensor(5, "g5b.img", (1, 3, 256, 256)).cuda())
    relclose(lg[:, :, ::4, ::4], T(g["logits_sub"]), TOL, "eval logits")
    relclose(ft[:, ::4, ::4, :], T(g["feats_sub"]), TOL, "eval features")
    assert np.allclose(H.checksum(lg), g["logits_checksum"], rtol=2e-3)
    agree = (lg.argmax(1).to(torch.uint8).cpu() == T(g["argmax"])).float().mean().item()
    assert agree > 0.995, agree
    assert torch.equal(ctr.cpu(), T(g["centers"]))


def test_forked_eval_plan_f16x2_is_bitwise_the_unforked_one(monkeypatch):
    """Inference plans run the five ASPP branches on their own streams (Plan.run_forward).  In the two-plane mode the branches share
    the fp16 planes of `out`, whose split must have been issued on the main stream BEFORE the fork point -- issued by the first
    branch it raced with the others (they read the previous image's planes, or uninitialised memory on the first call).  Every
    activation plane of the plan is poisoned before the checked forward; forked and unforked results must be bit-identical."""
    img1 = H.synth_tensor(71, "fork.img1", (1, 3, 256, 512)).cuda()
    img2 = H.synth_tensor(71, "fork.img2", (1, 3, 256, 512)).cuda()
    outs = {}
    for fork in ("0", "1"):
        monkeypatch.setenv("DML_FORK_BRANCHES", fork)
        m = build(train=False, fp32_products="f16x2")
        with torch.no_grad():
            m(img1)
            plan = next(iter(m._engine.plans.values()))
            assert bool(getattr(plan, "fwd_forks", None)) == (fork == "1")
            weight_planes = {e[4] for e in plan.prep_h2}
            n = 0
            for t in plan.keep:
                if isinstance(t, torch.Tensor) and t.dtype == torch.float16 and t.data_ptr() not in weight_planes:
                    t.fill_(float("nan"))
                    n += 1
            assert n > 50
            lg, _, ft = m(img2)
        torch.cuda.synchronize()
        assert torch.isfinite(lg).all()
        outs[fork] = (lg.clone(), ft.clone())
    assert torch.equal(outs["0"][0], outs["1"][0]) and torch.equal(outs["0"][1], outs["1"][1])


@pytest.mark.parametrize("shape", [(1, 3, 65, 97), (3, 3, 50, 34)])
def test_odd_input_sizes_eval_and_train_forward(shape):
    """Sizes that are no multiple of the output stride (ragged tiles everywhere, odd bilinear ratios): eval forward with
    calibrated running statistics, and the train-mode forward, against the fp64 oracle."""
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    img = H.synth_tensor(31, "odd.img", shape)
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=31))
    o = o.double()
    o.train()
    o.classifier.aspp.project[3].eval()
    if shape[0] > 1:
        m = build(seed=31)
        with torch.no_grad():
            lg, _, ft = m(img.cuda())
            olg, _, oft = o(img.double())
        relclose(lg, olg, TOL, "train-mode logits, odd size")
        relclose(ft, oft, TOL, "train-mode features, odd size")
    # calibrate running statistics with a few train-mode passes of the oracle, then compare eval mode
    cal = H.synth_tensor(32, "odd.cal", (4, 3, shape[2], shape[3]))
    with torch.no_grad():
        for _ in range(3):
            o(cal.double())
    o.eval()
    m = build(train=False, seed=31)
    m.load_state_dict({k: v.float() for k, v in o.state_dict().items()})
    with torch.no_grad():
        lg, _, ft = m(img.cuda())
        olg, _, oft = o(img.double())
    relclose(lg, olg, TOL, "eval logits, odd size")
    relclose(ft, oft, TOL, "eval features, odd size")


def test_against_oracle_fresh_input_all_param_grads():
    """Non-square input, DML loss with the variance term: every parameter gradient.

    At this size even the well-conditioned weights leave a few parameter gradients of the fp32 ORACLE several
    % away from its fp64 evaluation (ReLU sign flips), so the bar is relative to that: the HIP path must be as
    close to the exact (fp64) gradients as the reference's fp32 arithmetic is (x3) or within 1e-3."""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    m = build(seed=9)
    img = H.synth_tensor(9, "fresh.img", (3, 3, 96, 128))
    lab = H.synth_labels(9, "fresh.lab", (3, 96, 128), 16, 255, ignore_frac=0.05)
    _check_against_oracles(m, img, lab, strict=False)


@pytest.mark.parametrize("products", ["exact", "bf16x3", "f16x2"])
def test_against_oracle_nonsquare_strict(products):
    """Same check on a NON-SQUARE 2 x 3 x 128 x 192 input with CONDITIONED weights (tests/tools/mint_golden_nonsquare.py: the betas
    moved so that no ReLU input of the network lies within 64 * eps32 * sum|terms| -- and 6 x the reference's own fp32-vs-fp64 noise --
    of zero, proved on the reference in fp64): absolute bars against the fp64 gradients of the oracle run at test time, one set for
    all three fp32 arithmetic modes.  (Until round 5 this ran 2 x 3 x 64 x 96 with unconditioned weights in the exact mode only, and
    its bar on the worst gradient, 5e-2, had been set around ONE ReLU sign flip in a 48-sample BatchNorm; which element flips moved
    with any change of a convolution's summation order.)"""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    state = conditioned("g13n_nonsquare", 11)
    m = build(seed=11, fp32_products=products, state=state)
    img = H.synth_tensor(11, "g13n.img", (2, 3, 128, 192))
    lab = H.synth_labels(11, "g13n.lab", (2, 128, 192), 16, 255, ignore_frac=0.05)
    _check_against_oracles(m, img, lab, strict=True, seed=11, state=state)


@pytest.mark.parametrize("num_classes,output_stride", [(16, 8), (8, 16), (24, 16), (32, 8), (21, 16), (13, 16)])
def test_variants_against_oracle(num_classes, output_stride):
    """The factory's other configurations (network/modeling.py:140-148: output_stride 8 = dilations 2/4 in layer3/4
    and ASPP rates 12/24/36; embedding widths other than 16, incl. the factory default 21 and 13 = StreetHazards, which
    are carried padded to a multiple of 8 internally), train step vs the fp32 / fp64 oracle -- on CONDITIONED weights
    (tests/tools/mint_golden_variants.py: one record per output stride, proved on the reference in fp64; the embedding width only
    changes the last 1x1 convolution, which no ReLU follows).  Until round 5 this ran on unconditioned weights and its bars -- 5e-2
    on the worst gradient, 5e-3 on the median -- were set around ReLU sign flips of 40-sample BatchNorms."""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    state = conditioned("g14v_os%d" % output_stride, 21)
    m = build(seed=21, num_classes=num_classes, output_stride=output_stride, state=state)
    img = H.synth_tensor(21, "var.img", (2, 3, 64, 80))
    lab = H.synth_labels(21, "var.lab", (2, 64, 80), num_classes, 255, ignore_frac=0.05)
    _check_against_oracles(m, img, lab, strict=True, seed=21, num_classes=num_classes, output_stride=output_stride, state=state)


def test_unsupported_embedding_width_raises():
    import network
    m = network.deeplabv3plus_embedding_resnet101(num_classes=40, output_stride=16, pretrained_backbone=False).cuda()
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    with pytest.raises(NotImplementedError):
        m(torch.randn(2, 3, 64, 64, device="cuda"))


def test_default_factory_arguments_bf16_features_have_num_classes_channels():
    """network.deeplabv3plus_embedding_resnet101() with the reference's defaults (21 classes, output stride 8)."""
    import network
    import utils
    m = network.deeplabv3plus_embedding_resnet101(pretrained_backbone=False).cuda()
    m.set_compute_dtype(torch.bfloat16)
    m.train()
    x = torch.randn(2, 3, 64, 64, device="cuda")
    lab = torch.randint(0, 21, (2, 64, 64), device="cuda")
    lg, ctr, ft = m(x)
    assert lg.shape == (2, 21, 64, 64) and ft.shape == (2, 64, 64, 21) and ctr.shape == (21, 21) and ft.is_contiguous()
    # logits are exactly the distances of the returned features to the returned centers (utils.py:103-118)
    d = -((ft.unsqueeze(3) - ctr.to(ft.device)) ** 2).sum(-1).permute(0, 3, 1, 2)
    relclose(lg, d, 1e-5, "logits vs features / centers")
    (utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab, ft) + 1e-3 * ft.sum()).backward()      # gfeats path with padding
    g = m.classifier.classifier[3].weight.grad
    assert g.shape == (21, 256, 1, 1) and torch.isfinite(g).all() and g.abs().max() > 0


def _check_against_oracles(m, img, lab, strict, seed=9, num_classes=16, output_stride=16, prep=None, state=None):
    import utils
    from oracle import dmlnet_ref as O
    if prep is not None:
        prep(m)
    lg, _, ft = m(img.cuda())
    loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
    loss.backward()
    ref = {}
    for name, dt in (("f32", torch.float32), ("f64", torch.float64)):
        o = O.deeplabv3plus_embedding_resnet101(num_classes=num_classes, output_stride=output_stride)
        o.load_state_dict(state(H.shapes_of(o)) if state is not None else H.synth_state_dict(H.shapes_of(o), seed=seed))
        o = o.to(dt)
        o.train()
        o.classifier.aspp.project[3].eval()
        if prep is not None:
            prep(o)
        olg, _, oft = o(img.to(dt))
        oloss = O.dml_loss(olg, lab, alpha=0.01, ignore_index=255)
        oloss.backward()
        ref[name] = (olg.detach(), oft.detach(), float(oloss), {k: p.grad.double() for k, p in o.named_parameters()})
    relclose(lg, ref["f64"][0], TOL, "logits vs fp64 oracle")
    relclose(lg, ref["f32"][0], TOL, "logits vs fp32 oracle")
    relclose(ft, ref["f32"][1], TOL, "features vs oracle")
    assert abs(loss.item() - ref["f64"][2]) <= TOL * abs(ref["f64"][2])
    e_hip, e_ref = [], []
    for k, p in m.named_parameters():
        g64 = ref["f64"][3][k]
        sc = g64.abs().max().item() + 1e-30
        e_hip.append((p.grad.detach().cpu().double() - g64).abs().max().item() / sc)
        e_ref.append((ref["f32"][3][k] - g64).abs().max().item() / sc)
    e_hip, e_ref = np.array(e_hip), np.array(e_ref)
    print("grad error vs fp64 (max-norm): hip median %.2e p95 %.2e max %.2e | oracle fp32 median %.2e p95 %.2e "
          "max %.2e" % (np.median(e_hip), np.percentile(e_hip, 95), e_hip.max(), np.median(e_ref),
                        np.percentile(e_ref, 95), e_ref.max()))
    if strict is None:
        # Wiring check for the other factory configurations.  On this input ONE ReLU of the decoder sits within fp32
        # rounding of zero (tests/tools/debug_head.py: dz agrees with the fp64 oracle to 7e-5, dy differs at single elements
        # by the full dz value), which shifts every upstream gradient by ~2e-3 -- in the K = 16 / OS 16 configuration
        # just the same.  A mis-wired dilation, stride or channel count gives O(1) errors, so: logits / loss at 1e-3
        # (above), every gradient within 5e-2 of the fp64 one in max-norm and the median within 5e-3 -- or within 3x of
        # what the fp32 oracle itself manages where that is worse (output stride 8 with 32 classes: oracle 6.8e-2 / 6.5e-3).
        assert e_hip.max() <= max(5e-2, 3 * e_ref.max()) and np.median(e_hip) <= max(5e-3, 3 * np.median(e_ref))
        return
    if strict:
        # absolute bars on the conditioned cases -- every parameter gradient, max-norm against fp64 (measured on the non-square input,
        # exact / bf16x3 / f16x2: median 1.5e-4 / 1.1e-4 / 0.8e-4, worst tensor 2.8e-4 / 1.8e-4 / 1.5e-4; the six factory variants, exact:
        # median 0.8-1.1e-4, worst tensor 1.6-4.1e-4 -- 4.8e-4 with the stem regrouped; the fp32 oracle itself: 0.5-0.7e-4 / 1.0-1.4e-4)
        assert np.median(e_hip) <= 0.5 * TOL and np.percentile(e_hip, 95) <= TOL and e_hip.max() <= 2 * TOL
        return
    # individual tensors hit rare sign flips (either implementation can), the distribution must match
    assert np.median(e_hip) <= 3 * np.median(e_ref) + TOL
    assert np.percentile(e_hip, 95) <= 3 * np.percentile(e_ref, 95) + TOL
    assert e_hip.max() <= 10 * e_ref.max() + TOL


def _freeze_bn(which):
    def prep(model):
        for name, mod in model.named_modules():
            if isinstance(mod, torch.nn.BatchNorm2d) and (which == "all" or name.startswith(which)):
                mod.eval()
    return prep


@pytest.mark.parametrize("products", ["exact", "f16x2"])
@pytest.mark.parametrize("which", ["all", "backbone."])
def test_train_step_with_fixed_batchnorm_statistics(which, products):
    """model.train() with BatchNorm2d modules in eval() (main_self_distillation.py:432-435 of the reference freezes all
    of them; freezing only the backbone is the other common recipe): running statistics normalise and stay untouched,
    gamma / beta and everything upstream still get their gradients."""
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    # (f16x2: a BatchNorm with fixed statistics has no bound on its output -- those tensors go through dml_h2_split with the maximum
    # the apply kernel published; its backward is the batch-statistics one without correction terms, dy's planes scaled from max |g|)
    m = build(seed=33, fp32_products=products)
    before = {k: v.clone() for k, v in m.state_dict().items() if "running_" in k or "num_batches" in k}
    img = H.synth_tensor(33, "fix.img", (2, 3, 64, 80))
    lab = H.synth_labels(33, "fix.lab", (2, 64, 80), 16, 255, ignore_frac=0.05)
    _check_against_oracles(m, img, lab, strict=None, seed=33, prep=_freeze_bn(which))
    after = m.state_dict()
    changed = [k for k in before if not torch.equal(before[k], after[k])]
    if which == "all":
        assert not changed, changed[:5]
    else:
        assert changed and all(k.startswith("classifier.") for k in changed)
        assert int(after["classifier.project.1.num_batches_tracked"]) == int(before["classifier.project.1.num_batches_tracked"]) + 1
        assert int(after["backbone.bn1.num_batches_tracked"]) == int(before["backbone.bn1.num_batches_tracked"])
    g = m.backbone.layer3[5].bn2.weight.grad
    assert g is not None and torch.isfinite(g).all() and g.abs().max() > 0
    if which == "all" and products == "exact":
        # no batch statistics anywhere: a single image is a legal training batch now (the reference's B >= 2 rule comes
        # from the pooled ASPP branch's BatchNorm, network/utils.py:318-329)
        lg1, _, _ = m(img[:1].cuda())
        assert lg1.shape == (1, 16, 64, 80) and torch.isfinite(lg1).all()
        # ... and the bf16 plan (fused backward sums in the data gradients) tracks the fp32 one
        import utils
        mb = build(dtype=torch.bfloat16, seed=33)
        _freeze_bn("all")(mb)
        lgb, _, ftb = mb(img.cuda())
        utils.DMLLoss(alpha=0.01, ignore_index=255)(lgb, lab.cuda(), ftb).backward()
        mf = build(seed=33)
        _freeze_bn("all")(mf)
        lg, _, ft = mf(img.cuda())
        utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft).backward()
        relclose(lgb, lg, 0.15, "bf16 logits vs fp32 logits, fixed statistics")
        for mod_b, mod_f in ((mb.classifier.classifier[3], mf.classifier.classifier[3]),
                             (mb.backbone.layer4[2].bn3, mf.backbone.layer4[2].bn3)):
            gb, gf = mod_b.weight.grad.flatten(), mod_f.weight.grad.flatten()
            assert torch.isfinite(gb).all() and torch.nn.functional.cosine_similarity(gb, gf, dim=0).item() > 0.95


def test_features_out_carries_grad_and_eval_no_grad():
    m = build()
    img, lab = g8_inputs()
    lg, _, ft = m(img)
    assert lg.requires_grad and ft.requires_grad
    (ft.sum() * 1e-3 + lg.mean()).backward()
    assert m.backbone.conv1.weight.grad is not None and torch.isfinite(m.backbone.conv1.weight.grad).all()
    m.eval()
    with torch.no_grad():
        lg2, _, _ = m(img)
    assert not lg2.requires_grad and torch.isfinite(lg2).all()
    with pytest.raises(ValueError):
        m.train()
        m(img[:1])                                   # F12: same failure as the reference at batch 1


def test_bf16_mode_tracks_fp32():
    import utils
    img, lab = g8_inputs()
    ref, _, _ = build()(img)
    m = build(dtype=torch.bfloat16)
    lg, _, ft = m(img)
    loss = utils.CrossEntropyLoss(ignore_index=255)(lg, lab, ft)
    loss.backward()
    assert torch.isfinite(lg).all()
    relclose(lg, ref, 0.15, "bf16 logits vs fp32 logits")
    m32 = build()
    lg32, _, ft32 = m32(img)
    loss32 = utils.CrossEntropyLoss(ignore_index=255)(lg32, lab, ft32)
    loss32.backward()
    assert abs(loss.item() - loss32.item()) <= 0.03 * loss32.item()
    g, g32 = m.classifier.classifier[3].weight.grad.flatten(), m32.classifier.classifier[3].weight.grad.flatten()
    assert torch.isfinite(g).all()
    assert torch.nn.functional.cosine_similarity(g, g32, dim=0).item() > 0.97


def test_state_dict_roundtrip_into_oracle():
    from oracle import dmlnet_ref as O
    m = build(train=False)
    sd = m.state_dict()
    assert len(sd) == 674 and all(v.is_contiguous() for v in sd.values())
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.load_state_dict({k: v.cpu() for k, v in sd.items()})
    ref = H.synth_state_dict(H.shapes_of(o), seed=1)
    for k, v in o.state_dict().items():
        assert torch.equal(v, ref[k]), k


@pytest.mark.parametrize("dtype,products", [(torch.bfloat16, None), (torch.float32, "exact"), (torch.float32, "f16x2")])
def test_baseline_size_properties(dtype, products):
    """768x768 bs=16 (BASELINE configs[2]) train steps: size-independent properties of the head and loss, in bf16, exact fp32 and
    the bench headline's arithmetic (fp32 tensors, products of two fp16 planes per operand)."""
    import gc
    import utils
    from dmlnet.optim import FusedSGD
    gc.collect()                                      # the previous case's plan (tens of GB at this size) hangs in reference cycles
    torch.cuda.empty_cache()
    m = build(dtype=dtype, fp32_products=products)
    m.classifier.aspp.project[3].train()             # dropout on, as in the real step
    g = torch.Generator(device="cpu").manual_seed(1234)
    img = torch.randn(16, 3, 768, 768, generator=g).cuda()
    lab = torch.randint(0, 16, (16, 768, 768), generator=g)
    lab[:, :38] = 255
    lab = lab.cuda()
    opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.001},
                    {"params": m.classifier.parameters(), "lr": 0.01}], lr=0.01, momentum=0.9, weight_decay=1e-4).bind(m)
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
    losses = []
    for it in range(3):
        opt.zero_grad()
        lg, ctr, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        opt.step()
        losses.append(loss.item())
    assert all(np.isfinite(losses)), losses
    assert losses[-1] < losses[0], losses
    # F5: logits == -|f|^2 + 6 f_k - 9 for 3*I prototypes; argmax_k logit == argmax_k f
    f = ft.detach()
    closed = (-(f * f).sum(-1, keepdim=True) + 6 * f - 9).permute(0, 3, 1, 2)
    scale = lg.detach().abs().max().item()
    assert (lg.detach() - closed).abs().max().item() <= 1e-4 * scale
    assert (lg.detach().argmax(1) == f.argmax(-1)).float().mean().item() > 0.9999
    gsum = sum(float(p.grad.abs().sum()) for p in m.parameters())
    assert np.isfinite(gsum) and gsum > 0


def test_baseline_size_f16x2_step_against_exact_fp32_step():
    """768x768 bs=16, one train step in the headline arithmetic (f16x2) against the same step with exact fp32 products (the mode
    the golden fixtures of the small cases pin to the reference), the three-term bf16 split beside it as yardstick.  Logits and loss
    agree to 1e-4; sampled parameter gradients within 1e-2 (median) / 2e-2 (worst tensor) in relative 2-norm, their norms to 2e-3 (95th
    percentile) / 5e-3 -- measured 3.8e-3 / 5.8e-3 and 4.5e-4 / 1.3e-3 -- and f16x2 as close to the exact step as the three-term split.
    The BatchNorm of the ASPP image-pooling branch (network/utils.py:318-329: 16 samples per channel at this batch, its output
    broadcast over the whole map) runs on its running statistics here: with batch statistics ONE of its 4096 ReLU inputs sits within
    fp32 rounding of zero, and the side an implementation lands on moves EVERY gradient of the network by 2.3 % -- until round 5 the
    bars of this test (5e-2 / 0.2) were set around that one event, which any change of summation order could move from one mode to
    another (profiles/r05_stem_s2d_and_knife_edges.txt: the same three plans, 2.29e-2 apart or 3e-3 apart depending on that side)."""
    import utils
    g = torch.Generator(device="cpu").manual_seed(77)
    img = torch.randn(16, 3, 768, 768, generator=g).cuda()
    lab = torch.randint(0, 16, (16, 768, 768), generator=g)
    lab[:, :38] = 255
    lab = lab.cuda()
    import gc
    out = {}
    for products in ("exact", "bf16x3", "f16x2"):
        gc.collect()                     # plans of earlier tests (tens of GB each at this size) hang in reference cycles
        torch.cuda.empty_cache()
        assert torch.cuda.memory_allocated() < 20e9, "plans of earlier tests are still allocated"
        m = build(fp32_products=products)
        m.classifier.aspp.convs[4][2].eval()               # (the 16-sample BatchNorm: see above)
        crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
        lg, ctr, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        grads = {k: p.grad.detach().flatten()[:: max(1, p.numel() // 4096)].double().cpu() for k, p in m.named_parameters()}
        gn = {k: float(p.grad.detach().double().norm()) for k, p in m.named_parameters()}
        out[products] = (lg.detach()[:, :, ::16, ::16].double().cpu(), float(loss.detach()), grads, gn)
        del m, lg, ctr, ft, loss, crit
        gc.collect()
        torch.cuda.empty_cache()
    lg0, l0, g0, n0 = out["exact"]
    stats = {}
    for mode in ("bf16x3", "f16x2"):
        lg1, l1, g1, n1 = out[mode]
        assert np.isfinite(l0) and abs(l1 - l0) <= 1e-4 * abs(l0), (mode, l0, l1)
        assert (lg1 - lg0).abs().max().item() <= 1e-4 * lg0.abs().max().item(), mode
        keys = [k for k in g0 if n0[k] > 0.0]
        errs = np.array([(g1[k] - g0[k]).norm().item() / max(g0[k].norm().item(), 1e-300) for k in keys])
        nerr = np.array([abs(n1[k] - n0[k]) / n0[k] for k in keys])
        stats[mode] = (errs, nerr)
        print("%s vs exact fp32 at 16 x 768 x 768: sampled gradient error median %.2e p95 %.2e max %.2e (%s); norms median %.2e p95 "
              "%.2e max %.2e (%s); loss %.9g vs %.9g" % (mode, np.median(errs), np.percentile(errs, 95), errs.max(),
                                                      keys[int(errs.argmax())], np.median(nerr), np.percentile(nerr, 95), nerr.max(),
                                                      keys[int(nerr.argmax())], l1, l0))
        assert np.median(errs) <= 1e-2 and errs.max() <= 2e-2 and np.percentile(nerr, 95) <= 2e-3 and nerr.max() <= 5e-3, mode
    (e3, n3), (e2, n2) = stats["bf16x3"], stats["f16x2"]
    assert np.median(e2) <= 1.5 * np.median(e3) + 1e-3 and e2.max() <= 2.0 * e3.max() + 1e-3
    assert np.median(n2) <= 1.5 * np.median(n3) + 1e-4 and np.percentile(n2, 95) <= 2.0 * np.percentile(n3, 95) + 1e-4


def test_a_dropped_model_frees_its_plans():
    """The plan hands out its logits / features through an autograd node whose context holds model and plan; it must not own
    those tensors itself, or model + engine + plans form a cycle through C++ objects that the garbage collector never breaks
    (round 4: every dropped model kept its plans -- 60-100 GB each at 16 x 768 x 768 -- until the process ended)."""
    import gc
    import weakref
    import utils
    gc.collect()
    torch.cuda.empty_cache()
    base = torch.cuda.memory_allocated()
    m = build(fp32_products="f16x2")
    img, lab = g5_inputs()
    lg, ctr, ft = m(img)
    loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab, ft)
    loss.backward()
    torch.cuda.synchronize()
    plan = next(iter(m._engine.plans.values()))
    wp, wm = weakref.ref(plan), weakref.ref(m)
    assert torch.cuda.memory_allocated() - base > 500e6
    del m, lg, ctr, ft, loss, plan
    gc.collect()
    torch.cuda.empty_cache()
    assert wp() is None and wm() is None
    assert torch.cuda.memory_allocated() - base < 50e6, torch.cuda.memory_allocated() - base


def _multihead(dtype=torch.float32, fp32_products=None, state=None):
    import network
    m = network.deeplabv3plus_embedding_self_distillation_resnet101(num_classes=16, output_stride=16,
                                                                    pretrained_backbone=False)
    m.load_state_dict(state(H.shapes_of(m)) if state is not None else H.synth_state_dict(H.shapes_of(m), seed=12))
    m.cuda()
    m.set_compute_dtype(dtype, fp32_products=fp32_products)
    m.train()
    m.classifier.aspp.project[3].eval()
    m.classifier_1.aspp.project[3].eval()
    return m


@pytest.mark.parametrize("products", ["exact", "bf16x3", "f16x2"])
def test_g12l_self_distillation_model_matches_reference(products):
    """network.deeplabv3plus_embedding_self_distillation_resnet101: lists out, loss on the last head only (the base head's backward
    segment is skipped), against the conditioned 2 x 3 x 128 x 128 fixture minted from the reference model (every ReLU input with the
    proved margin, tests/tools/mint_golden_large.py) -- one set of bars for every fp32 arithmetic mode
    (1e-3 throughout, gradients included)."""
    import utils
    g = H.load_golden("g12l_multihead")
    m = _multihead(fp32_products=products, state=conditioned("g12l_multihead", 12))
    assert len(m.state_dict()) == int(g["n_keys"]) and list(m.state_dict().keys())[-4:] == [str(k) for k in g["keys"]]
    img = H.synth_tensor(12, "g12l.img", (2, 3, 128, 128)).cuda()
    lab = H.synth_labels(12, "g12l.lab", (2, 128, 128), 17, 255, ignore_rows=5).cuda()
    logits, centers, feats = m(img)
    assert [tuple(l.shape) for l in logits] == [(2, 16, 128, 128), (2, 17, 128, 128)]
    assert [tuple(f.shape) for f in feats] == [(2, 128, 128, 16), (2, 128, 128, 17)] and centers[1].shape == (17, 17)
    loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(logits[-1], lab, feats[-1])
    loss.backward()
    relclose(logits[0][:, :, ::4, ::4], T(g["logits0_sub"]), TOL, "base head logits")
    relclose(logits[1][:, :, ::4, ::4], T(g["logits1_sub"]), TOL, "incremental head logits")
    relclose(feats[1][:, ::4, ::4, :], T(g["feats1_sub"]), TOL, "incremental head features")
    assert np.allclose(H.checksum(logits[0]), g["logits0_checksum"], rtol=TOL)
    assert np.allclose(H.checksum(logits[1]), g["logits1_checksum"], rtol=TOL)
    assert abs(loss.item() - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    names = [str(n) for n in g["grad_names"]]
    rels = np.array([(np.abs(H.checksum(grads[k])[1:] - cs[1:]) / np.abs(cs[1:])).max() for k, cs in zip(names, g["grad_checksums"])])
    print("g12l %s: gradient checksums worst %.2e (%s) median %.2e" % (products, rels.max(), names[int(rels.argmax())], np.median(rels)))
    assert rels.max() <= TOL, "gradient checksums: %d of %d beyond 1e-3, worst %s" % ((rels > TOL).sum(), len(rels), names[int(rels.argmax())])
    for i, k in enumerate(str(k) for k in g["grad_keys"]):
        ref = T(g["grad_%d" % i])
        got = grads[k].detach().cpu()
        got = got if got.numel() <= 70000 else got.contiguous().flatten()[::97]
        relclose(got.reshape(ref.shape), ref, TOL, "grad " + k)
    for k in (str(k) for k in g["untouched"]):              # the reference leaves them None; here: untouched zeros
        assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k


def test_g12_self_distillation_model_matches_reference():
    """the 2 x 3 x 64 x 64 fixture (unconditioned weights): exact-fp32 mode only, see test_g5_full_train_step_matches_reference"""
    import utils
    g = H.load_golden("g12_multihead")
    m = _multihead()
    assert len(m.state_dict()) == int(g["n_keys"]) and list(m.state_dict().keys())[-4:] == [str(k) for k in g["keys"]]
    img = H.synth_tensor(12, "g12.img", (2, 3, 64, 64)).cuda()
    lab = H.synth_labels(12, "g12.lab", (2, 64, 64), 17, 255, ignore_rows=3).cuda()
    logits, centers, feats = m(img)
    assert [tuple(l.shape) for l in logits] == [(2, 16, 64, 64), (2, 17, 64, 64)]
    assert [tuple(f.shape) for f in feats] == [(2, 64, 64, 16), (2, 64, 64, 17)] and centers[1].shape == (17, 17)
    loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(logits[-1], lab, feats[-1])
    loss.backward()
    relclose(logits[0][:, :, ::4, ::4], T(g["logits0_sub"]), TOL, "base head logits")
    relclose(logits[1][:, :, ::4, ::4], T(g["logits1_sub"]), TOL, "incremental head logits")
    relclose(feats[1][:, ::4, ::4, :], T(g["feats1_sub"]), TOL, "incremental head features")
    assert np.allclose(H.checksum(logits[0]), g["logits0_checksum"], rtol=2e-3)
    assert abs(loss.item() - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    grads = OrderedDict((k, p.grad) for k, p in m.named_parameters())
    for i, k in enumerate(str(k) for k in g["grad_keys"]):
        ref = T(g["grad_%d" % i]).double()
        got = grads[k].detach().cpu()
        got = got if got.numel() <= 70000 else got.contiguous().flatten()[::97]
        e = ((got.reshape(ref.shape).double() - ref).abs().max() / (ref.abs().max() + 1e-12)).item()
        assert e <= 3 * TOL, "grad %s: rel %.3e" % (k, e)
    for k in (str(k) for k in g["untouched"]):              # the reference leaves them None; here: untouched zeros
        assert grads[k] is None or float(grads[k].abs().max()) == 0.0, k


def _mask_bits(a, b):
    """number of differing bits between two uint8 mask tensors"""
    x = torch.bitwise_xor(a, b)
    if not bool(x.any()):
        return 0
    lut = torch.tensor([bin(i).count("1") for i in range(256)], dtype=torch.int64, device=x.device)
    return int(lut[x.long()].sum())


@pytest.mark.parametrize("fixture", ["g5", "g12"])
def test_f16x2_distance_from_unconditioned_fixtures_is_relu_mask_flips(fixture):
    """ON RECORD, not a gate of the mode: how far the bench headline's arithmetic (f16x2) lands from the UNCONDITIONED reference
    fixtures g5 / g12 (2 x 3 x 64 x 64, 32-sample BatchNorms, pre-ReLU values within rounding of zero), and the proof of what the
    distance is made of.  Three runs: the exact mode (records every unit's 1-bit ReLU masks), f16x2 as it is (prints the flipped
    mask bits and the gradient checksums beyond the exact mode's bar), and f16x2 with the exact mode's masks imposed between
    its forward and its backward.  Asserted: forward outputs at 1e-3 without any help; at most a handful of mask bits differ;
    with those bits imposed EVERY gradient checksum / sampled gradient of the fixture is met as the exact mode meets it -- i.e.
    whatever distance there is, is the side of zero on which a few knife-edge ReLU inputs land.  Measured with
    the round-6 kernels: g5 0 flipped bits of 2.8 M, checksums worst 6.4e-4 (exact 3.9e-4), sampled gradients 2.8e-4 -- inside the
    exact mode's bars on its own; g12 1 flipped bit of 3.0 M, sampled gradients worst 2.2e-3 on its own (exact: 3.0e-3, bar 3e-3)."""
    import utils
    g = H.load_golden("g5_full_train" if fixture == "g5" else "g12_multihead")

    def run(products, impose=None):
        if fixture == "g5":
            m = build(fp32_products=products)
            img, lab = g5_inputs()
        else:
            m = _multihead(fp32_products=products)
            img = H.synth_tensor(12, "g12.img", (2, 3, 64, 64)).cuda()
            lab = H.synth_labels(12, "g12.lab", (2, 64, 64), 17, 255, ignore_rows=3).cuda()
        lg, _, ft = m(img)
        if fixture == "g12":
            lg, ft = lg[-1], ft[-1]
        plan = next(p for p in m._engine.plans.values() if p.training)
        torch.cuda.synchronize()
        masks = [u.mask.clone() if u.mask is not None else None for u in plan.units]
        flipped = None
        if impose is not None:
            assert len(impose) == len(masks)
            flipped = []
            for u, mine, theirs in zip(plan.units, masks, impose):
                assert (mine is None) == (theirs is None)
                if mine is not None:
                    flipped.append(_mask_bits(mine, theirs))
                    u.mask.copy_(theirs)
        loss = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(lg, lab, ft)
        loss.backward()
        torch.cuda.synchronize()
        grads = OrderedDict((k, p.grad.detach().clone() if p.grad is not None else None) for k, p in m.named_parameters())
        return lg.detach().clone(), float(loss.detach()), grads, masks, flipped

    def distance(grads):
        """relative misses against the fixture: (per-tensor checksum misses, per sampled gradient max-norm misses)"""
        if fixture == "g5":
            cs = np.array([(np.abs(H.checksum(gr)[1:] - c[1:]) / np.abs(c[1:])).max() for gr, c in zip(grads.values(), g["grad_checksums"])])
            keys = ("backbone.bn1.weight", "backbone.layer1.0.conv1.weight", "backbone.layer2.0.downsample.0.weight",
                    "backbone.layer3.5.bn2.bias", "classifier.project.0.weight", "classifier.aspp.project.1.weight",
                    "classifier.classifier.3.weight", "classifier.classifier.3.bias", "backbone.conv1.weight")
            refs = [T(g["grad__" + k.replace(".", "_")]) for k in keys]
            stride = 16
        else:
            cs = np.zeros(1)
            keys = [str(k) for k in g["grad_keys"]]
            refs = [T(g["grad_%d" % i]) for i in range(len(keys))]
            stride = 97
        el = []
        for k, ref in zip(keys, refs):
            got = grads[k].cpu()
            got = got if got.numel() <= 70000 else got.contiguous().flatten()[::stride]
            el.append(((got.reshape(ref.shape).double() - ref.double()).abs().max() / (ref.double().abs().max() + 1e-12)).item())
        return cs, np.array(el)

    lg0, loss0, grads0, masks0, _ = run("exact")
    lg1, loss1, grads1, masks1, _ = run("f16x2")
    lg2, loss2, grads2, _, flipped = run("f16x2", impose=masks0)
    ref_lg = T(g["logits"]) if fixture == "g5" else T(g["logits1_sub"])
    sub = (lambda t: t) if fixture == "g5" else (lambda t: t[:, :, ::4, ::4])
    relclose(sub(lg1), ref_lg, TOL, "f16x2 logits vs the unconditioned fixture")
    assert abs(loss1 - float(g["loss"])) <= TOL * abs(float(g["loss"]))
    nbits = sum(flipped)
    total = sum(int(mk.numel()) * 4 for mk in masks0 if mk is not None)
    where = [(i, n) for i, n in enumerate(flipped) if n]
    cs0, el0 = distance(grads0)
    cs1, el1 = distance(grads1)
    cs2, el2 = distance(grads2)
    bar_cs, bar_el = 2e-3, (2 * TOL if fixture == "g5" else 3 * TOL)          # the exact mode's bars on these fixtures
    print("%s unconditioned: f16x2 flips %d of %d ReLU mask bits against the exact mode (unit index, bits: %s)" % (fixture, nbits, total, where))
    print("   exact          : checksum misses > %.0e: %d (worst %.2e); sampled gradients worst %.2e" % (bar_cs, (cs0 > bar_cs).sum(), cs0.max(), el0.max()))
    print("   f16x2 as it is : checksum misses > %.0e: %d (worst %.2e); sampled gradients worst %.2e" % (bar_cs, (cs1 > bar_cs).sum(), cs1.max(), el1.max()))
    print("   f16x2 + exact's masks: checksum misses: %d (worst %.2e); sampled gradients worst %.2e" % ((cs2 > bar_cs).sum(), cs2.max(), el2.max()))
    assert nbits <= 64, "f16x2 and the exact mode disagree on %d ReLU mask bits: more than knife edges" % nbits
    print("   f16x2 on its own meets the exact mode's bars on this fixture: %s" % bool(cs1.max() <= bar_cs and el1.max() <= bar_el))
    # with the exact mode's masks, f16x2 lands where the exact mode lands (the fixture's bar, or the exact mode's own distance from
    # the fixture + 5e-4 where that one sits AT its bar: g12's worst sampled gradient is 2.96e-3 of 3e-3 in the exact mode itself)
    assert (cs2 <= np.maximum(bar_cs, cs0 + 5e-4)).all() and (el2 <= np.maximum(bar_el, el0 + 5e-4)).all(), \
        "f16x2 misses the fixture even with the exact mode's ReLU masks imposed"
    # (and whatever f16x2 misses on its own is bounded: one flipped element moves percent of its channel's d(beta), per mille behind it)
    assert cs1.max() <= 5e-2 and el1.max() <= 5e-2


def test_self_distillation_model_both_heads_against_oracle():
    """a loss on BOTH heads (every segment of the backward plan runs, d(out) / d(low) accumulate) vs the fp64 oracle"""
    import utils
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(32, torch.get_num_threads() or 8))
    m = _multihead()
    img = H.synth_tensor(12, "g12.img", (2, 3, 64, 64))
    lab16 = H.synth_labels(13, "mh.lab16", (2, 64, 64), 16, 255, ignore_frac=0.05)
    lab17 = H.synth_labels(12, "g12.lab", (2, 64, 64), 17, 255, ignore_rows=3)
    logits, _, feats = m(img.cuda())
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
    (crit(logits[0], lab16.cuda(), feats[0]) + 0.5 * crit(logits[1], lab17.cuda(), feats[1])).backward()
    o = O.deeplabv3plus_embedding_self_distillation_resnet101(output_stride=16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=12))
    o = o.double()
    o.train()
    o.classifier.aspp.project[3].eval()
    o.classifier_1.aspp.project[3].eval()
    ol, _, _ = o(img.double())
    (O.dml_loss(ol[0], lab16, alpha=0.01, ignore_index=255) + 0.5 * O.dml_loss(ol[1], lab17, alpha=0.01, ignore_index=255)).backward()
    relclose(logits[0], ol[0], TOL, "base head logits")
    relclose(logits[1], ol[1], TOL, "incremental head logits")
    errs = []
    for (k, p), (_, q) in zip(m.named_parameters(), o.named_parameters()):
        sc = q.grad.abs().max().item() + 1e-30
        errs.append((p.grad.detach().cpu().double() - q.grad).abs().max().item() / sc)
    errs = np.array(errs)
    print("both-heads grad error vs fp64: median %.2e p95 %.2e max %.2e" % (np.median(errs), np.percentile(errs, 95), errs.max()))
    # a missing or doubled head contribution to d(out) / d(low) would put O(1) errors on every backbone tensor; single
    # early-layer tensors sit at a few % from ReLU sign flips on this input (measured: median 1.7e-4, p95 6.6e-3, max 8.4e-2)
    assert np.median(errs) <= 1e-3 and np.percentile(errs, 95) <= 3e-2 and errs.max() <= 0.2


def test_g3_head_fixture_through_the_hip_plan():
    """G3 (minted from the reference's DeepLabHeadV3Plus, network/utils.py:8-32,308-361): the head alone -- low-level
    projection, five ASPP branches incl. the image-pooling BatchNorm, projection, x4 bilinear, concat, 3x3, final 1x1
    with bias -- forward, input gradients, parameter gradients and running statistics, from the plan's own head pieces."""
    from collections import OrderedDict
    import network.modeling as NM
    g = H.load_golden("g3_head")
    head = NM.DeepLabHeadV3Plus(2048, 256, 16, [6, 12, 18])
    shapes = OrderedDict(("classifier." + k, tuple(v.shape)) for k, v in head.state_dict().items())
    sd = H.synth_state_dict(shapes, seed=3)
    head.load_state_dict(OrderedDict((k[len("classifier."):], v) for k, v in sd.items()))
    head.train()
    head.aspp.project[3].eval()
    low = H.synth_tensor(3, "g3.low", (2, 256, 16, 16))
    out = H.synth_tensor(3, "g3.out", (2, 2048, 4, 4))
    wgt = H.synth_tensor(3, "g3.wgt", (2, 16, 16, 16))
    plan = H.piece_plan("head", head, [tuple(low.shape), tuple(out.shape)])
    y, (dlow, dout) = plan.run([low, out], wgt)
    relclose(y, T(g["y"]), TOL, "head output")
    relclose(dlow, T(g["dlow"]), TOL, "d low_level")
    relclose(dout, T(g["dout"]), TOL, "d out")
    pg = dict((k, p.grad) for k, p in head.named_parameters())
    for n, cs in zip([str(n) for n in g["grad_names"]], g["grad_checksums"]):
        assert np.allclose(H.checksum(pg[n]), cs, rtol=2e-3, atol=1e-5), n
    relclose(pg["classifier.3.weight"], T(g["grad__classifier_3_weight"]), TOL, "final conv weight gradient")
    relclose(pg["aspp.convs.2.0.weight"].detach().cpu()[::8, ::64], T(g["grad_sample__aspp_convs_2_0_weight"]), TOL, "aspp d12 sample")
    bufs = dict(head.named_buffers())
    relclose(bufs["project.1.running_var"], T(g["rv_project"]), TOL, "running_var")
    relclose(bufs["aspp.convs.4.2.running_mean"], T(g["rm_pool"]), TOL, "pooled running_mean")


@pytest.mark.parametrize("name,cfg", [
    ("s1", dict(inplanes=64, planes=16, stride=1, dilation=1, downsample=False)),
    ("s2", dict(inplanes=32, planes=16, stride=2, dilation=1, downsample=True)),
    ("d2", dict(inplanes=64, planes=16, stride=1, dilation=2, downsample=False)),
])
def test_g4_bottleneck_fixture_through_the_hip_plan(name, cfg):
    """G4 (minted from the reference's Bottleneck, backbone/resnet.py:75-115): stride 1, stride 2 + downsample branch,
    dilation 2 -- output, input gradient, every parameter gradient and the running statistics from Plan.block_fwd /
    block_bwd, the code the full model's plan is made of."""
    from collections import OrderedDict
    import torch.nn as nn
    import network.modeling as NM
    g = H.load_golden("g4_bottleneck")
    ds = None
    if cfg["downsample"]:
        ds = nn.Sequential(nn.Conv2d(cfg["inplanes"], cfg["planes"] * 4, 1, stride=cfg["stride"], bias=False),
                           nn.BatchNorm2d(cfg["planes"] * 4))
    blk = NM.Bottleneck(cfg["inplanes"], cfg["planes"], cfg["stride"], cfg["dilation"], ds)
    shapes = OrderedDict(("backbone.blk." + k, tuple(v.shape)) for k, v in blk.state_dict().items())
    sd = H.synth_state_dict(shapes, seed=4)
    blk.load_state_dict(OrderedDict((k[len("backbone.blk."):], v) for k, v in sd.items()))
    blk.train()
    x = H.synth_tensor(4, "g4.x." + name, (2, cfg["inplanes"], 8, 8))
    yshape = tuple(g[name + "_y"].shape)
    w = H.synth_tensor(4, "g4.w." + name, yshape)
    plan = H.piece_plan("block", blk, [tuple(x.shape)])
    y, (dx,) = plan.run([x], w)
    relclose(y, T(g[name + "_y"]), TOL, "block output")
    relclose(dx, T(g[name + "_dx"]), TOL, "d x")
    for k, p in blk.named_parameters():
        relclose(p.grad, T(g[name + "_grad__" + k.replace(".", "_")]), 2 * TOL, "grad " + k)
    for k, b in blk.named_buffers():
        if "num_batches" not in k:
            relclose(b, T(g[name + "_buf__" + k.replace(".", "_")]), TOL, k)


@pytest.mark.parametrize("both", [True, False])
def test_multihead_bf16_fp32_staged_gradients(both, monkeypatch):
    """DML_GRAD_STAGE32=1 on the two-head bf16 plan: d(out) is summed in fp32 over BOTH heads' segments and rounded by
    a conversion at the head of the backbone segment (which head runs last is only known per step), d(low) by layer2.0's
    downsample data gradient.  Same forward, so the gradients must equal the per-producer-rounding plan's up to that
    rounding noise -- with a loss on both heads and with the base head's segment skipped."""
    import utils
    img = H.synth_tensor(12, "g12.img", (2, 3, 96, 96)).cuda()
    lab16 = H.synth_labels(13, "mh.lab16", (2, 96, 96), 16, 255, ignore_frac=0.05).cuda()
    lab17 = H.synth_labels(12, "g12.lab", (2, 96, 96), 17, 255, ignore_rows=3).cuda()
    grads = {}
    for stage in ("0", "1"):
        monkeypatch.setenv("DML_GRAD_STAGE32", stage)
        m = _multihead(torch.bfloat16)
        logits, _, feats = m(img)
        crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
        loss = 0.5 * crit(logits[1], lab17, feats[1])
        if both:
            loss = loss + crit(logits[0], lab16, feats[0])
        loss.backward()
        plan = next(p for k, p in m._engine.plans.items() if k[4])
        n_stage = sum(1 for fn, args in plan.bwd if fn is plan.lib.dml_conv_igemm and args[0]._obj.acc32)
        assert n_stage == (4 if stage == "1" else 0)          # the four downsample blocks; d(out) goes through the conversion
        grads[stage] = {k: p.grad.detach().double().cpu().flatten() for k, p in m.named_parameters() if p.grad is not None}
    omc = []
    for k, a in grads["0"].items():
        b = grads["1"][k]
        if float(a.norm()) == 0.0 and float(b.norm()) == 0.0:
            continue
        omc.append(1.0 - float(a @ b) / (float(a.norm()) * float(b.norm()) + 1e-30))
    omc = np.array(omc)
    print("two heads, bf16: staged vs per-producer 1 - cos median %.2e max %.2e over %d tensors" % (np.median(omc), omc.max(), len(omc)))
    assert len(omc) > 300 and np.median(omc) <= 2e-3 and omc.max() <= 5e-2


def test_incremental_head_recipe_skips_the_trunk_backward():
    """main_self_distillation.py:354-357,432-435,497-499: only `classifier_1` trains, every BatchNorm2d on running
    statistics, loss on the last head.  With requires_grad = False on trunk and base head the backward plan stops at
    the new head's inputs; the new head's gradients are the same as in the full backward."""
    import utils
    img = H.synth_tensor(41, "inc.img", (2, 3, 64, 80)).cuda()
    lab = H.synth_labels(41, "inc.lab", (2, 64, 80), 17, 255, ignore_frac=0.05).cuda()
    grads = {}
    for frozen in (False, True):
        m = _multihead()
        _freeze_bn("all")(m)
        if frozen:
            for p in list(m.backbone.parameters()) + list(m.classifier.parameters()):
                p.requires_grad_(False)
        lg, _, ft = m(img)
        utils.CrossEntropyLoss(ignore_index=255)(lg[-1], lab, ft[-1]).backward()
        grads[frozen] = {k: p.grad.detach().clone() for k, p in m.classifier_1.named_parameters()}
        st = m._engine.store
        if frozen:
            assert float(st.flat_g[:st.split].abs().max()) == 0.0            # nothing was written for the trunk
            assert m._engine.plans and all(pl.skip_ranges for pl in m._engine.plans.values() if pl.training)
        else:
            assert float(st.flat_g[:st.split].abs().max()) > 0.0
    for k in grads[True]:      # same kernels on the same inputs; split-K partial sums may be folded in another order
        relclose(grads[True][k], grads[False][k], 1e-5, k)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_deferred_loss_gradient_equals_materialised(dtype):
    """utils.DMLLoss(fused_backward=True): the loss hands the model a marker instead of a gradient tensor and the head's
    backward runs as one kernel; every parameter gradient must equal the materialised path's.  Also the guard: a second
    consumer of the logits makes the backward raise instead of using a poisoned gradient."""
    import utils
    img = H.synth_tensor(61, "lazy.img", (2, 3, 64, 96)).cuda()
    lab = H.synth_labels(61, "lazy.lab", (2, 64, 96), 16, 255, ignore_frac=0.05).cuda()
    grads = {}
    for fused in (False, True):
        m = build(dtype=dtype, seed=61)
        lg, _, ft = m(img)
        loss = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=fused)(lg, lab, ft)
        (loss * 0.5).backward()
        torch.cuda.synchronize()
        grads[fused] = {k: p.grad.detach().clone() for k, p in m.named_parameters()}
        plan = next(p for k, p in m._engine.plans.items() if k[4])
        assert (plan.heads[0].unfused_range in plan.skip_ranges) == fused
    tol = 2e-5 if dtype == torch.float32 else 3e-2       # bf16: the low-resolution gradient is rounded once in both paths,
    worst = 0.0                                          # 1-ulp differences there spread through the backward
    for k in grads[True]:
        e = H.rel_err(grads[True][k], grads[False][k])
        worst = max(worst, e)
        assert e <= tol, (k, e)
    print("deferred vs materialised loss gradient (%s): worst parameter-gradient difference %.2e" % (dtype, worst))
    # odd size: not an exact x4 upsample -> the marker is materialised by dml_loss_bwd, same result as the plain path
    img2 = H.synth_tensor(62, "lazy.img2", (2, 3, 50, 66)).cuda()
    lab2 = H.synth_labels(62, "lazy.lab2", (2, 50, 66), 16, 255, ignore_frac=0.05).cuda()
    g2 = {}
    for fused in (False, True):
        m = build(dtype=torch.float32, seed=62)
        lg, _, ft = m(img2)
        utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=fused)(lg, lab2, ft).backward()
        g2[fused] = m.classifier.classifier[3].weight.grad.detach().clone()
    relclose(g2[True], g2[False], 1e-6, "materialised marker, odd size")
    # a second consumer of the logits: loud failure
    m = build(dtype=torch.float32, seed=61)
    lg, _, ft = m(img)
    total = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=True)(lg, lab, ft) + 1e-3 * lg.mean()
    with pytest.raises(RuntimeError):
        total.backward()
