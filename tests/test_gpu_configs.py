"""GPU: BASELINE.json configs that need the real sizes -- config #5 (1024 x 2048 open-world inference: eval forward,
argmax / max-softmax, dissum score, novel-prototype relabel; test_embedding.py:328-350,428-445 of the reference) on the
HIP path against the oracle, this repository's test_embedding.py driver end to end (also on two ranks), and
checkpoint resume of the fused optimizer (main_embedding.py:421-434 of the reference)."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers as H

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _calibrated_pair(seed, size):
    """oracle and HIP model with the same conditioned weights and running statistics calibrated on `size` inputs (random
    running statistics would not match the activations of a random-init net: saturated logits, nothing to compare)"""
    import network
    from oracle import dmlnet_ref as O
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=seed))
    o.train()
    o.classifier.aspp.project[3].eval()
    cal = H.synth_tensor(seed + 1, "cfg5.cal", (2, 3, size[0] // 4, size[1] // 4))
    with torch.no_grad():
        for _ in range(3):
            o(cal)
    o.eval()
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.load_state_dict(o.state_dict())
    m.cuda().eval()
    return m, o


@pytest.mark.parametrize("products", ["exact", "f16x2"])
def test_config5_full_size_inference_and_scores_against_oracle(products):
    """1 x 3 x 1024 x 2048, both fp32 modes (the reference's literal fp32 products, and the bench headline's two-plane fp16
    products), forked eval plan (the default at this size), against the fp32 oracle on the same input: logits / features at
    1e-3, argmax, max-softmax score, dissum map (both clip rules) and the relabel mask of test_embedding.py:428-445."""
    import utils
    from oracle import dmlnet_ref as O
    torch.set_num_threads(min(64, torch.get_num_threads() or 8))
    m, o = _calibrated_pair(51, (1024, 2048))
    m.set_compute_dtype(torch.float32, fp32_products=products)
    img = H.synth_tensor(51, "cfg5.img", (1, 3, 1024, 2048))
    with torch.no_grad():
        lg, ctr, ft = m(img.cuda())
        olg, _, oft = o(img)
    assert lg.shape == (1, 16, 1024, 2048) and ft.shape == (1, 1024, 2048, 16)
    e_lg, e_ft = H.rel_err(lg, olg), H.rel_err(ft, oft)
    print("config5 fp32 (%s products): logits rel %.2e, features rel %.2e" % (products, e_lg, e_ft))
    plan = next(iter(m._engine.plans.values()))
    assert getattr(plan, "fwd_forks", None), "the 1024x2048 batch-1 eval plan is expected to fork the ASPP branches"
    assert e_lg <= 1e-3 and e_ft <= 1e-3
    preds, msp = utils.argmax_msp(lg)
    opred = olg.argmax(1)
    agree = (preds.cpu() == opred).float().mean().item()
    assert agree > 0.999, agree
    assert H.max_abs(msp, O.msp_score(olg)) <= 2e-3
    for clip, inclusive in ((1000.0, False), (400.0, True)):
        sc = utils.dissum_score(lg, clip=clip, inclusive=inclusive)
        ref = O.dissum_score(olg[0].numpy().astype(np.float32), clip, inclusive)
        assert H.max_abs(sc[0], T(ref)) <= 2e-3, (clip, inclusive)
    # novel-prototype relabel: a prototype inside the feature cloud so that the mask is neither empty nor full
    proto = oft[0, ::64, ::64].reshape(-1, 16).mean(0).double().numpy()
    thresh = float(np.quantile(-((oft[0, ::8, ::8].reshape(-1, 16).numpy() - proto) ** 2).sum(1), 0.9))
    got = utils.novel_relabel(preds.clone(), lg, ft, proto, thresh, 16)
    ref = O.novel_relabel(opred[0].numpy().copy(), olg[0].numpy(), oft[0].numpy(), proto, thresh, 16)
    frac = float((T(ref) == 16).float().mean())
    diff = (got[0].cpu() != T(ref)).float().mean().item()
    print("config5 relabel: %.3f of the pixels relabelled by the oracle, disagreement %.2e" % (frac, diff))
    assert diff < 2e-3
    # the distance head's closed form at full size (F5) and bf16 mode tracking it
    closed = (-(ft * ft).sum(-1, keepdim=True) + 6 * ft - 9).permute(0, 3, 1, 2)
    assert (lg - closed).abs().max().item() <= 1e-4 * lg.abs().max().item()
    if products != "exact":
        return
    m.set_compute_dtype(torch.bfloat16)
    with torch.no_grad():
        lgb, _, _ = m(img.cuda())
    from oracle import bf16_emu
    bf16_emu.emulate_bf16_storage(o)
    with torch.no_grad():
        olgb, _, _ = o(img)
    e_b = H.rel_err(lgb, olgb)
    print("config5 bf16 vs bf16-storage oracle: logits rel %.2e (bf16 vs fp32 oracle %.2e)" % (e_b, H.rel_err(olgb, olg)))
    assert e_b <= 2e-2


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_eval_driver_synthetic_one_and_two_ranks():
    """this repository's test_embedding.py (the reference's evaluation entry point) end to end on synthetic frames; on two
    ranks (gloo, sharing the GPU) the confusion matrix and the per-image measures are reduced, so rank 0 prints the same
    scores as the single process."""
    drv = os.path.join(H.PKG, "test_embedding.py")
    args = [drv, "--synthetic", "--height", "256", "--width", "512", "--num_images", "4", "--dtype", "f32"]
    r1 = subprocess.run([sys.executable] + args, capture_output=True, text=True, cwd=H.PKG, timeout=900)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-3000:]
    assert "Mean IoU" in r1.stdout and "AUROC" in r1.stdout.upper()
    env = dict(os.environ, DML_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r2 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port())] + args, env=env, capture_output=True, text=True,
                        cwd=H.PKG, timeout=900)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-3000:]

    def scores(text):
        out = {}
        for ln in text.splitlines():
            for key in ("Overall Acc", "Mean Acc", "FreqW Acc", "Mean IoU"):
                if ln.startswith(key + ":"):
                    out[key] = float(ln.split(":")[1])
        return out

    s1, s2 = scores(r1.stdout), scores(r2.stdout)
    assert len(s1) == 4 and s1.keys() == s2.keys()
    for k in s1:
        assert abs(s1[k] - s2[k]) <= 1e-6, (k, s1[k], s2[k])
    # fewer images than ranks: rank 1 scores nothing and never creates its confusion matrix, but must still take part in
    # the SAME collectives (it used to skip the matrix all_reduce and pair its next one with rank 0's: hang / corruption)
    one = [drv, "--synthetic", "--height", "256", "--width", "512", "--num_images", "1", "--dtype", "f32"]
    r3 = subprocess.run([sys.executable] + one, capture_output=True, text=True, cwd=H.PKG, timeout=900)
    r4 = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(_free_port())] + one, env=env, capture_output=True, text=True,
                        cwd=H.PKG, timeout=300)
    assert r3.returncode == 0 and r4.returncode == 0, r4.stdout[-2000:] + r4.stderr[-3000:]
    s3, s4 = scores(r3.stdout), scores(r4.stdout)
    assert len(s3) == 4 and s3.keys() == s4.keys()
    for k in s3:
        assert abs(s3[k] - s4[k]) <= 1e-6, (k, s3[k], s4[k])


def test_optimizer_state_survives_save_and_resume():
    """--continue_training (main_embedding.py:421-434 of the reference): momentum buffers restored by load_state_dict()
    BEFORE the model reaches the GPU must be carried into the flat momentum buffer; the resumed run then equals the
    uninterrupted one."""
    import io
    import network
    import utils
    from dmlnet.optim import FusedSGD

    def make():
        m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
        m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=1))
        opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 1e-3}, {"params": m.classifier.parameters(), "lr": 1e-2}],
                       lr=1e-2, momentum=0.9, weight_decay=1e-4).bind(m)
        return m, opt

    img = H.synth_tensor(5, "g8.img", (2, 3, 64, 64)).cuda()
    lab = H.synth_labels(5, "g8.lab", (2, 64, 64), 16, 255, ignore_rows=3).cuda()
    crit = utils.CrossEntropyLoss(ignore_index=255)

    def step(m, opt):
        opt.zero_grad()
        lg, _, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        opt.step()
        return loss.item()

    def prepare(m):
        m.cuda().train()
        m.set_compute_dtype(torch.float32)
        m.classifier.aspp.project[3].eval()

    a, opt_a = make()
    prepare(a)
    for _ in range(2):
        step(a, opt_a)
    buf = io.BytesIO()
    torch.save({"model_state": a.state_dict(), "optimizer_state": opt_a.state_dict()}, buf)
    loss_a = step(a, opt_a)                                   # the uninterrupted third step

    buf.seek(0)
    ck = torch.load(buf, map_location="cpu")
    b, opt_b = make()                                         # still on the CPU, as in the driver
    b.load_state_dict(ck["model_state"])
    opt_b.load_state_dict(ck["optimizer_state"])
    prepare(b)
    loss_b = step(b, opt_b)
    assert abs(loss_a - loss_b) <= 1e-6 * abs(loss_a)
    sa, sb = a.state_dict(), b.state_dict()
    for k in ("backbone.conv1.weight", "backbone.layer3.7.conv2.weight", "classifier.classifier.3.weight",
              "classifier.classifier.3.bias", "backbone.layer2.1.bn2.weight"):
        assert H.rel_err(sb[k], sa[k]) <= 1e-6, k
    # the momentum itself: non-zero and equal (a zero-restart would differ at the 1e-2 level after one step)
    va, vb = a._engine.store.flat_v, b._engine.store.flat_v
    assert float(va.abs().max()) > 0 and H.rel_err(vb, va) <= 1e-6
    # a reload in the MIDDLE of a run (after the flat momentum buffer exists): c takes two steps of its own, then loads
    # the checkpoint; its next step must equal a's third step, i.e. the kernel must update the restored momentum
    buf.seek(0)
    ck = torch.load(buf, map_location="cpu")
    c, opt_c = make()
    prepare(c)
    step(c, opt_c)
    step(c, opt_c)
    step(c, opt_c)                                            # c's momentum now differs from the checkpoint's
    c.load_state_dict(ck["model_state"])
    opt_c.load_state_dict(ck["optimizer_state"])
    loss_c = step(c, opt_c)
    assert abs(loss_a - loss_c) <= 1e-6 * abs(loss_a)
    vc = c._engine.store.flat_v
    assert H.rel_err(vc, va) <= 1e-6
    p0 = next(iter(c.backbone.parameters()))
    assert opt_c.state[p0]["momentum_buffer"].data_ptr() == c._engine.store._view(vc, 0, p0).data_ptr()
