"""Anomaly sub-project model: dilated deep-stem ResNet-50 + pyramid-pooling embedding decoder,
inference branch.  CPU: oracle vs fixture G14 (minted from the reference's own classes, tests/tools/mint_golden_ppm.py) and the
module-tree contract; GPU: the HIP plan vs fixture / oracle through the C ABI, and its three kernels vs torch ops."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import helpers as H

TOL = 1e-3
G14 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g14_ppm.npz")
EXTRA = ("_tmp_running_mean", "_tmp_running_var", "_running_iter")


def _imgs():
    return [H.synth_tensor(14, "ppm.img0", (1, 3, 64, 96)), H.synth_tensor(14, "ppm.img1", (1, 3, 88, 120)),
            H.synth_tensor(14, "ppm.img2", (2, 3, 72, 72))]


def _weights(module):
    shapes = H.shapes_of(module)
    return H.synth_state_dict({k: v for k, v in shapes.items() if not k.endswith(EXTRA)}, seed=14)


def relclose(got, ref, tol, what):
    got, ref = torch.as_tensor(got).double().cpu(), torch.as_tensor(ref).double().cpu()
    err, scale = (got - ref).abs().max().item(), ref.abs().max().item() + 1e-12
    assert err <= tol * scale, "%s: max|d|=%.3e scale=%.3e rel=%.3e > %.1e" % (what, err, scale, err / scale, tol)


def test_oracle_matches_reference_fixture():
    from oracle import ppm_ref as O
    g = np.load(G14)
    o = O.SegmentationModuleOODRef()
    assert list(o.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in o.state_dict().values()] == [str(s) for s in g["key_shapes"]]
    o.load_state_dict(_weights(o), strict=False)
    o.eval()
    imgs, seg = _imgs(), tuple(int(v) for v in g["seg"])
    with torch.no_grad():
        for i, s in ((0, seg), (1, seg), (2, (72, 72))):
            p, f = o(imgs[i], s)
            relclose(p, g["pred%d" % i], 2e-5, "pred%d" % i)
            relclose(f, g["ft%d" % i], 2e-5, "ft%d" % i)
        ms, mf = O.evaluate_multiscale(o, imgs[:2], seg)
        relclose(ms, g["ms_scores"], 2e-5, "multi-scale scores")
        relclose(mf, g["ms_ft"], 2e-5, "multi-scale features")


def test_product_module_tree_matches_reference_fixture():
    import models
    g = np.load(G14)
    enc = models.ModelBuilder.build_encoder("resnet50dilated", fc_dim=2048)
    dec = models.ModelBuilder.build_decoder("ppm_deepsup_embedding", fc_dim=2048, num_class=13, use_softmax=True)
    m = models.SegmentationModuleOOD(enc, dec, None)
    assert list(m.state_dict().keys()) == [str(k) for k in g["keys"]]
    assert [str(tuple(v.shape)) for v in m.state_dict().values()] == [str(s) for s in g["key_shapes"]]
    assert enc.layer3[0].conv2.stride == (1, 1) and enc.layer3[0].conv2.dilation == (1, 1)      # models.py:315-328
    assert enc.layer3[1].conv2.dilation == (2, 2) and enc.layer4[0].conv2.dilation == (2, 2)
    assert enc.layer4[2].conv2.dilation == (4, 4) and enc.layer4[0].downsample[0].stride == (1, 1)
    assert float(dec.conv_last[1].bias[0].detach()) == pytest.approx(1e-4)                               # weights_init
    m.eval()
    with pytest.raises(RuntimeError, match="ROCm device"):
        m({"img_data": torch.zeros(1, 3, 64, 64)}, segSize=(64, 64))
    with pytest.raises(NotImplementedError):
        m({"img_data": torch.zeros(1, 3, 64, 64)})
    with pytest.raises(NotImplementedError):
        models.ModelBuilder.build_decoder("upernet")


def _product(dtype=torch.float32):
    import models
    enc = models.ModelBuilder.build_encoder("resnet50dilated", fc_dim=2048)
    dec = models.ModelBuilder.build_decoder("ppm_deepsup_embedding", fc_dim=2048, num_class=13, use_softmax=True)
    m = models.SegmentationModuleOOD(enc, dec, None)
    m.load_state_dict(_weights(m), strict=False)
    m.cuda().eval()
    return m.set_compute_dtype(dtype)


@pytest.mark.gpu
def test_hip_plan_matches_reference_fixture():
    import models
    g = np.load(G14)
    m = _product()
    imgs, seg = _imgs(), tuple(int(v) for v in g["seg"])
    for i, s in ((0, seg), (1, seg), (2, (72, 72))):
        p, f = m({"img_data": imgs[i].cuda()}, segSize=s)
        assert p.shape == g["pred%d" % i].shape and p.dtype == torch.float32
        relclose(p, g["pred%d" % i], TOL, "pred%d" % i)
        relclose(f, g["ft%d" % i], TOL, "ft%d" % i)
        assert (p.argmax(1).cpu().numpy() == g["pred%d" % i].argmax(1)).mean() > 0.995
    ms, mf = models.evaluate_multiscale(m, [t.cuda() for t in imgs[:2]], seg)
    relclose(ms, g["ms_scores"], TOL, "multi-scale scores")
    relclose(mf, g["ms_ft"], TOL, "multi-scale features")
    # the same shape twice in one list (plans are not re-entrant: second use goes to a later wave), twice in a row
    for _ in range(2):
        ms3, mf3 = models.evaluate_multiscale(m, [imgs[0].cuda(), imgs[1].cuda(), imgs[0].cuda()], seg)
        want = (2 * g["pred0"] + g["pred1"]) / 3
        relclose(ms3, want, TOL, "three copies, one shape repeated")
        relclose(mf3, (2 * g["ft0"] + g["ft1"]) / 3, TOL, "three copies, features")
    # bf16 plan tracks the fp32 one
    mb = _product(torch.bfloat16)
    pb, fb = mb({"img_data": imgs[1].cuda()}, segSize=seg)
    relclose(pb, g["pred1"], 0.1, "bf16 pred")
    relclose(fb, g["ft1"], 0.1, "bf16 features")


@pytest.mark.gpu
def test_hip_plan_vs_oracle_fresh_sizes_and_dissum_scores():
    """Sizes the fixture does not hold (odd, non-multiple-of-8 inputs; B = 2), and the scoring of
    eval_ood_traditional.py:301-305 on the result (device kernel vs numpy restatement)."""
    from oracle import ppm_ref as O
    import utils
    m = _product()
    o = O.SegmentationModuleOODRef()
    o.load_state_dict(_weights(o), strict=False)
    o.eval()
    for shape, seg in (((2, 3, 97, 131), (45, 61)), ((1, 3, 56, 200), (112, 400))):
        img = H.synth_tensor(15, "ppm.fresh%s" % (shape,), shape)
        with torch.no_grad():
            rp, rf = o(img, seg)
        p, f = m({"img_data": img.cuda()}, segSize=seg)
        relclose(p, rp, TOL, "pred %s" % (shape,))
        relclose(f, rf, TOL, "ft %s" % (shape,))
    s = utils.dissum_score(p, clip=400.0, inclusive=True)
    r = -rp.sum(1)
    r = torch.clamp(r, max=400.0)
    r = (r - r.amin((1, 2), keepdim=True)) / (r.amax((1, 2), keepdim=True) - r.amin((1, 2), keepdim=True))
    relclose(s, r, TOL, "dissum score")


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_adaptive_avgpool_kernel(dtype):
    from dmlnet import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    for (B, Hh, Ww, Cc, ld) in ((2, 11, 17, 64, 96), (1, 64, 37, 2048, 2048), (3, 6, 6, 16, 16)):
        x = torch.randn(B, Hh, Ww, ld, device="cuda").to(dtype)
        for S in (1, 2, 3, 6):
            y = torch.empty(B, S, S, Cc, device="cuda", dtype=dtype)
            ws = torch.empty(int(lib.dml_adaptive_avgpool_ws_elems(B, Hh, Ww, Cc, S)), device="cuda")
            _lib.check(lib.dml_adaptive_avgpool_fwd(x.data_ptr(), y.data_ptr(), ws.data_ptr(), B, Hh, Ww, Cc, ld, S,
                                                    1 if dtype == torch.bfloat16 else 0, st), "pool")
            ref = torch.nn.functional.adaptive_avg_pool2d(x[..., :Cc].float().permute(0, 3, 1, 2), S).permute(0, 2, 3, 1)
            tol = 1e-5 if dtype == torch.float32 else 8e-3
            assert (y.float() - ref).abs().max().item() <= tol * (ref.abs().max().item() + 1e-6) + 1e-6, (B, Hh, Ww, Cc, S)
    ws = torch.empty(64, device="cuda")
    assert lib.dml_adaptive_avgpool_fwd(x.data_ptr(), y.data_ptr(), ws.data_ptr(), 1, 4, 4, 16, 16, 6, 0, st) == -1   # S > H


@pytest.mark.gpu
def test_lowres_distance_and_upsample_kernels():
    from dmlnet import _lib
    lib = _lib.load()
    st = torch.cuda.current_stream().cuda_stream
    B, h, w, K, Kp = 2, 9, 13, 13, 16
    emb = torch.zeros(B, h, w, Kp, device="cuda")
    emb[..., :K] = torch.randn(B, h, w, K, device="cuda") * 2
    protos = torch.zeros(K, Kp, device="cuda")
    protos[:, :K] = 3.0 * torch.eye(K, device="cuda")
    out = torch.full((B, h, w, Kp), 7.0, device="cuda")
    _lib.check(lib.dml_proto_dist_nhwc(emb.data_ptr(), protos.data_ptr(), out.data_ptr(), B * h * w, K, Kp, Kp, Kp, st), "dist")
    ref = -((emb[..., None, :K] - 3.0 * torch.eye(K, device="cuda")) ** 2).sum(-1)
    assert torch.allclose(out[..., :K], ref, rtol=1e-6, atol=1e-5) and (out[..., K:] == 0).all()
    for (Hs, Ws) in ((70, 100), (33, 51), (9, 13), (5, 7)):
        dst = torch.empty(B, K, Hs, Ws, device="cuda")
        _lib.check(lib.dml_upsample_nhwc_to_nchw(out.data_ptr(), dst.data_ptr(), B, h, w, Kp, K, Hs, Ws, 1.0, 0, st), "up")
        r = torch.nn.functional.interpolate(ref.permute(0, 3, 1, 2), size=(Hs, Ws), mode="bilinear", align_corners=False)
        assert torch.allclose(dst, r, rtol=1e-5, atol=1e-4), (Hs, Ws)
        _lib.check(lib.dml_upsample_nhwc_to_nchw(out.data_ptr(), dst.data_ptr(), B, h, w, Kp, K, Hs, Ws, 0.5, 1, st), "up+")
        assert torch.allclose(dst, 1.5 * r, rtol=1e-5, atol=1e-4), (Hs, Ws)
