"""Input pipeline: oracle vs the reference fixtures G9 and the live Pillow (CPU), and the HIP
kernels vs fixtures / oracle through the C ABI (GPU)."""
import glob
import os
import random

import numpy as np
import pytest
import torch

import helpers as H  # noqa: F401  (path setup)
from oracle import transforms_ref as TR

G9 = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "g9_aug_*.npz")))
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def _fixture(path):
    g = np.load(path)
    return g, int(g["seed"]), tuple(int(v) for v in g["crop"])


@pytest.mark.parametrize("path", G9, ids=[os.path.basename(p)[7:-4] for p in G9])
def test_oracle_reproduces_reference_pipeline(path):
    """bit-exact incl. the order of `random` draws: the fixture stores only the seed"""
    g, seed, crop = _fixture(path)
    p = TR.sample_params(random.Random(seed), g["img"].shape[0], g["img"].shape[1], crop)
    img, lbl = TR.apply(g["img"], g["lbl"], p, crop, g["mean"], g["std"])
    assert img.dtype == np.float32 and np.array_equal(img, g["out_img"])
    assert np.array_equal(lbl, g["out_lbl"])


def test_fixtures_cover_all_orders_and_flips():
    orders, flips = set(), set()
    for path in G9:
        g, seed, crop = _fixture(path)
        p = TR.sample_params(random.Random(seed), g["img"].shape[0], g["img"].shape[1], crop)
        orders.add(tuple(c for c, _ in p["ops"]))
        flips.add(p["flip"])
    assert len(orders) == 6 and flips == {True, False}


def test_oracle_enhance_equals_live_pillow():
    PIL = pytest.importorskip("PIL")
    from PIL import Image, ImageEnhance
    rs = np.random.RandomState(3)
    for trial in range(120):
        a = (rs.rand(7, 9, 3) * 256).astype(np.uint8)
        f = [0.0, 1.0, 0.5, 1.5][trial] if trial < 4 else float(rs.uniform(0.5, 1.5))
        im = Image.fromarray(a)
        assert np.array_equal(np.array(im.convert("L")), TR.luma(a))
        assert np.array_equal(np.array(ImageEnhance.Brightness(im).enhance(f)), TR.adjust_brightness(a, f))
        assert np.array_equal(np.array(ImageEnhance.Contrast(im).enhance(f)), TR.adjust_contrast(a, f))
        assert np.array_equal(np.array(ImageEnhance.Color(im).enhance(f)), TR.adjust_saturation(a, f))


def test_product_sampler_draws_like_the_reference():
    """utils.ext_transforms.ExtCompose.sample consumes `random` exactly as the oracle / reference do"""
    import utils
    et = utils.ext_transforms
    tf = et.ExtCompose([et.ExtRandomCrop(size=(20, 28)), et.ExtColorJitter(brightness=0.5, contrast=0.5, saturation=0.5),
                        et.ExtRandomHorizontalFlip(), et.ExtToTensor(), et.ExtNormalize(mean=MEAN, std=STD)])
    for seed in (1, 2, 3, 15, 99):
        random.seed(seed)
        got, crop = tf.sample(3, 28, 36)
        rng = random.Random(seed)
        ref = [TR.sample_params(rng, 28, 36, (20, 28)) for _ in range(3)]
        assert crop == (20, 28) and got == ref
    with pytest.raises(NotImplementedError):
        et.ExtCompose([et.ExtToTensor(), et.ExtRandomCrop(4), et.ExtNormalize(MEAN, STD)])
    with pytest.raises(NotImplementedError):
        et.ExtColorJitter(hue=0.1)
    with pytest.raises(TypeError):
        tf(torch.zeros(2, 28, 36, 3, dtype=torch.uint8), None)          # CPU tensor: no fallback


def _compose(crop):
    import utils
    et = utils.ext_transforms
    return et.ExtCompose([et.ExtRandomCrop(size=crop), et.ExtColorJitter(brightness=0.5, contrast=0.5, saturation=0.5),
                          et.ExtRandomHorizontalFlip(), et.ExtToTensor(), et.ExtNormalize(mean=MEAN, std=STD)])


@pytest.mark.gpu
@pytest.mark.parametrize("path", G9, ids=[os.path.basename(p)[7:-4] for p in G9])
def test_device_pipeline_matches_reference_fixture(path):
    g, seed, crop = _fixture(path)
    tf = _compose(crop)
    random.seed(seed)
    img, lbl = tf(torch.from_numpy(g["img"]).cuda(), torch.from_numpy(g["lbl"]).cuda())
    torch.cuda.synchronize()
    assert img.dtype == torch.float32 and lbl.dtype == torch.int64
    assert np.array_equal(img.cpu().numpy(), g["out_img"])                      # bit-exact
    assert np.array_equal(lbl.cpu().numpy(), g["out_lbl"].astype(np.int64))


@pytest.mark.gpu
def test_device_pipeline_batch_vs_oracle_edge_factors():
    """batch of frames, explicit parameters: factors exactly 0 / 1 / at the range ends, 0..3 ops, ragged last segment"""
    rs = np.random.RandomState(5)
    B, Hh, Ww, crop = 6, 70, 300, (37, 261)          # 261 = 256 + 5: two segments per row
    img = (rs.rand(B, Hh, Ww, 3) * 256).astype(np.uint8)
    img[1] = (img[1] * 0.2).astype(np.uint8)
    img[2] = 255 - (img[2] * 0.1).astype(np.uint8)
    lbl = (rs.rand(B, Hh, Ww) * 19).astype(np.uint8)
    params = [
        {"i": 0, "j": 0, "ops": [], "flip": False},
        {"i": 33, "j": 39, "ops": [(1, 1.5)], "flip": True},
        {"i": 5, "j": 7, "ops": [(2, 0.5), (0, 1.5), (1, 0.5)], "flip": True},
        {"i": 9, "j": 1, "ops": [(0, 1.0), (1, 1.0), (2, 1.0)], "flip": False},
        {"i": 20, "j": 30, "ops": [(1, 0.0), (2, 1.4999)], "flip": False},
        {"i": 1, "j": 2, "ops": [(0, 0.73), (2, 1.31), (1, 1.27)], "flip": True},
    ]
    tf = _compose(crop)
    out, olb = tf(torch.from_numpy(img).cuda(), torch.from_numpy(lbl).cuda(), params=params)
    torch.cuda.synchronize()
    for b in range(B):
        ri, rl = TR.apply(img[b], lbl[b], params[b], crop, MEAN, STD)
        assert np.array_equal(out[b].cpu().numpy(), ri), "image %d" % b
        assert np.array_equal(olb[b].cpu().numpy(), rl.astype(np.int64)), "label %d" % b


@pytest.mark.gpu
def test_device_pipeline_cityscapes_size():
    """1024 x 2048 frames -> 768 x 768 crops (the reference's training geometry), checked against the oracle"""
    rs = np.random.RandomState(7)
    B, Hh, Ww, crop = 3, 1024, 2048, (768, 768)
    base = (rs.rand(B, Hh // 8, Ww // 8, 3) * 256).astype(np.uint8)
    img = np.ascontiguousarray(np.repeat(np.repeat(base, 8, axis=1), 8, axis=2))
    img ^= (rs.rand(B, Hh, Ww, 3) * 8).astype(np.uint8)
    lbl = np.ascontiguousarray(np.repeat(np.repeat((rs.rand(B, Hh // 64, Ww // 64) * 19).astype(np.uint8), 64, 1), 64, 2))
    tf = _compose(crop)
    random.seed(123)
    out, olb = tf(torch.from_numpy(img).cuda(), torch.from_numpy(lbl).cuda())
    torch.cuda.synchronize()
    for b in range(B):
        ri, rl = TR.apply(img[b], lbl[b], tf.last_params[b], crop, MEAN, STD)
        assert np.array_equal(out[b].cpu().numpy(), ri)
        assert np.array_equal(olb[b].cpu().numpy(), rl.astype(np.int64))


# ---------------------------------------------------------------------------------------------------------------------
# metrics after the path: confusion matrix on the device
# ---------------------------------------------------------------------------------------------------------------------
def test_metrics_oracle_matches_reference_fixture():
    from oracle import metrics_ref as MR
    g = np.load(os.path.join(os.path.dirname(G9[0]), "g10_metrics.npz"))
    n = int(g["n"])
    hist = sum(MR.fast_hist(g["lt"][b].flatten(), g["lp"][b].flatten(), n) for b in range(g["lt"].shape[0]))
    assert np.array_equal(hist, g["hist"])
    r = MR.results(hist)
    assert r["Overall Acc"] == g["overall"] and r["Mean Acc"] == g["mean_acc"] and r["FreqW Acc"] == g["fw"]
    assert r["Mean IoU"] == g["miou"]
    assert np.array_equal(np.array([r["Class IoU"][k] for k in range(n)]), g["class_iou"], equal_nan=True)


@pytest.mark.gpu
def test_stream_metrics_on_device():
    import metrics
    from oracle import metrics_ref as MR
    g = np.load(os.path.join(os.path.dirname(G9[0]), "g10_metrics.npz"))
    n = int(g["n"])
    m = metrics.StreamSegMetrics(n)
    lt, lp = torch.from_numpy(g["lt"]).cuda(), torch.from_numpy(g["lp"]).cuda()
    m.update(lt[:2], lp[:2])
    m.update(lt[2:], lp[2:])
    r = m.get_results()
    assert np.array_equal(m.confusion_matrix.cpu().numpy(), g["hist"].astype(np.int64))          # exact counts
    assert r["Overall Acc"] == g["overall"] and r["Mean IoU"] == g["miou"] and r["Mean Acc"] == g["mean_acc"]
    assert r["FreqW Acc"] == g["fw"]
    assert np.array_equal(np.array([r["Class IoU"][k] for k in range(n)]), g["class_iou"], equal_nan=True)
    m.reset()
    assert int(m.confusion_matrix.sum()) == 0
    # full-size, odd element count, labels outside [0, n): 3 x 1023 x 2047 pixels vs numpy bincount
    rs = np.random.RandomState(2)
    a = rs.randint(-1, n + 2, (3, 1023, 2047)).astype(np.int64)
    a[a == n + 1] = 255
    b = rs.randint(0, n, (3, 1023, 2047)).astype(np.int64)
    m.update(torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda())
    assert np.array_equal(m.confusion_matrix.cpu().numpy(), MR.fast_hist(a.flatten(), b.flatten(), n))
    m.reset()
    ta, tb = torch.from_numpy(a).cuda().flatten(), torch.from_numpy(b).cuda().flatten()
    m.update(ta[1:1001], tb[1:1001])                       # views that start 8 bytes off a 16-byte boundary
    assert np.array_equal(m.confusion_matrix.cpu().numpy(), MR.fast_hist(a.flatten()[1:1001], b.flatten()[1:1001], n))
    with pytest.raises(TypeError):
        m.update(torch.from_numpy(a), torch.from_numpy(b))


# ---------------------------------------------------------------------------------------------------------------------
# OOD measures after the path: AUROC / AUPR / FPR95 by a device sort
# ---------------------------------------------------------------------------------------------------------------------
G11_CASES = ("cont", "ties", "sep", "inv", "r20")


@pytest.mark.parametrize("case", G11_CASES)
def test_ood_oracle_matches_reference_fixture(case):
    from oracle import ood_measures_ref as OR
    g = np.load(os.path.join(os.path.dirname(G9[0]), "g11_ood_measures.npz"))
    a, p, f = OR.get_measures(g[case + "_pos"], g[case + "_neg"])
    ref = g[case + "_res"]
    assert abs(a - ref[0]) <= 1e-12 and abs(p - ref[1]) <= 1e-12 and f == ref[2]


@pytest.mark.gpu
@pytest.mark.parametrize("case", G11_CASES)
def test_ood_measures_on_device_match_reference_fixture(case):
    import anom_utils
    g = np.load(os.path.join(os.path.dirname(G9[0]), "g11_ood_measures.npz"))
    a, p, f = anom_utils.get_measures(torch.from_numpy(g[case + "_pos"]).cuda(), torch.from_numpy(g[case + "_neg"]).cuda())
    ref = g[case + "_res"]
    assert abs(a - ref[0]) <= 1e-12, (a, ref[0])
    assert abs(p - ref[1]) <= 1e-12, (p, ref[1])
    assert f == ref[2], (f, ref[2])


@pytest.mark.gpu
def test_ood_measures_full_image_vs_oracle():
    """1024 x 2048 score map with a mask, several OOD labels, clipped plateaus (ties) -- vs the oracle"""
    import anom_utils
    from oracle import ood_measures_ref as OR
    rs = np.random.RandomState(4)
    Hh, Ww = 1024, 2048
    lab = rs.randint(0, 14, (Hh, Ww)).astype(np.int64)
    conf = rs.randn(Hh, Ww).astype(np.float32) + (lab >= 12) * 0.8
    conf = np.clip(conf, -1.5, 2.0).astype(np.float32)                  # plateaus at both ends
    conf[rs.rand(Hh, Ww) < 0.1] = 0.0
    mask = rs.rand(Hh, Ww) < 0.9
    got = anom_utils.eval_ood_measure(torch.from_numpy(conf).cuda(), torch.from_numpy(lab).cuda(), [12, 13],
                                      mask=torch.from_numpy(mask).cuda())
    ref = OR.eval_ood_measure(conf, lab, [12, 13], mask=mask)
    assert abs(got[0] - ref[0]) <= 1e-12 and abs(got[1] - ref[1]) <= 1e-11 and got[2] == ref[2], (got, ref)
    # only one class present -> None, like the reference
    assert anom_utils.eval_ood_measure(torch.from_numpy(conf).cuda(), torch.from_numpy(lab).cuda(), [99]) is None
    # sortedness / permutation property of the device sort at full size: same measures for a shuffled input
    perm = rs.permutation(Hh * Ww)
    got2 = anom_utils.eval_ood_measure(torch.from_numpy(conf.reshape(-1)[perm]).cuda(), torch.from_numpy(lab.reshape(-1)[perm]).cuda(),
                                       [12, 13], mask=torch.from_numpy(mask.reshape(-1)[perm]).cuda())
    assert got2[0] == got[0] and got2[2] == got[2] and abs(got2[1] - got[1]) <= 1e-13


@pytest.mark.gpu
def test_prototype_extraction_matches_numpy_recipe():
    """utils.extract_prototype vs the reference's commented recipe (test_embedding.py:413-425):
    np.mean(features[labels_true == c], axis=0) when the class covers more than 5 % of the image."""
    import json
    import utils
    rs = np.random.RandomState(8)
    Hh, Ww, Cc = 96, 160, 16
    feats = (rs.randn(1, Hh, Ww, Cc) * 2).astype(np.float32)
    lab = rs.randint(0, 16, (Hh, Ww)).astype(np.int64)
    lab[:40, :60] = 15                                   # ~16 % of the image
    lab[lab == 3] = 4                                    # class 3 absent
    lab[90:, 150:] = 7
    fd, ld = torch.from_numpy(feats).cuda(), torch.from_numpy(lab).cuda()
    got = utils.extract_prototype(fd, ld, 15)
    ref = np.mean(feats[0][lab == 15], axis=0)
    assert len(got) == Cc and np.allclose(np.array(got), ref, rtol=1e-5, atol=1e-6)
    assert utils.extract_prototype(fd, ld, 3) is None                       # absent
    frac7 = (lab == 7).mean()
    assert (utils.extract_prototype(fd, ld, 7) is None) == (frac7 <= 0.05)   # below the 5 % rule of the recipe
    proto = utils.mean_prototype(json.loads(json.dumps([got, got])))        # round trip through the json format
    assert np.allclose(proto, ref, rtol=1e-5, atol=1e-6)


# ---- Cityscapes label encoding (datasets/cityscapes.py:132-154), fixture G13 minted from the reference class --------
G13_CASES = ["none", "shipped", "train16", "one", "first"]


def _g13():
    return np.load(os.path.join(os.path.dirname(G9[0]), "g13_cityscapes_labels.npz"))


@pytest.mark.parametrize("case", G13_CASES)
def test_label_oracle_and_host_tables_match_reference_fixture(case):
    from oracle import cityscapes_ref as CR
    from datasets import Cityscapes
    g = _g13()
    unk = [int(v) for v in g["unk_" + case]] if case != "none" else None
    t, tt = CR.encode_target(g["raw"], unk)
    assert np.array_equal(t, g["target_" + case]) and np.array_equal(tt, g["true_" + case])
    lut, lut_true = Cityscapes.label_luts(unk)                       # host logic of the product: one composed table
    assert lut.dtype == np.uint8 and lut.shape == (256,)
    assert np.array_equal(lut[g["raw"]].astype(np.int64), g["target_" + case])
    assert np.array_equal(lut_true[g["raw"]].astype(np.int64), g["true_" + case])
    assert (lut[34:] == 255).all()


def test_label_class_table_and_eval_relabel_match_reference_fixture():
    from oracle import cityscapes_ref as CR
    from datasets import Cityscapes
    g = _g13()
    assert np.array_equal(Cityscapes.id_to_train_id, g["id_to_train_id"])
    assert np.array_equal(Cityscapes.train_id_to_color, g["train_id_to_color"])
    assert len(Cityscapes.classes) == 35 and Cityscapes.classes[26].name == "car" and Cityscapes.unknown_target == [14, 15]
    assert np.array_equal(CR.eval_relabel(g["target_shipped"]), g["eval_relabel_shipped"])
    lut2 = Cityscapes.eval_relabel_lut()
    assert np.array_equal(lut2[g["target_shipped"]].astype(np.int64), g["eval_relabel_shipped"])
    dec = Cityscapes.decode_target(torch.from_numpy(g["target_shipped"][0].copy()))
    assert np.array_equal(dec.numpy(), g["decoded_shipped0"])
    with pytest.raises(TypeError, match="no CPU fallback"):
        Cityscapes.encode_target(torch.zeros(4, dtype=torch.uint8))
    with pytest.raises(NotImplementedError):
        Cityscapes("/nonexistent")


@pytest.mark.gpu
def test_label_encode_on_device_matches_reference_fixture():
    from datasets import Cityscapes
    g = _g13()
    raw = torch.from_numpy(g["raw"]).cuda()
    saved = Cityscapes.unknown_target
    try:
        for case in G13_CASES:
            Cityscapes.unknown_target = [int(v) for v in g["unk_" + case]] if case != "none" else None
            t, tt = Cityscapes.encode_target(raw)
            assert t.dtype == torch.int64 and t.shape == raw.shape
            assert np.array_equal(t.cpu().numpy(), g["target_" + case]), case
            assert np.array_equal(tt.cpu().numpy(), g["true_" + case]), case
            # odd sizes / misaligned views
            v = raw.flatten()[3:3 + 1001]
            t2, _ = Cityscapes.encode_target(v)
            assert np.array_equal(t2.cpu().numpy(), g["target_" + case].reshape(-1)[3:3 + 1001])
    finally:
        Cityscapes.unknown_target = saved
    e, _ = Cityscapes.encode_target(torch.empty(0, dtype=torch.uint8, device="cuda"))
    assert e.numel() == 0


@pytest.mark.gpu
def test_device_pipeline_with_label_tables_equals_encode_after_transform():
    """Reference order: transform (crop / flip on raw ids), then encode_target; the fused kernel must give that."""
    import utils.ext_transforms as et
    from datasets import Cityscapes
    from oracle import cityscapes_ref as CR
    rng = np.random.default_rng(5)
    B, Hh, Ww, crop = 3, 70, 90, (48, 64)
    img = rng.integers(0, 256, size=(B, Hh, Ww, 3), dtype=np.uint8)
    lbl = rng.integers(0, 34, size=(B, Hh, Ww), dtype=np.uint8)
    tf = [et.ExtRandomCrop(size=crop), et.ExtColorJitter(0.5, 0.5, 0.5), et.ExtRandomHorizontalFlip(), et.ExtToTensor(),
          et.ExtNormalize(mean=MEAN, std=STD)]
    plain = et.ExtCompose(tf)
    fused = et.ExtCompose(tf, label_luts=Cityscapes.label_luts([13, 14, 15]))
    random.seed(77)
    xi, xl = plain(torch.from_numpy(img).cuda(), torch.from_numpy(lbl).cuda())
    yi, yl, yt = fused(torch.from_numpy(img).cuda(), torch.from_numpy(lbl).cuda(), params=plain.last_params)
    assert torch.equal(xi, yi)
    want, want_true = CR.encode_target(xl.cpu().numpy().astype(np.uint8), [13, 14, 15])
    assert yl.dtype == torch.int64 and np.array_equal(yl.cpu().numpy(), want)
    assert np.array_equal(yt.cpu().numpy(), want_true)
    assert {True, False} >= {p["flip"] for p in plain.last_params}
