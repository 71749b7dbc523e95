"""CPU restatement of the reference's Cityscapes TRAIN input pipeline -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(utils/ext_transforms.py -> dml_aug_* in libdmlnet_hip.so) never does.

What it restates ("step before the path"):
  main_embedding.py:148-157   ExtRandomCrop(768) -> ExtColorJitter(brightness=.5, contrast=.5, saturation=.5)
                              -> ExtRandomHorizontalFlip() -> ExtToTensor() -> ExtNormalize(mean, std)
  utils/ext_transforms.py:357-366,368-393   random crop parameters / crop
  utils/ext_transforms.py:469-504           jitter factors and random order
  utils/ext_transforms.py:222-230           horizontal flip with p = 0.5
  utils/ext_transforms.py:282-293,313-322   uint8 HWC -> float CHW / 255, (x - mean) / std

The per-pixel arithmetic is NOT in /root/reference: ext_transforms.py calls torchvision.transforms.functional
(pin torchvision==0.6.0, requirements.txt:133), which calls Pillow (pin Pillow==8.0.1, requirements.txt:80).
Their published algorithm, restated here:
  F.adjust_brightness(img, f) = ImageEnhance.Brightness(img).enhance(f) = Image.blend(black, img, f)
  F.adjust_contrast(img, f)   = Image.blend(gray(m), img, f),  m = int(mean(img.convert("L")) + 0.5)
  F.adjust_saturation(img, f) = Image.blend(img.convert("L") as RGB, img, f)
  convert("L"):  L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16
  Image.blend(a, b, f): float32  t = a + f * (b - a);  0 <= f <= 1: (uint8) t (truncation);
                        otherwise clip t to [0, 255] first.
Parity pin: tests/golden/g9_*.npz are outputs of the reference's own ext_transforms classes driven through a
torchvision shim over the real Pillow in the authoring container (tests/tools/mint_golden_aug.py; Pillow 12.2.0 there),
and tests/test_oracle.py re-checks the three enhance functions against the live Pillow when it is importable.
"""
import numpy as np

ORDER_B, ORDER_C, ORDER_S = 0, 1, 2          # jitter op codes


def luma(rgb):
    a = rgb.astype(np.uint32)
    return ((a[..., 0] * 19595 + a[..., 1] * 38470 + a[..., 2] * 7471 + 0x8000) >> 16).astype(np.uint8)


def blend(degenerate, img, factor):
    f32 = np.float32(factor)
    d, x = degenerate.astype(np.float32), img.astype(np.float32)
    t = d + f32 * (x - d)                     # float32, multiply then add (no fused multiply-add)
    if 0.0 <= float(f32) <= 1.0:
        return t.astype(np.uint8)
    return np.clip(t, 0, 255).astype(np.uint8)


def adjust_brightness(img, f):
    return blend(np.zeros_like(img), img, f)


def contrast_mean(img):
    lum = luma(img)
    return int(float(lum.astype(np.uint64).sum()) / float(lum.size) + 0.5)


def adjust_contrast(img, f):
    return blend(np.full_like(img, contrast_mean(img)), img, f)


def adjust_saturation(img, f):
    return blend(np.repeat(luma(img)[..., None], 3, axis=-1), img, f)


_ADJ = {ORDER_B: adjust_brightness, ORDER_C: adjust_contrast, ORDER_S: adjust_saturation}


def sample_params(rng, H, W, crop, brightness=0.5, contrast=0.5, saturation=0.5, p_flip=0.5):
    """Draws from `rng` (a random.Random / the random module) in the reference's order:
    crop i, j (ext_transforms.py:362-365; none if the image already has the crop size), then the jitter factors
    b, c, s and the shuffle (:483-499), then the flip coin (:229)."""
    th, tw = crop
    if W == tw and H == th:
        i, j = 0, 0
    else:
        i = rng.randint(0, H - th)
        j = rng.randint(0, W - tw)
    ops = []
    for code, amount in ((ORDER_B, brightness), (ORDER_C, contrast), (ORDER_S, saturation)):
        if amount:
            ops.append((code, rng.uniform(max(0.0, 1.0 - amount), 1.0 + amount)))
    rng.shuffle(ops)
    flip = rng.random() < p_flip
    return {"i": i, "j": j, "ops": ops, "flip": bool(flip)}


def apply(img_u8, lbl_u8, params, crop, mean, std):
    """img_u8 [H,W,3], lbl_u8 [H,W] -> (float32 [3,th,tw], int64-able uint8 [th,tw])."""
    th, tw = crop
    i, j = params["i"], params["j"]
    x = img_u8[i:i + th, j:j + tw]
    y = lbl_u8[i:i + th, j:j + tw]
    for code, f in params["ops"]:
        x = _ADJ[code](x, f)
    if params["flip"]:
        x, y = x[:, ::-1], y[:, ::-1]
    t = x.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)            # F.to_tensor: .float().div(255)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return (t - m) / s, np.ascontiguousarray(y)                                # F.normalize: sub_ then div_
