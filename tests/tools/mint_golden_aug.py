"""Mint tests/golden/g9_aug_*.npz from the REFERENCE's own input-pipeline classes (authoring container only).

/root/reference/DeepLabV3Plus-Pytorch/utils/ext_transforms.py is loaded by file path and run unchanged; it
needs torchvision.transforms.functional (pin torchvision==0.6.0), which is not installed here, so a shim module
forwards the few functions it calls to the real Pillow exactly as torchvision 0.6.0 does (functional.py:
crop -> Image.crop, hflip -> transpose(FLIP_LEFT_RIGHT), adjust_* -> ImageEnhance.*.enhance, to_tensor ->
byte HWC -> float CHW / 255, normalize -> sub_/div_).  Outputs are data only: input image / label, the seed
given to `random`, and the tensors the reference pipeline returned.
"""
import importlib.util, os, random, sys, types
import numpy as np
import torch
from PIL import Image, ImageEnhance

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference/DeepLabV3Plus-Pytorch/utils/ext_transforms.py"

F = types.ModuleType("torchvision.transforms.functional")
F.crop = lambda img, i, j, h, w: img.crop((j, i, j + w, i + h))
F.hflip = lambda img: img.transpose(Image.FLIP_LEFT_RIGHT)
F.adjust_brightness = lambda img, f: ImageEnhance.Brightness(img).enhance(f)
F.adjust_contrast = lambda img, f: ImageEnhance.Contrast(img).enhance(f)
F.adjust_saturation = lambda img, f: ImageEnhance.Color(img).enhance(f)


def _to_tensor(pic):
    a = np.array(pic, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    return torch.from_numpy(a).permute(2, 0, 1).contiguous().float().div(255)


def _normalize(t, mean, std, inplace=False):
    t = t.clone()
    m = torch.as_tensor(mean, dtype=t.dtype)
    s = torch.as_tensor(std, dtype=t.dtype)
    return t.sub_(m[:, None, None]).div_(s[:, None, None])


F.to_tensor, F.normalize = _to_tensor, _normalize
tv = types.ModuleType("torchvision")
tvt = types.ModuleType("torchvision.transforms")
tv.transforms, tvt.functional = tvt, F
sys.modules.update({"torchvision": tv, "torchvision.transforms": tvt, "torchvision.transforms.functional": F})
sys.dont_write_bytecode = True
spec = importlib.util.spec_from_file_location("ref_ext_transforms", REF)
et = importlib.util.module_from_spec(spec)
spec.loader.exec_module(et)

MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]


def run(seed, H, W, crop, name, gray=False):
    rs = np.random.RandomState(seed)
    img = (rs.rand(H, W, 3) * 256).astype(np.uint8)
    if gray:
        img[:] = img[..., :1]
    # smooth-ish content so that the contrast mean is not always ~127
    img = (img.astype(np.float32) * rs.uniform(0.3, 1.0)).astype(np.uint8)
    lbl = (rs.rand(H, W) * 19).astype(np.uint8)
    lbl[rs.rand(H, W) < 0.05] = 255
    tf = et.ExtCompose([                                  # main_embedding.py:148-157 of the reference
        et.ExtRandomCrop(size=crop),
        et.ExtColorJitter(brightness=0.5, contrast=0.5, saturation=0.5),
        et.ExtRandomHorizontalFlip(),
        et.ExtToTensor(),
        et.ExtNormalize(mean=MEAN, std=STD),
    ])
    random.seed(seed)
    t, l = tf(Image.fromarray(img), Image.fromarray(lbl))
    out = os.path.join(ROOT, "tests", "golden", "g9_aug_%s.npz" % name)
    np.savez_compressed(out, seed=seed, img=img, lbl=lbl, crop=np.array(crop), out_img=t.numpy(), out_lbl=l.numpy(),
                        mean=np.array(MEAN), std=np.array(STD))
    print(name, t.shape, l.shape, l.dtype, float(t.mean()))


if __name__ == "__main__":
    run(11, 40, 56, (24, 32), "a")
    run(12, 33, 47, (24, 32), "b")
    run(13, 24, 32, (24, 32), "fullsize")       # crop == image: no crop draw (ext_transforms.py:359-360)
    run(14, 30, 60, (16, 48), "gray", gray=True)
    for k in range(15, 23):                      # more seeds: every op order and both flip outcomes
        run(k, 28, 36, (20, 28), "s%d" % k)
