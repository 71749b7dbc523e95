"""The reference's joint image/label transforms (utils/ext_transforms.py there) for BATCHES RESIDENT ON THE MI355X.

Same class names and constructor arguments as the reference, so the driver's transform block reads the same
(main_embedding.py:148-157):

    train_transform = et.ExtCompose([
        et.ExtRandomCrop(size=(768, 768)),
        et.ExtColorJitter(brightness=0.5, contrast=0.5, saturation=0.5),
        et.ExtRandomHorizontalFlip(),
        et.ExtToTensor(),
        et.ExtNormalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]),
    ])
    images, labels = train_transform(frames_u8, labels_u8)      # [B,H,W,3] / [B,H,W] uint8 CUDA tensors

The reference runs them per sample on PIL images in 16 DataLoader workers; at several hundred images/s per GPU that
host pipeline is the bottleneck.  Here the classes are specifications: ExtCompose draws the
random parameters on the host with the `random` module in the reference's order (crop i, j -> jitter factors ->
shuffle -> flip coin; ext_transforms.py:362-365, 483-499, 229) and runs two HIP kernels (dml_aug_contrast_sum,
dml_aug_apply) that reproduce Pillow's arithmetic bit for bit.  Output: float32 NCHW images and int64 labels (the
reference casts its uint8 labels to long right after the loader, main_embedding.py:463).
There is no CPU fallback: inputs must be CUDA tensors and libdmlnet_hip.so must be present.
"""
import ctypes as C
import numbers
import random

import torch

from dmlnet import _lib


class ExtRandomCrop(object):
    def __init__(self, size, padding=0, pad_if_needed=False):
        self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else tuple(int(s) for s in size)
        if padding or pad_if_needed:
            raise NotImplementedError("ExtRandomCrop padding is not used by the Cityscapes pipeline and not implemented")

    @staticmethod
    def get_params(img_hw, output_size):
        """(i, j, th, tw); draws nothing when the frame already has the crop size (ext_transforms.py:357-366)."""
        h, w = img_hw
        th, tw = output_size
        if w == tw and h == th:
            return 0, 0, h, w
        i = random.randint(0, h - th)
        j = random.randint(0, w - tw)
        return i, j, th, tw


class ExtColorJitter(object):
    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        self.brightness = self._check_input(brightness, "brightness")
        self.contrast = self._check_input(contrast, "contrast")
        self.saturation = self._check_input(saturation, "saturation")
        if hue:
            raise NotImplementedError("hue jitter is not used by the reference's drivers and not implemented")

    @staticmethod
    def _check_input(value, name):
        if isinstance(value, numbers.Number):
            if value < 0:
                raise ValueError("If {} is a single number, it must be non negative.".format(name))
            value = [max(1 - value, 0), 1 + value]
        elif isinstance(value, (tuple, list)) and len(value) == 2:
            if not 0 <= value[0] <= value[1]:
                raise ValueError("{} values should be between (0, inf)".format(name))
            value = list(value)
        else:
            raise TypeError("{} should be a single number or a list/tuple with lenght 2.".format(name))
        return None if value[0] == value[1] == 1 else value

    def get_params(self):
        """[(op code, factor), ...] in application order; draws as ext_transforms.py:483-499."""
        ops = []
        for code, rng in ((0, self.brightness), (1, self.contrast), (2, self.saturation)):
            if rng is not None:
                ops.append((code, random.uniform(rng[0], rng[1])))
        random.shuffle(ops)
        return ops


class ExtRandomHorizontalFlip(object):
    def __init__(self, p=0.5):
        self.p = p


class ExtToTensor(object):
    def __init__(self, normalize=True, target_type="uint8"):
        if not normalize:
            raise NotImplementedError("ExtToTensor(normalize=False) is not implemented")


class ExtNormalize(object):
    def __init__(self, mean, std):
        self.mean, self.std = [float(m) for m in mean], [float(s) for s in std]


class ExtCompose(object):
    """[ExtRandomCrop]? [ExtColorJitter]? [ExtRandomHorizontalFlip]? ExtToTensor ExtNormalize, fused on the device."""

    def __init__(self, transforms, label_luts=None):
        """label_luts: optional (lut, lut_true) uint8[256] tables (datasets.Cityscapes.label_luts) -- the dataset's
        encode_target folded into the crop kernel; __call__ then returns (images, labels, labels_true) like the
        reference dataset's __getitem__ (datasets/cityscapes.py:171-197)."""
        self.transforms = list(transforms)
        self.label_luts = label_luts
        self._dev_luts = None
        order = [ExtRandomCrop, ExtColorJitter, ExtRandomHorizontalFlip, ExtToTensor, ExtNormalize]
        pos = -1
        self.crop = self.jitter = self.flip = self.norm = None
        seen_tensor = False
        for t in self.transforms:
            if type(t) not in order or order.index(type(t)) <= pos:
                raise NotImplementedError("unsupported transform sequence for the device pipeline: %r" % (t,))
            pos = order.index(type(t))
            if isinstance(t, ExtRandomCrop): self.crop = t
            elif isinstance(t, ExtColorJitter): self.jitter = t
            elif isinstance(t, ExtRandomHorizontalFlip): self.flip = t
            elif isinstance(t, ExtToTensor): seen_tensor = True
            elif isinstance(t, ExtNormalize): self.norm = t
        if not seen_tensor or self.norm is None:
            raise NotImplementedError("the device pipeline ends with ExtToTensor, ExtNormalize")
        self.last_params = None

    def sample(self, B, H, W):
        """One (i, j, ops, flip) per image, drawn in the reference's per-sample order."""
        out = []
        for _ in range(B):
            if self.crop is not None:
                i, j, th, tw = ExtRandomCrop.get_params((H, W), self.crop.size)
            else:
                i, j, th, tw = 0, 0, H, W
            ops = self.jitter.get_params() if self.jitter is not None else []
            flip = self.flip is not None and random.random() < self.flip.p
            out.append({"i": i, "j": j, "ops": ops, "flip": bool(flip)})
        return out, (th, tw)

    def __call__(self, img, lbl, params=None):
        lib = _lib.load()
        if not (isinstance(img, torch.Tensor) and img.is_cuda and img.dtype == torch.uint8):
            raise TypeError("the device pipeline takes uint8 CUDA tensors [B,H,W,3] (there is no CPU fallback)")
        single = img.dim() == 3
        if single:
            img, lbl = img[None], (lbl[None] if lbl is not None else None)
        B, H, W, Cc = img.shape
        if Cc != 3 or not img.is_contiguous():
            raise ValueError("frames must be contiguous [B,H,W,3]")
        if lbl is not None and (lbl.dtype != torch.uint8 or tuple(lbl.shape) != (B, H, W) or not lbl.is_contiguous()
                                or lbl.device != img.device):
            raise ValueError("labels must be contiguous uint8 [B,H,W] on the frames' device")
        if params is None:
            params, (th, tw) = self.sample(B, H, W)
        else:
            th, tw = self.crop.size if self.crop is not None else (H, W)
        self.last_params = params
        arr = (_lib.AugSample * B)()
        for b, p in enumerate(params):
            a = arr[b]
            a.i, a.j, a.flip, a.n_ops = p["i"], p["j"], int(p["flip"]), len(p["ops"])
            for k, (code, f) in enumerate(p["ops"]):
                a.op[k], a.factor[k] = code, f
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        dev_params = host.to(img.device, non_blocking=False)
        lsum = torch.empty(B, dtype=torch.int32, device=img.device)
        out = torch.empty((B, 3, th, tw), dtype=torch.float32, device=img.device)
        olb = torch.empty((B, th, tw), dtype=torch.int64, device=img.device) if lbl is not None else None
        st = torch.cuda.current_stream(img.device).cuda_stream
        m, s = self.norm.mean, self.norm.std
        _lib.check(lib.dml_aug_contrast_sum(img.data_ptr(), dev_params.data_ptr(), lsum.data_ptr(), B, H, W, th, tw, st),
                   "dml_aug_contrast_sum")
        if self.label_luts is not None:
            if lbl is None:
                raise ValueError("label tables were given but no labels")
            if self._dev_luts is None or self._dev_luts[0].device != img.device:
                self._dev_luts = tuple(torch.as_tensor(t, dtype=torch.uint8).to(img.device).contiguous()
                                       for t in self.label_luts)
                if any(t.numel() != 256 for t in self._dev_luts):
                    raise ValueError("label tables have 256 entries")
            olt = torch.empty_like(olb)
            _lib.check(lib.dml_aug_apply_encoded(img.data_ptr(), lbl.data_ptr(), dev_params.data_ptr(), lsum.data_ptr(),
                                                 out.data_ptr(), olb.data_ptr(), B, H, W, th, tw, m[0], m[1], m[2], s[0],
                                                 s[1], s[2], self._dev_luts[0].data_ptr(), self._dev_luts[1].data_ptr(),
                                                 olt.data_ptr(), st), "dml_aug_apply_encoded")
            return (out[0], olb[0], olt[0]) if single else (out, olb, olt)
        _lib.check(lib.dml_aug_apply(img.data_ptr(), lbl.data_ptr() if lbl is not None else None, dev_params.data_ptr(),
                                     lsum.data_ptr(), out.data_ptr(), olb.data_ptr() if olb is not None else None, B, H, W,
                                     th, tw, m[0], m[1], m[2], s[0], s[1], s[2], st), "dml_aug_apply")
        if single:
            return out[0], (olb[0] if olb is not None else None)
        return out, olb
