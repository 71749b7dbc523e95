"""Cityscapes label space on the device (reference datasets/cityscapes.py).

Keeps the reference's class attributes (`classes`, `train_id_to_color`, `id_to_train_id`, `unknown_target`) and the
`encode_target` / `decode_target` names (datasets/cityscapes.py:65-71,132-160).  The reference encodes every label PNG
with numpy in the loader workers: a table lookup, then one full-image pass per unknown class and per shifted id.  All of
that is a pointwise function of the raw id, so it is folded here into ONE 256-entry table (`label_luts`, host) that the
augmentation kernel applies while it crops (`utils.ext_transforms.ExtCompose(..., label_lut=...)` ->
dml_aug_apply_encoded) or `encode_target` applies to a whole uint8 CUDA tensor (dml_label_encode).  There is no CPU
path: `encode_target` raises on anything but a CUDA uint8 tensor.
The file-system side (`__init__`, `__getitem__`: PNG decoding) is out of scope (storage / IO).
"""
from collections import namedtuple

import numpy as np
import torch

from dmlnet import _lib

CityscapesClass = namedtuple("CityscapesClass", ["name", "id", "train_id", "category", "category_id", "has_instances",
                                                 "ignore_in_eval", "color"])

# the public cityscapesScripts label table: name, id, train id, category, category id, instances, ignored in eval, colour
_TABLE = """unlabeled|0|255|void|0|0|1|0,0,0
ego vehicle|1|255|void|0|0|1|0,0,0
rectification border|2|255|void|0|0|1|0,0,0
out of roi|3|255|void|0|0|1|0,0,0
static|4|255|void|0|0|1|0,0,0
dynamic|5|255|void|0|0|1|111,74,0
ground|6|255|void|0|0|1|81,0,81
road|7|0|flat|1|0|0|128,64,128
sidewalk|8|1|flat|1|0|0|244,35,232
parking|9|255|flat|1|0|1|250,170,160
rail track|10|255|flat|1|0|1|230,150,140
building|11|2|construction|2|0|0|70,70,70
wall|12|3|construction|2|0|0|102,102,156
fence|13|4|construction|2|0|0|190,153,153
guard rail|14|255|construction|2|0|1|180,165,180
bridge|15|255|construction|2|0|1|150,100,100
tunnel|16|255|construction|2|0|1|150,120,90
pole|17|5|object|3|0|0|153,153,153
polegroup|18|255|object|3|0|1|153,153,153
traffic light|19|6|object|3|0|0|250,170,30
traffic sign|20|7|object|3|0|0|220,220,0
vegetation|21|8|nature|4|0|0|107,142,35
terrain|22|9|nature|4|0|0|152,251,152
sky|23|10|sky|5|0|0|70,130,180
person|24|11|human|6|1|0|220,20,60
rider|25|12|human|6|1|0|255,0,0
car|26|13|vehicle|7|1|0|0,0,142
truck|27|14|vehicle|7|1|0|0,0,70
bus|28|15|vehicle|7|1|0|0,60,100
caravan|29|255|vehicle|7|1|1|0,0,90
trailer|30|255|vehicle|7|1|1|0,0,110
train|31|16|vehicle|7|1|0|0,80,100
motorcycle|32|17|vehicle|7|1|0|0,0,230
bicycle|33|18|vehicle|7|1|0|119,11,32
license plate|-1|255|vehicle|7|0|1|0,0,142"""


def _parse(line):
    n, i, t, cat, ci, inst, ign, col = line.split("|")
    return CityscapesClass(n, int(i), int(t), cat, int(ci), inst == "1", ign == "1", tuple(int(v) for v in col.split(",")))


class Cityscapes(object):
    CityscapesClass = CityscapesClass
    classes = [_parse(l) for l in _TABLE.splitlines()]
    train_id_to_color = np.array([c.color for c in classes if c.train_id not in (-1, 255)] + [(0, 0, 0)])
    id_to_train_id = np.array([c.train_id for c in classes])
    unknown_target = [14, 15]                      # as shipped (datasets/cityscapes.py:71); README: [13, 14, 15] to train
    _lut_cache = {}

    def __init__(self, *args, **kwargs):
        raise NotImplementedError("the PNG-reading dataset is outside this build's scope (SURVEY.md section 8); "
                                  "use Cityscapes.encode_target / label_luts with frames already in HBM")

    @classmethod
    def label_luts(cls, unknown_target="class"):
        """(lut, lut_true): uint8[256] tables, lut[raw id] = encode_target's `target`, lut_true = its `target_true`.
        Raw values a label PNG cannot hold meaningfully (34..255) map like numpy's indexing error would not: 255."""
        if isinstance(unknown_target, str):
            unknown_target = cls.unknown_target
        train = np.full(256, 255, dtype=np.int64)
        train[:34] = cls.id_to_train_id[:34]
        enc = train.copy()
        if unknown_target is not None:
            # compose the reference's sequential passes per id instead of per pixel: after removing `cont` classes the
            # next unknown class h sits at h - cont; it is parked, everything above moves down by one
            ids = np.arange(19)
            alive = np.ones(19, dtype=bool)
            for cont, h in enumerate(unknown_target):
                hit = alive & (ids == h - cont)
                alive &= ~hit
                ids = np.where(alive & (ids > h - cont), ids - 1, ids)
            final = np.where(alive, ids, 255)
            sel = enc < 19
            enc[sel] = final[enc[sel]]
        return enc.astype(np.uint8), train.astype(np.uint8)

    @classmethod
    def device_luts(cls, device, unknown_target="class"):
        key = (str(device), None if unknown_target is None else (unknown_target if isinstance(unknown_target, str)
                                                                  else tuple(unknown_target)),
               tuple(cls.unknown_target) if cls.unknown_target is not None else None)
        if key not in cls._lut_cache:
            a, b = cls.label_luts(unknown_target)
            cls._lut_cache[key] = (torch.from_numpy(a).to(device), torch.from_numpy(b).to(device))
        return cls._lut_cache[key]

    @classmethod
    def encode_target(cls, target):
        """uint8 CUDA tensor of raw label ids (any shape) -> (target, target_true) int64 CUDA tensors."""
        if not (isinstance(target, torch.Tensor) and target.is_cuda and target.dtype == torch.uint8):
            raise TypeError("encode_target takes a uint8 CUDA tensor of raw label ids (there is no CPU fallback)")
        lib = _lib.load()
        t = target.contiguous()
        lut, lut_true = cls.device_luts(t.device)
        out = torch.empty(t.shape, dtype=torch.int64, device=t.device)
        out_true = torch.empty_like(out)
        st = torch.cuda.current_stream(t.device).cuda_stream
        _lib.check(lib.dml_label_encode(t.data_ptr(), t.numel(), lut.data_ptr(), lut_true.data_ptr(), out.data_ptr(),
                                        out_true.data_ptr(), st), "dml_label_encode")
        return out, out_true

    @classmethod
    def decode_target(cls, target):
        """train ids -> RGB for visualisation (datasets/cityscapes.py:156-160; 255 -> the 20th, black, entry)."""
        t = torch.as_tensor(target).clone()
        t[t == 255] = 19
        return torch.as_tensor(cls.train_id_to_color, device=t.device)[t.long()]

    @staticmethod
    def eval_relabel_lut(held_out=13, new_id=16):
        """Table for the evaluation-time relabel of test_embedding.py:448-451 (held-out class -> new_id, every id above
        it moves down by one, and the 254 that 255 became goes back to 255), composable with label_luts: lut2[lut[raw]]."""
        t = np.arange(256, dtype=np.int64)
        out = np.where(t > held_out, t - 1, t)
        out[held_out] = new_id
        out[out == 254] = 255
        return out.astype(np.uint8)
