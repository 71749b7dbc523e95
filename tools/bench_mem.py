"""GPU box: what the chip sustains for pure writes / pure reads / copy -- the ceilings behind the HBM-bound kernels'
roofline fractions (MI355X_MICROARCH.md quotes 6.29 TB/s for a float4 copy, i.e. 3.15 read + 3.15 write)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
out = {}
for mb in (302, 1208):
    n = mb * (1 << 20) // 4
    a = torch.empty(n, device="cuda"); b = torch.empty(n, device="cuda")
    def t(fn, it=20):
        for _ in range(3): fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / it
    w = t(lambda: lib.dml_fill_f32(a.data_ptr(), n, 1.0, st))
    w2 = t(lambda: a.fill_(2.0))
    c = t(lambda: b.copy_(a))
    r = t(lambda: a.sum())
    out["%d MB" % mb] = {"fill (dml_fill_f32) TB/s": 4 * n / w / 1e12, "fill (torch) TB/s": 4 * n / w2 / 1e12,
                         "copy TB/s (read+write)": 8 * n / c / 1e12, "sum (read only) TB/s": 4 * n / r / 1e12}
print(json.dumps(out, indent=1))
