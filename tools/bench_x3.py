"""Micro-benchmark (GPU box): fp32 convolutions, exact fp32 MFMA vs the three-term bf16 split (DmlConvDesc.f32_split)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

SHAPES = [(16, 48, 48, 256, 256, 3, 1), (16, 48, 48, 1024, 256, 1, 1), (16, 48, 48, 256, 1024, 1, 1), (16, 192, 192, 320, 256, 3, 1),
          (16, 48, 48, 2048, 256, 3, 12), (16, 192, 192, 64, 256, 1, 1), (16, 96, 96, 128, 512, 1, 1), (16, 48, 48, 512, 512, 3, 2)]
for (B, H, W, Cc, N, k, dil) in SHAPES:
    pad = dil * (k // 2)
    x = torch.randn(B, H, W, Cc, device="cuda")
    w = torch.randn(N, k, k, Cc, device="cuda") * 0.05
    y = torch.empty(B, H, W, N, device="cuda")
    M = B * H * W
    stats = torch.empty((M + 63) // 64 * N * 2, device="cuda")
    fl = 2.0 * M * N * k * k * Cc
    line = "B%d %dx%d C%d->N%d k%d d%d | " % (B, H, W, Cc, N, k, dil)
    for split in (0, 1):
        d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=0, y_f32=0,
                     accum=0, mode=0)
        d.f32_split = split
        t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
        line += "%s %.1f us %.0f TF | " % ("split" if split else "exact", t * 1e6, fl / t / 1e12)
    print(line)

from dmlnet._lib import WgradDesc
print("weight gradient")
for (B, H, W, Cc, N, k, dil) in SHAPES:
    pad = dil * (k // 2)
    x = torch.randn(B, H, W, Cc, device="cuda")
    dy = torch.randn(B, H, W, N, device="cuda")
    dw = torch.zeros(N, k, k, Cc, device="cuda")
    ws = torch.empty(64 << 20, device="cuda")
    fl = 2.0 * B * H * W * N * k * k * Cc
    line = "B%d %dx%d C%d->N%d k%d d%d | " % (B, H, W, Cc, N, k, dil)
    for split in (0, 1):
        wg = WgradDesc(x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N,
                       R=k, S=k, stride=1, dil=dil, pad=pad, dtype=0, splitk=0, ws=ws.data_ptr(), ws_elems=ws.numel(), f32_split=split)
        t = timeit(lambda: lib.dml_conv_wgrad(C.byref(wg), st))
        line += "%s %.1f us %.0f TF | " % ("split" if split else "exact", t * 1e6, fl / t / 1e12)
    print(line)
