"""Micro-benchmark (GPU box): the 256 x 256 weight-gradient kernel with one / two K steps per barrier
(dml_debug_wgrad_depth), single launches with the split-K workspace at the step's shapes (16 images), rotating operand
sets."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import WgradDesc
lib = _lib.load()
lib.dml_debug_wgrad_depth.restype = C.c_int
lib.dml_debug_wgrad_depth.argtypes = [C.c_int]
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
B = 16


def timeit(fns, n=30):
    for f in fns: f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


SHAPES = [(48, 256, 256, 3, 1), (48, 1024, 256, 1, 1), (48, 256, 1024, 1, 1), (48, 512, 512, 3, 2), (48, 2048, 256, 3, 12),
          (192, 320, 256, 3, 1), (48, 512, 2048, 1, 1), (96, 128, 128, 3, 1), (96, 128, 512, 1, 1)]
ws = torch.empty(40 << 20, device="cuda")
for (H, Cc, N, k, dil) in SHAPES:
    M = B * H * H
    fl = 2.0 * M * N * Cc * k * k
    nsets = max(2, min(6, int(1.0e9 // (M * (Cc + N) * 2))))
    sets = [(torch.randn(B, H, H, Cc, device="cuda").to(bf), torch.randn(B, H, H, N, device="cuda").to(bf)) for _ in range(nsets)]
    dw = torch.zeros(N, k, k, Cc, device="cuda")
    descs = [WgradDesc(x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), B=B, Hi=H, Wi=H, C=Cc, ldx=Cc, Ho=H, Wo=H, N=N, ldy=N,
                       R=k, S=k, stride=1, dil=dil, pad=dil * (k // 2), dtype=1, splitk=0, Cm=0, ws=ws.data_ptr(),
                       ws_elems=ws.numel()) for (x, dy) in sets]
    res = []
    for depth in (1, 2):
        lib.dml_debug_wgrad_depth(depth)
        res.append(timeit([lambda d=d: lib.dml_conv_wgrad(C.byref(d), st) for d in descs]))
    print("%dx%d k%d d%d C%d -> N%d | wgrad (+fold) depth 1: %.1f us %.0f TF | depth 2: %.1f us %.0f TF" % (
        H, H, k, dil, Cc, N, res[0] * 1e6, fl / res[0] / 1e12, res[1] * 1e6, fl / res[1] / 1e12), flush=True)
