"""Micro-benchmark (GPU box): store-heavy 1x1 launches with plain / non-temporal epilogue stores -- run once per value of
DML_CONV_NT (0 | 1 | 2), rotating operand sets.  profiles/r02_conv_nt_stores.txt."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
B = 16
def timeit(fns, n=40):
    for f in fns: f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for (H, Cc, N) in ((192, 64, 256), (96, 128, 512), (48, 256, 1024), (48, 1024, 256)):
    M = B * H * H
    w = (torch.randn(N, 1, 1, Cc, device="cuda") * 0.05).to(bf)
    nsets = max(2, min(8, int(1.2e9 // (M * (Cc + N) * 2))))
    sets = [(torch.randn(B, H, H, Cc, device="cuda").to(bf), torch.empty(B, H, H, N, device="cuda", dtype=bf), torch.empty((M + 63) // 64 * N * 2, device="cuda")) for _ in range(nsets)]
    for use_stats in (0, 1):
        descs = [ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=s.data_ptr() if use_stats else None, B=B, Hi=H, Wi=H, C=Cc, ldx=Cc, Ho=H, Wo=H, N=N, ldy=N, R=1, S=1, stride=1, dil=1, pad=0, dtype=1, y_f32=0, accum=0, mode=0) for (x, y, s) in sets]
        t = timeit([lambda d=d: lib.dml_conv_igemm(C.byref(d), st) for d in descs])
        print("DML_CONV_NT=%s %dx%d K=%d->N=%d stats=%d: %.1f us" % (os.environ.get("DML_CONV_NT", "1"), H, H, Cc, N, use_stats, t * 1e6), flush=True)
