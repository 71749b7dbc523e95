"""Micro-benchmark (GPU box): the short-K launches of the bottlenecks (16 images) -- forward with BN statistics, data
gradient with the fused BN-backward sums -- on rotating buffer sets (operands come from HBM / MALL as in the step), with the
persistent short-K kernel off and on (dml_debug_conv_persist).  (The 96-row-tile comparison kept in
profiles/r02_bm96_variant.txt was made with this script on a build that had that tile.)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc
lib = _lib.load()
lib.dml_debug_conv_persist.restype = C.c_int
lib.dml_debug_conv_persist.argtypes = [C.c_int]
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


tws = torch.empty(512 * 128 * 128, device="cuda"); tcnt = torch.zeros(128, dtype=torch.int32, device="cuda")
SHAPES = [(48, 1024, 256, 1), (48, 256, 1024, 1), (48, 512, 2048, 1), (96, 128, 512, 1), (96, 512, 128, 1),
          (192, 64, 256, 1), (192, 256, 64, 1), (192, 64, 64, 3), (96, 128, 128, 3)]

for (H, Cc, N, k) in SHAPES:
    W = H
    M = B * H * W
    fl = 2.0 * M * N * Cc * k * k
    w = (torch.randn(N, k, k, Cc, device="cuda") * 0.05).to(bf)
    per_set = M * (Cc + 3 * N) * 2
    nsets = max(2, min(8, int(1.2e9 // per_set)))
    sets = []
    for s in range(nsets):
        x = torch.randn(B, H, W, Cc, device="cuda").to(bf)
        y = torch.empty(B, H, W, N, device="cuda", dtype=bf)
        stats = torch.empty((M + 47) // 48 * N * 2, device="cuda")
        ybn = torch.randn(M, N, device="cuda").to(bf)
        bits = torch.randint(0, 256, (M * N // 8,), device="cuda", dtype=torch.uint8)
        part = torch.empty((M + 47) // 48 * N * 2, device="cuda")
        sets.append((x, y, stats, ybn, bits, part))
    mean, invstd = torch.randn(N, device="cuda") * 0.2, torch.rand(N, device="cuda") + 0.5
    line = "%dx%d k%d K=%d -> N=%d | " % (H, W, k, Cc * k * k, N)
    for mode, use_stats, bnr, accum, nm in ((0, 1, 0, 0, "fwd+stats"), (1, 0, 0, 0, "dgrad"), (1, 0, 1, 0, "dgrad+bnr"),
                                            (1, 0, 1, 1, "dgrad+bnr+accum")):
        descs = []
        for (x, y, stats, ybn, bits, part) in sets:
            d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr() if use_stats else None,
                         pre_scale=None, pre_shift=None, B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k,
                         stride=1, dil=1, pad=k // 2, dtype=1, y_f32=0, accum=accum, mode=mode, pre_relu=0)
            if bnr:
                if M // 64 > 4096 or N <= 32:
                    continue
                d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), bits.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), N, 1
            d.tail_ws, d.tail_ws_elems, d.tail_counters, d.tail_counters_len = tws.data_ptr(), tws.numel(), tcnt.data_ptr(), 128
            descs.append(d)
        if not descs:
            continue
        res = []
        for on in (0, 1):
            lib.dml_debug_conv_persist(24 if on else 0)
            t = timeit([lambda d=d: lib.dml_conv_igemm(C.byref(d), st) for d in descs])
            res.append(t)
        line += "%s %.1f -> %.1f us (%.0f -> %.0f TF) | " % (nm, res[0] * 1e6, res[1] * 1e6, fl / res[0] / 1e12, fl / res[1] / 1e12)
    print(line, flush=True)
