"""Micro-benchmark (GPU box): the four 1x1 launches of a layer3 bottleneck (48 x 48 maps, 16 images) -- forward with BN
statistics, data gradient plain / with the fused BN-backward sums -- on cache-hot operands (same buffers every launch)
and on rotating buffer sets (operands come from HBM, as in the step)."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
B, H, W = 16, 48, 48
M = B * H * W
NSETS = int(os.environ.get("NSETS", "8"))


def timeit(fns, n=40):
    for f in fns: f()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(n): fns[i % len(fns)]()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


tws = torch.empty(512 * 128 * 128, device="cuda"); tcnt = torch.zeros(128, dtype=torch.int32, device="cuda")
for (Cc, N) in ((1024, 256), (256, 1024)):
    fl = 2.0 * M * N * Cc
    w = (torch.randn(N, 1, 1, Cc, device="cuda") * 0.05).to(bf)
    sets = []
    for s in range(NSETS):
        x = torch.randn(B, H, W, Cc, device="cuda").to(bf)
        y = torch.empty(B, H, W, N, device="cuda", dtype=bf)
        stats = torch.empty((M + 63) // 64 * N * 2, device="cuda")
        ybn = torch.randn(M, N, device="cuda").to(bf)
        bits = torch.randint(0, 256, (M * N // 8,), device="cuda", dtype=torch.uint8)
        part = torch.empty((M + 63) // 64 * N * 2, device="cuda")
        sets.append((x, y, stats, ybn, bits, part))
    mean, invstd = torch.randn(N, device="cuda") * 0.2, torch.rand(N, device="cuda") + 0.5
    line = "K=%d -> N=%d | " % (Cc, N)
    for mode, use_stats, bnr, tail, nm in ((0, 0, 0, 0, "fwd"), (0, 1, 0, 0, "fwd+stats"), (0, 1, 0, 1, "fwd+stats+tail"),
                                           (1, 0, 0, 0, "dgrad"), (1, 0, 1, 0, "dgrad+bnr"), (1, 0, 1, 1, "dgrad+bnr+tail")):
        descs = []
        for (x, y, stats, ybn, bits, part) in sets:
            d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr() if use_stats else None,
                         pre_scale=None, pre_shift=None, B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=1, S=1,
                         stride=1, dil=1, pad=0, dtype=1, y_f32=0, accum=0, mode=mode, pre_relu=0)
            if bnr:
                d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), bits.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), N, 1
            if tail:
                d.tail_ws, d.tail_ws_elems, d.tail_counters, d.tail_counters_len = tws.data_ptr(), tws.numel(), tcnt.data_ptr(), 128
            descs.append(d)
        hot = timeit([lambda d=descs[0]: lib.dml_conv_igemm(C.byref(d), st)])
        cold = timeit([lambda d=d: lib.dml_conv_igemm(C.byref(d), st) for d in descs])
        line += "%s hot %.1f / rot %.1f us (%.0f TF) | " % (nm, hot * 1e6, cold * 1e6, fl / cold / 1e12)
    print(line)
