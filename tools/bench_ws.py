"""Micro-benchmark (GPU box): the wave-specialised conv kernel (conv_ws_kernel) against the ring kernel (conv_igemm_dma_kernel)
on the train step's shapes, same process, same operands (tile-major weights), selected per launch through
DmlConvDesc.ws_min_tiles.  Forward with / without BN statistics, data gradient plain / with the fused BN-backward sums and
identity add.     python3 tools/bench_ws.py [fwd|dgrad|all]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc, PrepDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16
NEVER = 2 ** 31 - 1


def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def tiled(master, N, RS, Cc):
    w = torch.empty(N * RS * Cc, device="cuda", dtype=bf)
    arr = (PrepDesc * 1)(PrepDesc(master.data_ptr(), w.data_ptr(), None, N, RS, Cc, Cc, 1, 0))
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    assert lib.dml_prep_weights(tab.data_ptr(), 1, 1, st) == 0
    torch.cuda.synchronize()
    return w


SHAPES = [  # B, H, W, C, N, k, dil
    (16, 48, 48, 256, 256, 3, 1), (16, 48, 48, 1024, 256, 1, 1), (16, 48, 48, 256, 1024, 1, 1),
    (16, 192, 192, 320, 256, 3, 1), (16, 48, 48, 2048, 256, 3, 12), (16, 192, 192, 64, 256, 1, 1),
    (16, 96, 96, 128, 512, 1, 1), (16, 96, 96, 512, 128, 1, 1), (16, 48, 48, 512, 512, 3, 2), (16, 48, 48, 512, 2048, 1, 1)]
if os.environ.get("BENCH_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["BENCH_SHAPES"].split(";")]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for (B, H, W, Cc, N, k, dil) in SHAPES:
    pad = dil * (k // 2)
    M = B * H * W
    x = torch.randn(B, H, W, Cc, device="cuda").to(bf)
    w = tiled((torch.randn(N, k, k, Cc, device="cuda") * 0.05).contiguous(), N, k * k, Cc)
    y = torch.empty(B, H, W, N, device="cuda", dtype=bf)
    stats = torch.empty((M + 47) // 48 * N * 2, device="cuda")
    fl = 2.0 * M * N * k * k * Cc
    line = "B%d %dx%d C%d->N%d k%d d%d | " % (B, H, W, Cc, N, k, dil)
    if which in ("all", "fwd"):
        for use_stats in (False, True):
            for nm, mt in (("ring", NEVER), ("ws", 1)):
                d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr() if use_stats else None,
                             B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k,
                             stride=1, dil=dil, pad=pad, dtype=1, y_f32=0, accum=0, mode=0)
                d.w_tiled, d.ws_min_tiles = 1, mt
                t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
                line += "fwd%s %s %.1fus %.0fTF | " % ("+st" if use_stats else "", nm, t * 1e6, fl / t / 1e12)
    if which in ("all", "dgrad"):
        # data gradient of this conv: output B x H x W x Cc from dy B x H x W x N
        dy = torch.randn(B, H, W, N, device="cuda").to(bf)
        wt = tiled((torch.randn(Cc, k, k, N, device="cuda") * 0.05).contiguous(), Cc, k * k, N)
        gx = torch.zeros(B, H, W, Cc, device="cuda", dtype=bf)
        rdz = torch.randn(B, H, W, Cc, device="cuda").to(bf)
        ypre = torch.randn(B, H, W, Cc, device="cuda").to(bf)
        mask = torch.randint(0, 256, (M * Cc // 8,), dtype=torch.uint8, device="cuda")
        mean, invstd = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5
        part = torch.empty((M + 47) // 48 * Cc * 2, device="cuda")
        if Cc % 128 == 0:
            for lab, res_, bnr_ in (("plain", 0, 0), ("bnr", 0, 1), ("res+bnr", 1, 1)):
                for nm, mt in (("ring", NEVER), ("ws", 1)):
                    d = ConvDesc(x=dy.data_ptr(), w=wt.data_ptr(), y=gx.data_ptr(), bias=None, stats=None, B=B, Hi=H, Wi=W, C=N, ldx=N, Ho=H, Wo=W, N=Cc, ldy=Cc, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=1,
                                 y_f32=0, accum=0, mode=1)
                    d.w_tiled, d.ws_min_tiles = 1, mt
                    if res_:
                        d.res_dz, d.res_mask, d.res_ld = rdz.data_ptr(), mask.data_ptr(), Cc
                    if bnr_:
                        d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ypre.data_ptr(), mask.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                        d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), Cc, 1
                    t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
                    line += "dgrad %s %s %.1fus %.0fTF | " % (lab, nm, t * 1e6, fl / t / 1e12)
    print(line, flush=True)
