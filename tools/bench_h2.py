"""Micro-benchmark (GPU box): fp32 convolutions of the train step's shapes with the products (a) exact (v_mfma_f32_16x16x4_f32),
(b) three-term bf16 split (f32_split = 1), (c) two fp16 planes per operand (f32_split = 2: conv_ws_kernel<.., 2>,
conv_wgrad_h2_kernel).  Forward with BN statistics, data gradient, weight gradient.  TFLOP/s are fp32-equivalent.
    python3 tools/bench_h2.py [fwd|dgrad|wgrad|all] [only=<variant>]      (only=...: one variant, for rocprofv3 --pmc passes)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc, WgradDesc
lib = _lib.load()
torch.manual_seed(int(os.environ.get("BENCH_SEED", "0")))      # (same inputs in every process: BENCH_DUMP digests are comparable)
st = torch.cuda.current_stream().cuda_stream
import hashlib
DUMP = os.environ.get("BENCH_DUMP")        # file: one line per launch with sha256 of every output tensor (bit-equality checks between builds / configurations)


def digest(tag, *tensors):
    if not DUMP:
        return
    torch.cuda.synchronize()
    with open(DUMP, "a") as fh:
        fh.write(tag + " " + " ".join(hashlib.sha256(t.detach().cpu().numpy().tobytes()).hexdigest()[:16] for t in tensors) + "\n")


def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


def planes(t2d, layout):
    rows, Cc = t2d.shape
    pl = torch.empty((2, rows * Cc), device="cuda", dtype=torch.float16)
    work = torch.zeros(1025, device="cuda")
    assert lib.dml_h2_split(t2d.data_ptr(), rows, Cc, Cc, pl.data_ptr(), rows * Cc, Cc, layout, work.data_ptr(), 0, st) == 0
    return pl, work


SHAPES = [  # B, H, W, C, N, k, dil
    (16, 48, 48, 256, 256, 3, 1), (16, 48, 48, 1024, 256, 1, 1), (16, 48, 48, 256, 1024, 1, 1),
    (16, 192, 192, 320, 256, 3, 1), (16, 48, 48, 2048, 256, 3, 12), (16, 48, 48, 512, 512, 3, 2), (16, 96, 96, 128, 512, 1, 1)]
if os.environ.get("BENCH_SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["BENCH_SHAPES"].split(";")]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
only = next((a[5:] for a in sys.argv[2:] if a.startswith("only=")), None)
VARIANTS = [v for v in (("exact", 0), ("x3", 1), ("h2", 2)) if only in (None, v[0])]
for (B, H, W, Cc, N, k, dil) in SHAPES:
    pad = dil * (k // 2)
    M = B * H * W
    x = torch.randn(B, H, W, Cc, device="cuda")
    w = (torch.randn(N, k, k, Cc, device="cuda") * 0.05).contiguous()
    wt = w.permute(3, 1, 2, 0).contiguous()
    dy = torch.randn(B, H, W, N, device="cuda") * 1e-3
    y = torch.empty(B, H, W, N, device="cuda")
    gx = torch.empty(B, H, W, Cc, device="cuda")
    stats = torch.empty((M + 47) // 48 * N * 2, device="cuda")
    xp, xw = planes(x.view(M, Cc), 0)
    yp, yw = planes(dy.view(M, N), 0)
    wp, ww = planes(w.view(N, -1), 1)
    wtp, wtw = (planes(wt.view(Cc, -1), 1) if Cc % 64 == 0 else (None, None))
    fl = 2.0 * M * N * k * k * Cc
    line = "B%d %dx%d C%d->N%d k%d d%d | " % (B, H, W, Cc, N, k, dil)
    for nm, sp in VARIANTS:
        if which in ("all", "fwd"):
            d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=None if os.environ.get("BENCH_NOSTATS") else stats.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc,
                         Ho=H, Wo=W, N=N, ldy=N, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=0, y_f32=0, accum=0, mode=0)
            d.f32_split = sp
            if sp == 2:
                d.x_planes, d.x_unscale, d.x_plane_stride = xp.data_ptr(), xw.data_ptr() + 4096, xp.shape[1]
                d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
            t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
            digest("fwd %s %s" % (line.strip(), nm), y, stats)
            line += "fwd+st %s %.1fus %.0fTF | " % (nm, t * 1e6, fl / t / 1e12)
        if which in ("all", "dgrad") and Cc % 128 == 0:
            d = ConvDesc(x=dy.data_ptr(), w=wt.data_ptr(), y=gx.data_ptr(), bias=None, stats=None, B=B, Hi=H, Wi=W, C=N, ldx=N,
                         Ho=H, Wo=W, N=Cc, ldy=Cc, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=0, y_f32=0, accum=0, mode=1)
            d.f32_split = sp
            if sp == 2:
                d.x_planes, d.x_unscale, d.x_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
                d.w_planes, d.w_unscale, d.w_plane_stride = wtp.data_ptr(), wtw.data_ptr() + 4096, wtp.shape[1]
            epi = int(os.environ.get("BENCH_EPI", "0"))      # 1: identity-branch gradient, 2: fused BN-backward sums, 3: both (as conv1's data gradient in the plan)
            if sp == 2 and epi:
                keep = []
                if epi & 1:
                    rdz = torch.randn(M, Cc, device="cuda") * 1e-3
                    rmask = torch.randint(0, 256, (M * (Cc // 4),), device="cuda", dtype=torch.uint8)
                    d.res_dz, d.res_mask, d.res_ld = rdz.data_ptr(), rmask.data_ptr(), Cc
                    keep += [rdz, rmask]
                if epi & 2:
                    by = torch.randn(M, Cc, device="cuda")
                    bmask = torch.randint(0, 256, (M * (Cc // 4),), device="cuda", dtype=torch.uint8)
                    bmean, binv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
                    part = torch.empty((M + 47) // 48 * Cc * 2, device="cuda")
                    gmx = torch.zeros(1025, device="cuda")
                    d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = by.data_ptr(), bmask.data_ptr(), bmean.data_ptr(), binv.data_ptr()
                    d.bnr_partials, d.bnr_ldy, d.bnr_relu, d.bnr_gmax = part.data_ptr(), Cc, 1, gmx.data_ptr()
                    keep += [by, bmask, bmean, binv, part, gmx]
            t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
            digest("dgrad epi%d %s %s" % (int(os.environ.get("BENCH_EPI", "0")), line.split("|")[0].strip(), nm), gx, *([part] if (sp == 2 and epi & 2) else []))
            line += "dgrad %s %.1fus %.0fTF | " % (nm, t * 1e6, fl / t / 1e12)
        if which in ("all", "wgrad"):
            ws = torch.empty(48 << 20, device="cuda")
            dw = torch.zeros(N, k, k, Cc, device="cuda")
            wg = WgradDesc(x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N,
                           R=k, S=k, stride=1, dil=dil, pad=pad, dtype=0, splitk=int(os.environ.get("BENCH_SPLITK", "0")), Cm=0, ws=ws.data_ptr(), ws_elems=ws.numel(),
                           f32_split=sp)
            if sp == 2:
                wg.x_planes, wg.x_unscale, wg.x_plane_stride = xp.data_ptr(), xw.data_ptr() + 4096, xp.shape[1]
                wg.dy_planes, wg.dy_unscale, wg.dy_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
            t = timeit(lambda: lib.dml_conv_wgrad(C.byref(wg), st))
            line += "wgrad %s %.1fus %.0fTF | " % (nm, t * 1e6, fl / t / 1e12)
    print(line, flush=True)
