"""Micro-benchmark (GPU box): conv kernels on the headline shapes; sweeps wgrad split-K and epilogue options.
The abl / phases / dmaphases modes need the tuning build: make -C open-world-semantic-segmentation_amd/csrc tuning, then
DML_LIB_PATH=open-world-semantic-segmentation_amd/dmlnet/libdmlnet_hip_tuning.so python3 tools/bench_conv.py phases"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc, WgradDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
bf = torch.bfloat16

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

SHAPES = [  # B, H, W, C, N, k, dil
    (16, 48, 48, 256, 256, 3, 1), (16, 48, 48, 1024, 256, 1, 1), (16, 48, 48, 256, 1024, 1, 1),
    (16, 192, 192, 320, 256, 3, 1), (16, 48, 48, 2048, 256, 3, 12), (16, 192, 192, 64, 256, 1, 1),
    (16, 96, 96, 128, 512, 1, 1), (16, 48, 48, 512, 512, 3, 2)]
if os.environ.get("BENCH_SHAPES"):          # "B,H,W,C,N,k,d;..." overrides the list
    SHAPES = [tuple(int(v) for v in t.split(",")) for t in os.environ["BENCH_SHAPES"].split(";")]
which = sys.argv[1] if len(sys.argv) > 1 else "all"
for (B, H, W, Cc, N, k, dil) in SHAPES:
    pad = dil * (k // 2)
    x = torch.randn(B, H, W, Cc, device="cuda").to(bf)
    w = (torch.randn(N, k, k, Cc, device="cuda") * 0.05).to(bf)
    y = torch.empty(B, H, W, N, device="cuda", dtype=bf)
    dy = torch.randn(B, H, W, N, device="cuda").to(bf)
    M = B * H * W
    stats = torch.empty((M + 63) // 64 * N * 2, device="cuda")
    fl = 2.0 * M * N * k * k * Cc
    line = "B%d %dx%d C%d->N%d k%d d%d | " % (B, H, W, Cc, N, k, dil)
    if which in ("all", "fwd"):
        for use_stats in (True, False):
            d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr() if use_stats else None,
                         B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k,
                         stride=1, dil=dil, pad=pad, dtype=1, y_f32=0, accum=0, mode=0)
            t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
            line += "fwd%s %.1fus %.0fTF | " % ("+st" if use_stats else "", t * 1e6, fl / t / 1e12)
            if use_stats:                                # the same with the K-split of the last round's tiles allowed
                tws = torch.empty(512 * 128 * 128, device="cuda"); tcnt = torch.zeros(128, dtype=torch.int32, device="cuda")
                d.tail_ws, d.tail_ws_elems, d.tail_counters, d.tail_counters_len = tws.data_ptr(), tws.numel(), tcnt.data_ptr(), 128
                t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
                line += "fwd+st+tail %.1fus %.0fTF | " % (t * 1e6, fl / t / 1e12)

    if which == "abl":
        import ctypes
        lib.dml_debug_conv_ablate.restype = ctypes.c_int
        lib.dml_debug_conv_ablate.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=None, B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=1,
                     y_f32=0, accum=0, mode=0)
        for abl, nm in ((0, "full"), (1, "no-global/no-ldswrite"), (2, "no-mfma")):
            t = timeit(lambda: lib.dml_debug_conv_ablate(C.byref(d), abl, None, None, None, st))
            line += "%s %.1fus (%.0fTF-equiv) | " % (nm, t * 1e6, fl / t / 1e12)
        # normalise-on-load probe: per-input-channel affine + ReLU between the global load and the LDS write
        psc, psh = torch.rand(Cc, device="cuda") + 0.5, torch.randn(Cc, device="cuda") * 0.1
        t = timeit(lambda: lib.dml_debug_conv_ablate(C.byref(d), 4, None, psc.data_ptr(), psh.data_ptr(), st))
        line += "affine+relu on load %.1fus | " % (t * 1e6)
        # what it would replace: one BN apply pass over the input tensor
        yb = torch.empty_like(x); mk = torch.empty(x.numel() // 8, dtype=torch.uint8, device="cuda")
        mean0 = torch.zeros(Cc, device="cuda")
        t = timeit(lambda: lib.dml_bn_apply(x.data_ptr(), None, yb.data_ptr(), psc.data_ptr(), psh.data_ptr(), mean0.data_ptr(),
                                            mk.data_ptr(), B * H * W, Cc, Cc, 0, Cc, 1, 1, 0.0, 0, None, None, 0, 0, None, 0, None, st))
        line += "bn_apply of the input %.1fus | " % (t * 1e6)
    if which == "phases":
        import ctypes
        lib.dml_debug_conv_ablate.restype = ctypes.c_int
        lib.dml_debug_conv_ablate.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        nblk = ((M + 127) // 128) * ((N + 127) // 128)
        dbg = torch.zeros(nblk * 4 * 8, device="cuda")
        d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=None, B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k, stride=1, dil=dil,
                     pad=pad, dtype=1, y_f32=0, accum=0, mode=0)
        t = timeit(lambda: lib.dml_debug_conv_ablate(C.byref(d), 3, dbg.data_ptr(), None, None, st))
        torch.cuda.synchronize()
        full = dbg.view(nblk * 4, 8).cpu()
        ph = full[:, :6]
        names = ("ld-issue", "frag-read", "mfma", "vmcnt", "lds-write", "barrier")
        line += "%.1fus %.0fTF; cycles per K-step (mean over %d waves, total %.0f): " % (t * 1e6, fl / t / 1e12, ph.shape[0], ph.sum(1).mean())
        line += ", ".join("%s %.0f" % (n, v) for n, v in zip(names, ph.mean(0).tolist()))
        line += " | p10/p90 total %.0f/%.0f" % (ph.sum(1).quantile(0.1), ph.sum(1).quantile(0.9))
        line += " | prologue %.0f (p90 %.0f) epilogue %.0f (p90 %.0f) loop %.0f cycles" % (
            full[:, 6].mean(), full[:, 6].quantile(0.9), full[:, 7].mean(), full[:, 7].quantile(0.9),
            ph.sum(1).mean() * (k * k * Cc // 32))
    if which == "epi":
        # data gradient of this conv (output = B x H x W x Cc) with the fused epilogue options of the train step
        M_ = B * H * W
        wt = (torch.randn(Cc, k, k, N, device="cuda") * 0.05).to(bf)
        gx = torch.zeros(B, H, W, Cc, device="cuda", dtype=bf)
        rdz = torch.randn(B, H, W, Cc, device="cuda").to(bf)
        ypre = torch.randn(B, H, W, Cc, device="cuda").to(bf)
        mask = torch.randint(0, 256, (M_ * Cc // 8,), dtype=torch.uint8, device="cuda")
        mean, invstd = torch.randn(Cc, device="cuda") * 0.1, torch.rand(Cc, device="cuda") + 0.5
        part = torch.empty((M_ + 63) // 64 * Cc * 2, device="cuda")
        tws = torch.empty(512 * 128 * 128, device="cuda"); tcnt = torch.zeros(128, dtype=torch.int32, device="cuda")
        fl_ = 2.0 * M_ * N * k * k * Cc
        for nm, acc_, res_, bnr_ in (("plain", 0, 0, 0), ("accum", 1, 0, 0), ("res", 0, 1, 0), ("bnr", 0, 0, 1), ("res+bnr", 0, 1, 1)):
            d = ConvDesc(x=dy.data_ptr(), w=wt.data_ptr(), y=gx.data_ptr(), bias=None, stats=None, B=B, Hi=H, Wi=W, C=N, ldx=N, Ho=H, Wo=W, N=Cc, ldy=Cc, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=1,
                         y_f32=0, accum=acc_, mode=1)
            d.tail_ws, d.tail_ws_elems, d.tail_counters, d.tail_counters_len = tws.data_ptr(), tws.numel(), tcnt.data_ptr(), 128
            if res_:
                d.res_dz, d.res_mask, d.res_ld = rdz.data_ptr(), mask.data_ptr(), Cc
            if bnr_:
                d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ypre.data_ptr(), mask.data_ptr(), mean.data_ptr(), invstd.data_ptr()
                d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), Cc, 1
            t = timeit(lambda: lib.dml_conv_igemm(C.byref(d), st))
            by = 2.0 * M_ * (N + Cc * (1 + acc_ + res_ + bnr_)) + M_ * Cc / 8 * (res_ + bnr_)
            line += "%s %.1fus %.0fTF %.2fTB/s | " % (nm, t * 1e6, fl_ / t / 1e12, by / t / 1e12)
    if which == "dmaphases":
        # cycles per K step of the LDS-DMA 128 x 128 kernel by phase (s_memtime stamps, dml_debug_conv_ablate 5)
        import ctypes
        lib.dml_debug_conv_ablate.restype = ctypes.c_int
        lib.dml_debug_conv_ablate.argtypes = [ctypes.POINTER(ConvDesc), ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
        nblk = ((M + 127) // 128) * ((N + 127) // 128)
        dbg = torch.zeros(nblk * 4 * 8, device="cuda")
        d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=None, B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N, ldy=N, R=k, S=k, stride=1, dil=dil,
                     pad=pad, dtype=1, y_f32=0, accum=0, mode=0)
        t = timeit(lambda: lib.dml_debug_conv_ablate(C.byref(d), int(os.environ.get("DMAPH_ABL", "5")), dbg.data_ptr(), None, None, st))
        torch.cuda.synchronize()
        full = dbg.view(nblk * 4, 8).cpu()
        names = ("vmcnt-wait", "barrier", "dma-issue", "frag-read+mfma-issue")
        line += "%d tiles, %.1f us (with stamps); s_memtime ticks per K step, mean over %d waves: " % (nblk, t * 1e6, full.shape[0])
        line += ", ".join("%s %.0f" % (n, v) for n, v in zip(names, full[:, :4].mean(0).tolist()))
        line += " | whole step %.0f (p10 %.0f, p90 %.0f)" % (full[:, 4].mean(), full[:, 4].quantile(0.1), full[:, 4].quantile(0.9))
    if which in ("all", "wgrad"):
        dw = torch.zeros(N, k, k, Cc, device="cuda")
        for sk in (0, 4, 8, 16, 32, 64):
            wg = WgradDesc(x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N,
                           ldy=N, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=1, splitk=sk)
            t = timeit(lambda: lib.dml_conv_wgrad(C.byref(wg), st))
            line += "wg sk%d %.1fus %.0fTF | " % (sk, t * 1e6, fl / t / 1e12)
        ws = torch.empty(40 << 20, device="cuda")
        for sk in (0, 8, 16, 32, 64):
            wg = WgradDesc(x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W, N=N,
                           ldy=N, R=k, S=k, stride=1, dil=dil, pad=pad, dtype=1, splitk=sk, Cm=0, ws=ws.data_ptr(),
                           ws_elems=ws.numel())
            t = timeit(lambda: lib.dml_conv_wgrad(C.byref(wg), st))
            line += "WS sk%d %.1fus %.0fTF | " % (sk, t * 1e6, fl / t / 1e12)
    print(line)
