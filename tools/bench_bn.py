"""Micro-benchmark (GPU box): BatchNorm apply / backward kernels on the headline shapes, in TB/s of algorithmic bytes.
    python3 tools/bench_bn.py [bf16|f32|planes]     planes: fp32 tensors, outputs as fp16 planes (the f16x2 plan's launches)"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
mode = sys.argv[1] if len(sys.argv) > 1 else "bf16"
dt_t, dt_c, es = (torch.bfloat16, 1, 2) if mode == "bf16" else (torch.float32, 0, 4)
V = 16 // es

def timeit(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3

SHAPES = [(16 * 192 * 192, 256), (16 * 192 * 192, 64), (16 * 96 * 96, 512), (16 * 48 * 48, 1024), (16 * 48 * 48, 256),
          (16 * 48 * 48, 2048)]
# a pool of distinct buffers so that consecutive launches do not hit the same lines in L2 / MALL
for (M, N) in SHAPES:
    E = M * N
    nbuf = max(2, int(1200e6 // (E * es)))
    ys = [torch.randn(M, N, device="cuda").to(dt_t) for _ in range(nbuf)]
    zs = [torch.empty(M, N, device="cuda", dtype=dt_t) for _ in range(nbuf)]
    rs = [torch.randn(M, N, device="cuda").to(dt_t) for _ in range(min(nbuf, 3))]
    pls = [torch.empty(2, E, device="cuda", dtype=torch.float16) for _ in range(nbuf)] if mode == "planes" else None
    work = torch.zeros(1025, device="cuda"); work[1024] = 2.0 ** -10
    mk = torch.empty(E // V, device="cuda", dtype=torch.uint8)
    sc, sh, mu, inv = (torch.rand(N, device="cuda") + 0.5 for _ in range(4))
    part = torch.empty(4096 * N * 2, device="cuda")
    coef = torch.rand(4 * N, device="cuda")
    nb = C.c_int(0)
    it = [0]
    def pl(i):
        return (pls[i].data_ptr(), E, N, work.data_ptr() + 4096) if pls is not None else (None, 0, 0, None)
    def apply(res, z=True):
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_apply(ys[i].data_ptr(), rs[i % len(rs)].data_ptr() if res else None, zs[i].data_ptr() if z else None,
                                    sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), mk.data_ptr(), M, N, N, N, N, 1, dt_c, 0.0, 0, None,
                                    *pl(i), 0, None, st), "apply")
    def apply_rp():          # residual from the planes of another tensor, output as planes only (round 5: block outputs)
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_apply(ys[i].data_ptr(), pls[(i + 1) % nbuf].data_ptr(), None,
                                    sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), mk.data_ptr(), M, N, N, N, N, 1, dt_c, 0.0, 0, None,
                                    *pl(i), E, work.data_ptr() + 4096, st), "apply")
    def reduce():
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_bwd_reduce(zs[i].data_ptr(), ys[i].data_ptr(), None, mk.data_ptr(), mu.data_ptr(), inv.data_ptr(),
                                         part.data_ptr(), M, N, N, N, N, 1, 1.0, dt_c, C.byref(nb),
                                         work.data_ptr() if mode == "planes" else None, st), "reduce")
    def bapply(res, dy=True):
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_bwd_apply(zs[i].data_ptr(), ys[i].data_ptr(), None, mk.data_ptr(), coef.data_ptr(),
                                        zs[(i + 1) % nbuf].data_ptr() if dy else None, rs[i % len(rs)].data_ptr() if res else None,
                                        M, N, N, N, N, N, N, 1, 1.0, 0, dt_c, None, *pl(i), st), "bwd apply")
    line = "M=%d N=%d | " % (M, N)
    m8 = 1.0 / (8 * V / 8) / es * 0 + 1.0 / V / es           # mask bytes per element, in units of es
    if mode == "planes":
        cases = (("apply->planes", lambda: apply(False, False), (2 + m8) * es * E), ("apply+res->z+planes", lambda: apply(True), (4 + m8) * es * E),
                 ("apply+res(planes)->planes", apply_rp, (3 + m8) * es * E),
                 ("bwd_reduce", reduce, (2 + m8) * es * E), ("bwd_apply->planes", lambda: bapply(False, False), (3 + m8) * es * E),
                 ("bwd_apply+dres->planes", lambda: bapply(True, False), (4 + m8) * es * E))
    else:
        cases = (("apply", lambda: apply(False), (2 + m8) * es * E), ("apply+res", lambda: apply(True), (3 + m8) * es * E),
                 ("bwd_reduce", reduce, (2 + m8) * es * E), ("bwd_apply", lambda: bapply(False), (3 + m8) * es * E),
                 ("bwd_apply+dres", lambda: bapply(True), (4 + m8) * es * E))
    for name, fn, bytes_ in cases:
        t = timeit(fn)
        line += "%s %.1fus %.2fTB/s | " % (name, t * 1e6, bytes_ / t / 1e12)
    print(line, flush=True)
    del ys, zs, rs, pls
