"""Micro-benchmark (GPU box): BatchNorm apply / backward kernels on the headline shapes, in TB/s of algorithmic bytes."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
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

SHAPES = [(16 * 192 * 192, 256), (16 * 192 * 192, 64), (16 * 96 * 96, 512), (16 * 48 * 48, 1024), (16 * 48 * 48, 256),
          (16 * 48 * 48, 2048)]
# a pool of distinct buffers so that consecutive launches do not hit the same lines in L2 / MALL
for (M, N) in SHAPES:
    E = M * N
    nbuf = max(2, int(600e6 // (E * 2)))
    ys = [torch.randn(M, N, device="cuda").to(bf) for _ in range(nbuf)]
    zs = [torch.empty(M, N, device="cuda", dtype=bf) for _ in range(nbuf)]
    rs = [torch.randn(M, N, device="cuda").to(bf) for _ in range(min(nbuf, 3))]
    mk = torch.empty(M * N // 8, device="cuda", dtype=torch.uint8)
    sc, sh, mu, inv = (torch.rand(N, device="cuda") + 0.5 for _ in range(4))
    part = torch.empty(4096 * N * 2, device="cuda")
    coef = torch.rand(4 * N, device="cuda")
    nb = C.c_int(0)
    it = [0]
    def apply(res):
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_apply(ys[i].data_ptr(), rs[i % len(rs)].data_ptr() if res else None, zs[i].data_ptr(), sc.data_ptr(),
                                    sh.data_ptr(), mu.data_ptr(), mk.data_ptr(), M, N, N, N, N, 1, 1, 0.0, 0, None, None, 0, 0, None, st), "apply")
    def reduce():
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_bwd_reduce(zs[i].data_ptr(), ys[i].data_ptr(), None, mk.data_ptr(), mu.data_ptr(), inv.data_ptr(),
                                         part.data_ptr(), M, N, N, N, N, 1, 1.0, 1, C.byref(nb), None, st), "reduce")
    def bapply(res):
        i = it[0] = (it[0] + 1) % nbuf
        _lib.check(lib.dml_bn_bwd_apply(zs[i].data_ptr(), ys[i].data_ptr(), None, mk.data_ptr(), coef.data_ptr(),
                                        zs[(i + 1) % nbuf].data_ptr(), rs[i % len(rs)].data_ptr() if res else None, M, N, N, N, N, N, N,
                                        1, 1.0, 0, 1, None, None, 0, 0, None, st), "bwd apply")
    line = "M=%d N=%d | " % (M, N)
    for name, fn, bytes_ in (("apply", lambda: apply(False), 4.125 * E), ("apply+res", lambda: apply(True), 6.125 * E),
                             ("bwd_reduce", reduce, 4.125 * E), ("bwd_apply", lambda: bapply(False), 6.125 * E),
                             ("bwd_apply+dres", lambda: bapply(True), 8.125 * E)):
        t = timeit(fn)
        line += "%s %.1fus %.2fTB/s | " % (name, t * 1e6, bytes_ / t / 1e12)
    print(line, flush=True)
    del ys, zs, rs
