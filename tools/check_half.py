"""GPU box: numeric check of one two-plane forward launch (whatever configuration the environment selects) against F.conv2d in fp64."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream
B, H, W, Cc, N = 16, 48, 48, 256, 1024
M = B * H * W
torch.manual_seed(0)
x = torch.randn(B, H, W, Cc, device="cuda")
w = (torch.randn(N, 1, 1, Cc, device="cuda") * 0.05).contiguous()
y = torch.zeros(B, H, W, N, device="cuda")
stats = torch.zeros((M + 47) // 48 * N * 2, device="cuda")


def planes(t2d, layout):
    rows, c = t2d.shape
    pl = torch.empty((2, rows * c), device="cuda", dtype=torch.float16)
    work = torch.zeros(1025, device="cuda")
    assert lib.dml_h2_split(t2d.data_ptr(), rows, c, c, pl.data_ptr(), rows * c, c, layout, work.data_ptr(), 0, st) == 0
    return pl, work


xp, xw = planes(x.view(M, Cc), 0)
wp, ww = planes(w.view(N, -1), 1)
d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr(), B=B, Hi=H, Wi=W, C=Cc, ldx=Cc, Ho=H, Wo=W,
             N=N, ldy=N, R=1, S=1, stride=1, dil=1, pad=0, dtype=0, y_f32=0, accum=0, mode=0)
d.f32_split = 2
d.x_planes, d.x_unscale, d.x_plane_stride = xp.data_ptr(), xw.data_ptr() + 4096, xp.shape[1]
d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
assert lib.dml_conv_igemm(C.byref(d), st) == 0
torch.cuda.synchronize()
ref = (x.view(M, Cc).double() @ w.view(N, Cc).double().t())
got = y.view(M, N).double()
err = (got - ref).abs()
print("max |d| %.3e of scale %.3e; rows with error > 1e-3: %d; first bad rows %s; bad columns (of 64-blocks) %s" % (
    err.max().item(), ref.abs().max().item(), int((err.max(1).values > 1e-3).sum()),
    (err.max(1).values > 1e-3).nonzero().flatten()[:8].tolist(), sorted(set(((err.max(0).values > 1e-3).nonzero().flatten() // 64).tolist()))))
s = stats.view(-1, N, 2)[:, :, 0].sum(0).double()
print("statistics: column sums rel err %.3e" % ((s - ref.sum(0)).abs().max() / ref.sum(0).abs().max()).item())
