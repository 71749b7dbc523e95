"""GPU box: the stem's exact-fp32 convolution as shipped (7x7 stride 2 on the image padded to 8 channels: K = 392) against its
space-to-depth form (4x4 stride 1 on [B][H/2][W/2][12]: K = 192) -- forward (with BN statistics) and weight gradient, same kernels.
    python3 tools/probe_stem_s2d.py"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
from dmlnet import _lib
from dmlnet._lib import ConvDesc, WgradDesc
lib = _lib.load()
st = torch.cuda.current_stream().cuda_stream

def timeit(fn, n=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n

B, H, W, N = 16, 768, 768, 64
Ho = Wo = 384
M = B * Ho * Wo
dy = torch.randn(B, Ho, Wo, N, device="cuda")
ws = torch.empty(64 << 20, device="cuda")
for name, (Hi, Wi, Cc, k, s, p) in (("7x7 s2 C=8 (shipped)", (H, W, 8, 7, 2, 3)), ("4x4 s1 C=12 (space-to-depth)", (H // 2, W // 2, 12, 4, 1, 2))):
    x = torch.randn(B, Hi, Wi, Cc, device="cuda")
    w = torch.randn(N, k, k, Cc, device="cuda") * 0.05
    y = torch.empty(B, Ho, Wo, N, device="cuda")
    stats = torch.zeros((M + 63) // 64 * N * 2, device="cuda")
    d = ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=None, stats=stats.data_ptr(), B=B, Hi=Hi, Wi=Wi, C=Cc, ldx=Cc,
                 Ho=Ho, Wo=Wo, N=N, ldy=N, R=k, S=k, stride=s, dil=1, pad=p, dtype=0, y_f32=0, accum=0, mode=0)
    t_f = timeit(lambda: _lib.check(lib.dml_conv_igemm(C.byref(d), st), "fwd"))
    dw = torch.zeros(N, k, k, Cc, device="cuda")
    wg = WgradDesc(x=x.data_ptr(), dy=dy.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hi, Wi=Wi, C=Cc, ldx=Cc, Ho=Ho, Wo=Wo, N=N, ldy=N,
                   R=k, S=k, stride=s, dil=1, pad=p, dtype=0, splitk=0, Cm=Cc, ws=ws.data_ptr(), ws_elems=ws.numel(), f32_split=0)
    t_w = timeit(lambda: _lib.check(lib.dml_conv_wgrad(C.byref(wg), st), "wgrad"))
    wg.f32_split = 1
    t_w3 = timeit(lambda: _lib.check(lib.dml_conv_wgrad(C.byref(wg), st), "wgrad x3"))
    print("%-30s forward %.3f ms   weight gradient exact %.3f ms, three-term %.3f ms" % (name, t_f, t_w, t_w3))

# equivalence of the two forms through the library's own pack / weight kernels, against torch on a small image
torch.manual_seed(0)
B, H, W, N = 2, 20, 28, 64
img = torch.randn(B, 3, H, W, device="cuda")
wm = torch.randn(N, 3, 7, 7, device="cuda") * 0.1
ref = torch.nn.functional.conv2d(img.double(), wm.double(), stride=2, padding=3)
wcl = wm.permute(0, 2, 3, 1).contiguous()                     # the parameter store's layout [N][7][7][3]
x2 = torch.empty(B, H // 2, W // 2, 12, device="cuda")
w2 = torch.empty(N, 4, 4, 12, device="cuda")
_lib.check(lib.dml_pack_input_s2d(img.data_ptr(), x2.data_ptr(), B, 3, H, W, st), "pack")
_lib.check(lib.dml_s2d_weights(wcl.data_ptr(), w2.data_ptr(), N, 7, 3, st), "weights")
y = torch.empty(B, H // 2, W // 2, N, device="cuda")
d = ConvDesc(x=x2.data_ptr(), w=w2.data_ptr(), y=y.data_ptr(), bias=None, stats=None, B=B, Hi=H // 2, Wi=W // 2, C=12, ldx=12,
             Ho=H // 2, Wo=W // 2, N=N, ldy=N, R=4, S=4, stride=1, dil=1, pad=2, dtype=0, y_f32=0, accum=0, mode=0)
_lib.check(lib.dml_conv_igemm(C.byref(d), st), "fwd")
torch.cuda.synchronize()
print("s2d forward vs torch fp64: max rel err %.2e" % ((y.permute(0, 3, 1, 2).double() - ref).abs().max() / ref.abs().max()).item())
gy = torch.randn_like(ref)
gref = torch.nn.grad.conv2d_weight(img.double(), wm.shape, gy, stride=2, padding=3)
dyd = gy.float().permute(0, 2, 3, 1).contiguous()
dw2 = torch.zeros(N, 4, 4, 12, device="cuda")
ws2 = torch.empty(1 << 22, device="cuda")
for split in (0, 1):
    dw2.zero_()
    wg = WgradDesc(x=x2.data_ptr(), dy=dyd.data_ptr(), dw=dw2.data_ptr(), B=B, Hi=H // 2, Wi=W // 2, C=12, ldx=12, Ho=H // 2, Wo=W // 2,
                   N=N, ldy=N, R=4, S=4, stride=1, dil=1, pad=2, dtype=0, splitk=0, Cm=12, ws=ws2.data_ptr(), ws_elems=ws2.numel(),
                   f32_split=split)
    _lib.check(lib.dml_conv_wgrad(C.byref(wg), st), "wgrad")
    dw = torch.full((N, 7, 7, 3), 1.0, device="cuda")
    _lib.check(lib.dml_s2d_wgrad(dw2.data_ptr(), dw.data_ptr(), N, 7, 3, st), "unpack")
    torch.cuda.synchronize()
    err = ((dw - 1.0).permute(0, 3, 1, 2).double() - gref).abs().max() / gref.abs().max()
    print("s2d weight gradient (f32_split %d) vs torch fp64: max rel err %.2e" % (split, err.item()))
