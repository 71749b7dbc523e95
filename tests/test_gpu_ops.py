"""GPU: every C-ABI kernel against a plain PyTorch fp32 reference of the same op (computed on the CPU),
and against the golden vectors where the reference pins the op (G1, G2, G6, G7).

Tolerances: fp32 kernels 2e-5 relative to the tensor's max magnitude (exact-fp32 MFMA, different summation
order); bf16 kernels are compared with the reference evaluated on the SAME bf16-rounded inputs, 1e-2
relative (bf16 output rounding is 2^-8 = 3.9e-3).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import helpers as H

pytestmark = pytest.mark.gpu

DT = {"f32": (0, torch.float32, 2e-5), "bf16": (1, torch.bfloat16, 1e-2)}


@pytest.fixture(scope="module")
def lib():
    from dmlnet import _lib
    return _lib.load()


def st():
    return torch.cuda.current_stream().cuda_stream


def chk(rc):
    assert rc == 0, "kernel returned %d" % rc


def nhwc(t, dtype):          # NCHW cpu -> NHWC cuda
    return t.permute(0, 2, 3, 1).contiguous().to("cuda", dtype)


def nchw(t):                 # NHWC cuda -> NCHW cpu float
    return t.float().cpu().permute(0, 3, 1, 2).contiguous()


def rnd(name, shape, scale=1.0):
    return H.synth_tensor(77, name, shape, scale=scale)


def qz(t, dtype):            # round through the storage dtype
    return t.to(dtype).float()


def relclose(got, ref, tol, what=""):
    err = (got.double() - ref.double()).abs().max().item()
    scale = ref.double().abs().max().item() + 1e-12
    assert err <= tol * scale, "%s: max|d|=%.3e scale=%.3e rel=%.3e > %.1e" % (what, err, scale, err / scale, tol)


CONV_CASES = [
    # name, B, H, W, Cin, Cout, k, stride, dil
    ("1x1", 2, 12, 10, 64, 128, 1, 1, 1),
    ("1x1_n48", 2, 9, 7, 256, 48, 1, 1, 1),
    ("1x1_s2", 2, 12, 10, 64, 256, 1, 2, 1),
    ("3x3", 2, 11, 13, 64, 64, 3, 1, 1),
    ("3x3_s2", 2, 12, 14, 128, 128, 3, 2, 1),
    ("3x3_d2", 2, 10, 9, 64, 96, 3, 1, 2),
    ("3x3_d6", 2, 8, 8, 128, 256, 3, 1, 6),
    ("3x3_c320", 1, 9, 9, 320, 256, 3, 1, 1),
    ("7x7_s2_stem", 2, 22, 18, 8, 64, 7, 2, 1),
    ("1x1_pool_rows", 4, 1, 1, 256, 256, 1, 1, 1),
]


def conv_ref(x, w, k, stride, dil, bias=None):
    pad = 3 if k == 7 else dil * (k // 2)
    return F.conv2d(x, w, bias, stride=stride, padding=pad, dilation=dil), pad


def make_desc(lib, x, w, y, B, Hi, Wi, Cin, Ho, Wo, Cout, k, stride, dil, pad, dt, mode=0, stats=None, bias=None,
              y_f32=0, accum=0, ldx=None, ldy=None):
    from dmlnet._lib import ConvDesc
    return ConvDesc(x=x.data_ptr(), w=w.data_ptr(), y=y.data_ptr(), bias=bias.data_ptr() if bias is not None else None,
                    stats=stats.data_ptr() if stats is not None else None, B=B, Hi=Hi, Wi=Wi, C=Cin, ldx=ldx or Cin, Ho=Ho, Wo=Wo, N=Cout, ldy=ldy or Cout, R=k, S=k,
                    stride=stride, dil=dil, pad=pad, dtype=dt, y_f32=y_f32, accum=accum, mode=mode)


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("case", CONV_CASES, ids=[c[0] for c in CONV_CASES])
def test_conv_fwd_dgrad_wgrad(lib, case, dname):
    name, B, Hh, Ww, Cin, Cout, k, stride, dil = case
    dt, tdt, tol = DT[dname]
    x = qz(rnd(name + ".x", (B, Cin, Hh, Ww)), tdt).requires_grad_(True)
    w = qz(rnd(name + ".w", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5), tdt).requires_grad_(True)
    y_ref, pad = conv_ref(x, w, k, stride, dil)
    Ho, Wo = y_ref.shape[2:]
    gy = qz(rnd(name + ".gy", tuple(y_ref.shape)), tdt)
    y_ref.backward(gy)

    xd = nhwc(x.detach(), tdt)
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to("cuda", tdt)          # K R S C
    wtd = w.detach().permute(1, 2, 3, 0).contiguous().to("cuda", tdt)         # C R S K (dgrad copy)
    yd = torch.empty((B, Ho, Wo, Cout), device="cuda", dtype=tdt)
    M = B * Ho * Wo
    groups = (M + 63) // 64
    stats = torch.zeros(groups * Cout * 2, device="cuda")
    d = make_desc(lib, xd, wd, yd, B, Hh, Ww, Cin, Ho, Wo, Cout, k, stride, dil, pad, dt, stats=stats)
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    torch.cuda.synchronize()
    relclose(nchw(yd), y_ref.detach(), tol, "conv fwd " + name)

    # fused BN statistics: finalize and compare with the batch statistics of the fp32 result
    sc, sh, mu, inv = (torch.empty(Cout, device="cuda") for _ in range(4))
    rm, rv = torch.zeros(Cout, device="cuda"), torch.ones(Cout, device="cuda")
    chk(lib.dml_bn_finalize(stats.data_ptr(), M, Cout, 64, None, None, rm.data_ptr(), rv.data_ptr(), 0.1, 1e-5,
                            sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), inv.data_ptr(), st()))
    yr = y_ref.detach().double()
    mean_ref, var_ref = yr.mean(dim=(0, 2, 3)), yr.var(dim=(0, 2, 3), unbiased=False)
    stol = 1e-4 if dname == "f32" else 2e-2
    assert (mu.cpu().double() - mean_ref).abs().max() <= stol * (yr.abs().max() + 1e-6)
    relclose(inv.cpu(), (var_ref + 1e-5).rsqrt().float(), stol, "invstd " + name)
    if M > 1:
        relclose(rv.cpu(), (0.9 + 0.1 * yr.var(dim=(0, 2, 3), unbiased=True)).float(), stol, "running_var " + name)

    # data gradient (transposed conv as a gather)
    if name != "7x7_s2_stem":
        gyd = nhwc(gy, tdt)
        dxd = torch.full((B, Hh, Ww, Cin), 7.0, device="cuda", dtype=tdt)
        dd = make_desc(lib, gyd, wtd, dxd, B, Ho, Wo, Cout, Hh, Ww, Cin, k, stride, dil, pad, dt, mode=1)
        chk(lib.dml_conv_igemm(C.byref(dd), st()))
        torch.cuda.synchronize()
        relclose(nchw(dxd), x.grad, tol, "conv dgrad " + name)
        # accumulate flag
        dd.accum = 1
        chk(lib.dml_conv_igemm(C.byref(dd), st()))
        torch.cuda.synchronize()
        relclose(nchw(dxd), 2 * x.grad, 2 * tol, "conv dgrad accum " + name)

    # weight gradient (fp32 atomics into a zeroed buffer)
    from dmlnet._lib import WgradDesc
    gyd = nhwc(gy, tdt)
    dw = torch.zeros((Cout, k, k, Cin), device="cuda")
    wg = WgradDesc(x=xd.data_ptr(), dy=gyd.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hh, Wi=Ww, C=Cin, ldx=Cin, Ho=Ho,
                   Wo=Wo, N=Cout, ldy=Cout, R=k, S=k, stride=stride, dil=dil, pad=pad, dtype=dt, splitk=0)
    chk(lib.dml_conv_wgrad(C.byref(wg), st()))
    torch.cuda.synchronize()
    relclose(dw.cpu().permute(0, 3, 1, 2), w.grad, tol if dname == "f32" else 2e-3, "conv wgrad " + name)
    wg.splitk = 3
    dw.zero_()
    chk(lib.dml_conv_wgrad(C.byref(wg), st()))
    torch.cuda.synchronize()
    relclose(dw.cpu().permute(0, 3, 1, 2), w.grad, tol if dname == "f32" else 2e-3, "conv wgrad splitk " + name)
    # workspace path: partial tiles with plain stores + reduce kernel (no atomics); also drops padded channels
    ws = torch.empty(6 * dw.numel(), device="cuda")
    for sk in (0, 4):
        wg.splitk, wg.ws, wg.ws_elems = sk, ws.data_ptr(), ws.numel()
        dw.fill_(1.0)
        chk(lib.dml_conv_wgrad(C.byref(wg), st()))
        torch.cuda.synchronize()
        relclose(dw.cpu().permute(0, 3, 1, 2) - 1.0, w.grad, tol if dname == "f32" else 2e-3, "conv wgrad ws " + name)
    if Cin >= 8:
        cm = Cin - 5
        dwc = torch.zeros((Cout, k, k, cm), device="cuda")
        wg.splitk, wg.Cm, wg.dw = 0, cm, dwc.data_ptr()
        chk(lib.dml_conv_wgrad(C.byref(wg), st()))
        torch.cuda.synchronize()
        relclose(dwc.cpu().permute(0, 3, 1, 2), w.grad[:, :cm], tol if dname == "f32" else 2e-3, "conv wgrad Cm " + name)


WGRAD_BIG_CASES = [
    # name, B, H, W, Cin, Cout, k, stride, dil, Cm
    ("3x3_d2", 2, 20, 24, 64, 256, 3, 1, 2, 0),
    ("1x1", 3, 16, 18, 1024, 512, 1, 1, 1, 0),
    ("3x3_s2", 2, 40, 36, 96, 256, 3, 2, 1, 0),
    ("3x3_c320_cm304", 1, 24, 24, 320, 256, 3, 1, 1, 304),
    ("3x3_w7", 5, 16, 7, 64, 256, 3, 1, 1, 0),
]


@pytest.mark.parametrize("case", WGRAD_BIG_CASES, ids=[c[0] for c in WGRAD_BIG_CASES])
def test_conv_wgrad_large_tile(lib, case):
    """bf16 weight gradient through the 256 x 256 tile kernel (N % 256 == 0, >= 16 pixel slabs, workspace given):
    partial last kc tile, stride 2, rows narrower than a 32-pixel slab, ragged last slab, dropped pad channels."""
    from dmlnet._lib import WgradDesc
    name, B, Hh, Ww, Cin, Cout, k, stride, dil, Cm = case
    x = qz(rnd("wgb.x" + name, (B, Cin, Hh, Ww)), torch.bfloat16)
    w = qz(rnd("wgb.w" + name, (Cout, Cin, k, k), scale=0.05), torch.bfloat16).requires_grad_(True)
    y, pad = conv_ref(x, w, k, stride, dil)
    Ho, Wo = y.shape[2], y.shape[3]
    gy = qz(rnd("wgb.gy" + name, tuple(y.shape)), torch.bfloat16)
    (y * gy).sum().backward()
    xd, gyd = nhwc(x, torch.bfloat16), nhwc(gy, torch.bfloat16)
    cm = Cm or Cin
    dw = torch.full((Cout, k, k, cm), 1.0, device="cuda")
    ws = torch.empty(30 * Cout * k * k * Cin, device="cuda")
    for sk in (0, 5):
        dw.fill_(1.0)
        wg = WgradDesc(x=xd.data_ptr(), dy=gyd.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hh, Wi=Ww, C=Cin, ldx=Cin, Ho=Ho,
                       Wo=Wo, N=Cout, ldy=Cout, R=k, S=k, stride=stride, dil=dil, pad=pad, dtype=1, splitk=sk, Cm=Cm,
                       ws=ws.data_ptr(), ws_elems=ws.numel())
        chk(lib.dml_conv_wgrad(C.byref(wg), st()))
        torch.cuda.synchronize()
        relclose(dw.cpu().permute(0, 3, 1, 2) - 1.0, w.grad[:, :cm], 2e-3, "wgrad large tile %s splitk=%d" % (name, sk))


@pytest.mark.parametrize("accum,relu", [(0, 1), (1, 1), (1, 0)])
def test_dgrad_emits_bn_backward_partials(lib, accum, relu):
    """DmlConvDesc.bnr_*: the data gradient's epilogue writes the BN-backward sums of the tensor it stores; they must
    equal what dml_bn_bwd_reduce computes from that stored tensor (same bf16-rounded values), incl. a ragged last
    64-row group and the accumulate path."""
    B, Hh, Ww, Cin, Cout, k = 2, 13, 11, 128, 64, 3          # dgrad output: M = 286 rows (4.47 groups) x 128 channels
    M = B * Hh * Ww
    gy = qz(rnd("bnr.gy", (B, Cout, Hh, Ww)), torch.bfloat16)
    w = qz(rnd("bnr.w", (Cout, Cin, k, k), scale=0.05), torch.bfloat16)
    gyd = nhwc(gy, torch.bfloat16)
    wt = w.permute(1, 2, 3, 0).contiguous().to("cuda", torch.bfloat16)          # wt[Cin][R][S][Cout]
    dx = (torch.randn(B, Hh, Ww, Cin, device="cuda") * 0.3).to(torch.bfloat16) if accum else \
        torch.empty(B, Hh, Ww, Cin, device="cuda", dtype=torch.bfloat16)
    ybn = (torch.randn(M, Cin, device="cuda") * 1.5 + 0.3).to(torch.bfloat16)
    bits = torch.randint(0, 256, (M * Cin // 8,), device="cuda", dtype=torch.uint8)
    mean, invstd = torch.randn(Cin, device="cuda") * 0.2, torch.rand(Cin, device="cuda") + 0.5
    G = (M + 63) // 64
    part = torch.full((G * Cin * 2,), 7.0, device="cuda")
    d = make_desc(lib, gyd, wt, dx, B, Hh, Ww, Cout, Hh, Ww, Cin, k, 1, 1, 1, 1, mode=1, accum=accum)
    d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), bits.data_ptr(), mean.data_ptr(), invstd.data_ptr()
    d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), Cin, relu
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    # reference: the stand-alone reduce on the tensor the conv stored
    part2 = torch.zeros(4096 * Cin * 2, device="cuda")
    nb = C.c_int(0)
    chk(lib.dml_bn_bwd_reduce(dx.data_ptr(), ybn.data_ptr(), None, bits.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                              part2.data_ptr(), M, Cin, Cin, Cin, Cin, relu, 1.0, 1, C.byref(nb), None, st()))
    torch.cuda.synchronize()
    a = part.view(G, Cin, 2).double().sum(0)
    b = part2[: nb.value * Cin * 2].view(nb.value, Cin, 2).double().sum(0)
    relclose(a[:, 0].cpu(), b[:, 0].cpu(), 2e-6, "sum g")
    relclose(a[:, 1].cpu(), b[:, 1].cpu(), 2e-6, "sum g xhat")
    # and per group against torch on the stored values
    g = dx.view(M, Cin).float()
    if relu:
        mk = ((bits.view(M, Cin // 8, 1) >> torch.arange(8, device="cuda").view(1, 1, 8)) & 1).reshape(M, Cin).float()
        g = g * mk
    xh = (ybn.float() - mean) * invstd
    pad = G * 64 - M
    gp = torch.cat([g, torch.zeros(pad, Cin, device="cuda")]).view(G, 64, Cin)
    xp = torch.cat([xh, torch.zeros(pad, Cin, device="cuda")]).view(G, 64, Cin)
    ref = torch.stack([gp.sum(1), (gp * xp).sum(1)], dim=-1)
    relclose(part.view(G, Cin, 2).cpu(), ref.cpu(), 1e-5, "per-group partials")


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("res,relu,Cout", [(True, 1, 128), (False, 1, 48), (False, 0, 64)])
def test_conv_inference_epilogue(lib, dname, res, relu, Cout):
    """DmlConvDesc.post_*: conv + BatchNorm(running stats) + residual + ReLU in one launch vs torch (eval mode);
    scale / shift from the table form of dml_bn_eval_coeffs."""
    from dmlnet._lib import BnEvalDesc
    dt, tdt, tol = DT[dname]
    B, Hh, Ww, Cin, k = 2, 9, 7, 64, 3
    x = qz(rnd("post.x", (B, Cin, Hh, Ww)), tdt)
    w = qz(rnd("post.w", (Cout, Cin, k, k), scale=0.05), tdt)
    r = qz(rnd("post.r", (B, Cout, Hh, Ww)), tdt) if res else None
    gamma, beta = rnd("post.g", (Cout,)) * 0.2 + 1, rnd("post.b", (Cout,)) * 0.1
    rm, rv = rnd("post.rm", (Cout,)) * 0.1, rnd("post.rv", (Cout,)).abs() + 0.5
    ref = F.batch_norm(F.conv2d(x, w, padding=1), rm, rv, gamma, beta, training=False, eps=1e-5)
    if res:
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    xd, wd = nhwc(x, tdt), w.permute(0, 2, 3, 1).contiguous().to("cuda", tdt)
    rd = nhwc(r, tdt) if res else None
    g_d, b_d, rm_d, rv_d = gamma.cuda(), beta.cuda(), rm.cuda(), rv.cuda()
    sc, sh = torch.empty(Cout, device="cuda"), torch.empty(Cout, device="cuda")
    tab = (BnEvalDesc * 1)(BnEvalDesc(g_d.data_ptr(), b_d.data_ptr(), rv_d.data_ptr(), sc.data_ptr(), sh.data_ptr(), Cout, 1e-5))
    tab_d = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).cuda()
    chk(lib.dml_bn_eval_coeffs_table(tab_d.data_ptr(), 1, st()))
    z = torch.empty((B, Hh, Ww, Cout), device="cuda", dtype=tdt)
    d = make_desc(lib, xd, wd, z, B, Hh, Ww, Cin, Hh, Ww, Cout, k, 1, 1, 1, dt)
    d.post_scale, d.post_shift, d.post_mean = sc.data_ptr(), sh.data_ptr(), rm_d.data_ptr()
    if res:
        d.post_res, d.post_ldres = rd.data_ptr(), Cout
    else:
        d.post_res, d.post_ldres = None, 0
    d.post_relu = relu
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    torch.cuda.synchronize()
    relclose(nchw(z), qz(ref, tdt) if dname == "bf16" else ref, max(tol, 1e-5) if dname == "f32" else 1e-2, "conv+bn+res+relu")


def test_sync_bn_building_blocks(lib):
    """Synchronised BN pieces on one device: statistics / backward sums of two half-batches, merged the way two ranks
    would after their all_gather / all_reduce, must equal the single-rank results on the whole batch."""
    Mh, N = 64 * 7 + 13, 40
    g = torch.Generator().manual_seed(9)
    y = (torch.randn(2 * Mh, N, generator=g) * (torch.rand(N, generator=g) + 0.2) + torch.randn(N, generator=g)).cuda()
    dz = torch.randn(2 * Mh, N, generator=g).cuda()
    gam, bet = (torch.rand(N, generator=g) + 0.5).cuda(), torch.randn(N, generator=g).cuda()

    def stats(t):
        p = torch.zeros(((t.shape[0] + 63) // 64) * N * 2, device="cuda")
        chk(lib.dml_bn_stats(t.data_ptr(), p.data_ptr(), t.shape[0], N, N, 0, st()))
        return p

    # whole batch, ordinary path
    ref = [torch.empty(N, device="cuda") for _ in range(4)]
    rm_ref, rv_ref = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    chk(lib.dml_bn_finalize(stats(y).data_ptr(), 2 * Mh, N, 64, gam.data_ptr(), bet.data_ptr(), rm_ref.data_ptr(), rv_ref.data_ptr(),
                            0.1, 1e-5, *[t.data_ptr() for t in ref], st()))
    # two "ranks"
    mom = torch.empty(2, N, 2, device="cuda", dtype=torch.float64)
    for r in range(2):
        chk(lib.dml_bn_moments(stats(y[r * Mh:(r + 1) * Mh]).data_ptr(), Mh, N, 64, mom[r].data_ptr(), st()))
    got = [torch.empty(N, device="cuda") for _ in range(4)]
    rm, rv = torch.zeros(N, device="cuda"), torch.ones(N, device="cuda")
    chk(lib.dml_bn_finalize_moments(mom.data_ptr(), 2, Mh, N, gam.data_ptr(), bet.data_ptr(), rm.data_ptr(), rv.data_ptr(), 0.1,
                                    1e-5, *[t.data_ptr() for t in got], st()))
    torch.cuda.synchronize()
    for a, b, what in zip(got + [rm, rv], ref + [rm_ref, rv_ref], ("scale", "shift", "mean", "invstd", "running_mean", "running_var")):
        assert torch.allclose(a, b, rtol=2e-6, atol=1e-7), what
    # backward: local sums of the halves add up to the whole, coefficients from the global sums
    mean, invstd = ref[2], ref[3]
    part = torch.zeros(4096 * N * 2, device="cuda")
    nb = C.c_int(0)

    def reduce(dzs, ys):
        chk(lib.dml_bn_bwd_reduce(dzs.data_ptr(), ys.data_ptr(), None, None, mean.data_ptr(), invstd.data_ptr(), part.data_ptr(),
                                  dzs.shape[0], N, N, N, N, 0, 1.0, 0, C.byref(nb), None, st()))

    coef_ref, dg_ref, db_ref = torch.empty(4 * N, device="cuda"), torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    reduce(dz, y)
    chk(lib.dml_bn_bwd_finalize(part.data_ptr(), nb, 2 * Mh, N, gam.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                                dg_ref.data_ptr(), db_ref.data_ptr(), coef_ref.data_ptr(), st()))
    sums = torch.zeros(2, N, 2, device="cuda", dtype=torch.float64)
    dg, db = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
    for r in range(2):
        reduce(dz[r * Mh:(r + 1) * Mh], y[r * Mh:(r + 1) * Mh])
        chk(lib.dml_bn_bwd_sums(part.data_ptr(), nb, N, sums[r].data_ptr(), dg.data_ptr(), db.data_ptr(), st()))
    tot = sums.sum(0).contiguous()                      # what all_reduce(sum) gives every rank
    coef = torch.empty(4 * N, device="cuda")
    chk(lib.dml_bn_bwd_coef(tot.data_ptr(), 2 * Mh, N, gam.data_ptr(), mean.data_ptr(), invstd.data_ptr(), coef.data_ptr(), st()))
    torch.cuda.synchronize()
    relclose(coef.cpu(), coef_ref.cpu(), 2e-6, "coef")
    relclose(dg.cpu(), dg_ref.cpu(), 2e-6, "dgamma (sum of the ranks' local gradients)")
    relclose(db.cpu(), db_ref.cpu(), 2e-6, "dbeta")


def test_conv_bias_f32_out_and_slices(lib):
    """Final 1x1 with bias writing fp32 from bf16 operands; producer writing into a concat-buffer slice."""
    B, Hh, Ww, Cin, K = 2, 6, 5, 256, 16
    x = qz(rnd("fin.x", (B, Cin, Hh, Ww)), torch.bfloat16)
    w = qz(rnd("fin.w", (K, Cin, 1, 1), scale=0.05), torch.bfloat16)
    b = rnd("fin.b", (K,), scale=0.1)
    ref = F.conv2d(x, w, b)
    xd, wd = nhwc(x, torch.bfloat16), w.permute(0, 2, 3, 1).contiguous().to("cuda", torch.bfloat16)
    y = torch.empty((B, Hh, Ww, K), device="cuda", dtype=torch.float32)
    d = make_desc(lib, xd, wd, y, B, Hh, Ww, Cin, Hh, Ww, K, 1, 1, 1, 0, 1, bias=b.cuda(), y_f32=1)
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    torch.cuda.synchronize()
    relclose(nchw(y), ref, 1e-5, "bias / fp32 out")
    # input taken from a channel slice (ldx > C), output into a slice (ldy > N)
    big_in = torch.zeros((B, Hh, Ww, 320), device="cuda", dtype=torch.bfloat16)
    big_in[..., 64:320] = xd
    big_out = torch.full((B, Hh, Ww, 48), 5.0, device="cuda", dtype=torch.float32)
    from dmlnet._lib import ConvDesc
    d2 = make_desc(lib, big_in, wd, big_out, B, Hh, Ww, Cin, Hh, Ww, K, 1, 1, 1, 0, 1, bias=b.cuda(), y_f32=1,
                   ldx=320, ldy=48)
    d2.x = big_in.data_ptr() + 64 * 2
    d2.y = big_out.data_ptr() + 8 * 4
    chk(lib.dml_conv_igemm(C.byref(d2), st()))
    torch.cuda.synchronize()
    relclose(nchw(big_out[..., 8:24]), ref, 1e-5, "sliced in/out")
    assert (big_out[..., :8] == 5).all() and (big_out[..., 24:] == 5).all()


@pytest.mark.parametrize("dname", ["f32", "bf16"])
def test_prep_weight_and_unpad(lib, dname):
    dt, tdt, _ = DT[dname]
    N, RS, Cm, Cp = 24, 9, 304, 320
    w = rnd("pw", (N, RS, Cm)).cuda()
    out = torch.empty((N, RS, Cp), device="cuda", dtype=tdt)
    outt = torch.empty((Cp, RS, N), device="cuda", dtype=tdt)
    chk(lib.dml_prep_weight(w.data_ptr(), out.data_ptr(), outt.data_ptr(), N, RS, Cm, Cp, dt, st()))
    torch.cuda.synchronize()
    ref = torch.zeros((N, RS, Cp))
    ref[..., :Cm] = w.cpu()
    ref = qz(ref, tdt)
    assert torch.equal(out.float().cpu(), ref)
    assert torch.equal(outt.float().cpu(), ref.permute(2, 1, 0).contiguous())
    g = rnd("pw.g", (N, RS, Cp)).cuda()
    dst = torch.ones((N, RS, Cm), device="cuda")
    chk(lib.dml_unpad_wgrad(g.data_ptr(), dst.data_ptr(), N, RS, Cm, Cp, st()))
    torch.cuda.synchronize()
    assert torch.allclose(dst.cpu(), 1 + g.cpu()[..., :Cm])


def test_pack_input(lib):
    x = rnd("pk", (2, 3, 7, 9)).cuda()
    for dt, tdt in ((0, torch.float32), (1, torch.bfloat16)):
        y = torch.empty((2, 7, 9, 8), device="cuda", dtype=tdt)
        chk(lib.dml_pack_input(x.data_ptr(), y.data_ptr(), 2, 3, 7, 9, 8, dt, st()))
        torch.cuda.synchronize()
        assert torch.equal(y[..., :3].float().cpu(), qz(x.cpu().permute(0, 2, 3, 1), tdt))
        assert (y[..., 3:] == 0).all()


@pytest.mark.parametrize("B,H,W,N", [(2, 20, 28, 64), (1, 6, 4, 8), (3, 34, 18, 24)])
def test_space_to_depth_stem_equals_the_7x7_stride2_convolution(lib, B, H, W, N):
    """dml_pack_input_s2d / dml_s2d_weights / dml_s2d_wgrad: the stem (backbone/resnet.py:139, nn.Conv2d(3, 64, 7, stride=2,
    padding=3, bias=False)) as a 4x4 stride-1 convolution on the space-to-depth image.  The regrouping itself is exact (pure
    copies, zero taps); forward and weight gradient through the library's fp32 kernels must match F.conv2d at fp32 level, including
    the image border (the regrouped filter's zero taps reach one row / column beyond it) and an unaligned image pointer."""
    from dmlnet._lib import ConvDesc, WgradDesc
    g = torch.Generator().manual_seed(B * 100 + H)
    buf = torch.randn(B * 3 * H * W + 1, generator=g).cuda()
    img = buf[1:].view(B, 3, H, W)                               # 4-byte aligned only
    wm = (torch.randn(N, 3, 7, 7, generator=g) * 0.1).cuda()
    wcl = wm.permute(0, 2, 3, 1).contiguous()                     # the parameter store's layout [N][7][7][3]
    H2, W2 = H // 2, W // 2
    x2 = torch.empty(B, H2, W2, 12, device="cuda")
    w2 = torch.full((N, 4, 4, 12), 9.0, device="cuda")
    chk(lib.dml_pack_input_s2d(img.data_ptr(), x2.data_ptr(), B, 3, H, W, st()))
    chk(lib.dml_s2d_weights(wcl.data_ptr(), w2.data_ptr(), N, 7, 3, st()))
    torch.cuda.synchronize()
    # the regrouping, element by element
    want = img.view(B, 3, H2, 2, W2, 2).permute(0, 2, 4, 3, 5, 1).reshape(B, H2, W2, 12)      # channel (dy, dx, c)
    assert torch.equal(x2, want)
    wz = torch.zeros(N, 8, 8, 3, device="cuda")
    wz[:, 1:, 1:, :] = wcl                                        # tap t = 2 r2 + dy - 1: one zero row / column in front
    assert torch.equal(w2, wz.view(N, 4, 2, 4, 2, 3).permute(0, 1, 3, 2, 4, 5).reshape(N, 4, 4, 12))
    # forward
    ref = torch.nn.functional.conv2d(img.double(), wm.double(), stride=2, padding=3)
    y = torch.empty(B, H2, W2, N, device="cuda")
    d = ConvDesc(x=x2.data_ptr(), w=w2.data_ptr(), y=y.data_ptr(), bias=None, stats=None, B=B, Hi=H2, Wi=W2, C=12, ldx=12, Ho=H2, Wo=W2,
                 N=N, ldy=N, R=4, S=4, stride=1, dil=1, pad=2, dtype=0, y_f32=0, accum=0, mode=0)
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    torch.cuda.synchronize()
    assert (y.permute(0, 3, 1, 2).double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    # weight gradient, added to whatever the parameter's gradient holds
    gy = torch.randn(ref.shape, generator=g, dtype=torch.float64).cuda()
    gref = torch.nn.grad.conv2d_weight(img.double(), wm.shape, gy, stride=2, padding=3)
    dyd = gy.float().permute(0, 2, 3, 1).contiguous()
    ws = torch.empty(1 << 22, device="cuda")
    for split in (0, 1):
        dw2 = torch.zeros(N, 4, 4, 12, device="cuda")
        wg = WgradDesc(x=x2.data_ptr(), dy=dyd.data_ptr(), dw=dw2.data_ptr(), B=B, Hi=H2, Wi=W2, C=12, ldx=12, Ho=H2, Wo=W2, N=N, ldy=N,
                       R=4, S=4, stride=1, dil=1, pad=2, dtype=0, splitk=0, Cm=12, ws=ws.data_ptr(), ws_elems=ws.numel(), f32_split=split)
        chk(lib.dml_conv_wgrad(C.byref(wg), st()))
        dw = torch.full((N, 7, 7, 3), 1.0, device="cuda")
        chk(lib.dml_s2d_wgrad(dw2.data_ptr(), dw.data_ptr(), N, 7, 3, st()))
        torch.cuda.synchronize()
        err = ((dw - 1.0).permute(0, 3, 1, 2).double() - gref).abs().max().item() / gref.abs().max().item()
        assert err <= 3e-6, (split, err)
    # odd image sizes / even kernels are refused
    assert lib.dml_pack_input_s2d(img.data_ptr(), x2.data_ptr(), B, 3, H - 1, W, st()) == -2
    assert lib.dml_s2d_weights(wcl.data_ptr(), w2.data_ptr(), N, 6, 3, st()) == -1
    assert lib.dml_s2d_wgrad(w2.data_ptr(), wcl.data_ptr(), N, 5, 3, st()) == -1       # (padding 2: even, no such regrouping)


@pytest.mark.parametrize("Cc", [72, 256])
def test_bn_finalize_many_groups(lib, Cc):
    """G >= 2048 row groups takes the folded two-stage finalize (the 192x192 / 384x384 layers); compare with fp64
    statistics, including a channel with |mean| >> std and a ragged last group."""
    M = 64 * 2500 + 17
    g = torch.Generator().manual_seed(5)
    y = torch.randn(M, Cc, generator=g) * (torch.rand(Cc, generator=g) * 2 + 0.01) + torch.randn(Cc, generator=g) * 3
    y[:, 1] = y[:, 1] * 1e-2 + 40.0
    yd = y.cuda()
    groups = (M + 63) // 64
    part = torch.zeros(groups * Cc * 2, device="cuda")
    chk(lib.dml_bn_stats(yd.data_ptr(), part.data_ptr(), M, Cc, Cc, 0, st()))
    sc, sh, mu, inv = (torch.empty(Cc, device="cuda") for _ in range(4))
    rm, rv = torch.zeros(Cc, device="cuda"), torch.ones(Cc, device="cuda")
    gam = (torch.rand(Cc, generator=g) + 0.5).cuda()
    chk(lib.dml_bn_finalize(part.data_ptr(), M, Cc, 64, gam.data_ptr(), None, rm.data_ptr(), rv.data_ptr(),
                            0.1, 1e-5, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), inv.data_ptr(), st()))
    torch.cuda.synchronize()
    y64 = y.double()
    mean, var = y64.mean(0), y64.var(0, unbiased=False)
    rt = torch.full((Cc,), 5e-6, dtype=torch.float64)
    rt[1] = 1e-3                      # fp32 group sums of a channel with mean/std ~ 1e4 carry ~1e-4 of its std

    def close(got, ref, what):
        bad = (got.cpu().double() - ref).abs() > rt * ref.abs() + 1e-12
        assert not bad.any(), "%s: channels %s" % (what, bad.nonzero().flatten().tolist()[:8])

    close(mu, mean, "mean")
    close(inv, 1 / torch.sqrt(var + 1e-5), "invstd")
    close(sc, gam.cpu().double() / torch.sqrt(var + 1e-5), "scale")
    assert (sh == 0).all()
    close(rm, 0.1 * mean, "running_mean")
    close(rv, 0.9 + 0.1 * y64.var(0, unbiased=True), "running_var")


@pytest.mark.parametrize("Cc,G", [(72, 9216), (256, 2049), (64, 2048)])
def test_bn_bwd_finalize_many_partial_rows(lib, Cc, G):
    """More than 2048 partial rows (a data gradient's fused sums on a 192 x 192 layer: one row per 64 pixels) are
    folded in two stages before the finalize; same coefficients and parameter gradients as fp64 sums."""
    g = torch.Generator().manual_seed(9)
    part = torch.randn(G, Cc, 2, generator=g)
    M = 64 * G - 5
    gam = torch.rand(Cc, generator=g) + 0.5
    mu, inv = torch.randn(Cc, generator=g), torch.rand(Cc, generator=g) + 0.5
    pd, dg, db, coef = part.cuda().contiguous(), torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda"), \
        torch.empty(4 * Cc, device="cuda")
    gam_d, mu_d, inv_d = gam.cuda(), mu.cuda(), inv.cuda()
    chk(lib.dml_bn_bwd_finalize(pd.data_ptr(), G, M, Cc, gam_d.data_ptr(), mu_d.data_ptr(), inv_d.data_ptr(),
                                dg.data_ptr(), db.data_ptr(), coef.data_ptr(), st()))
    torch.cuda.synchronize()
    s = part.double().sum(0)
    A = gam.double() * inv.double()
    ref = torch.stack([A, -A * inv.double() * s[:, 1] / M, -A * s[:, 0] / M, mu.double()])
    relclose(coef.view(4, Cc).cpu(), ref, 2e-6, "coef")
    relclose(db.cpu(), s[:, 0], 2e-6, "dbeta")
    relclose(dg.cpu(), s[:, 1], 2e-6, "dgamma")


@pytest.mark.parametrize("res", ["none", "f32", "planes"])
@pytest.mark.parametrize("shape", [(3 * 9 * 11, 48), (2 * 37 * 5, 264)])
def test_bn_apply_planes_only_output_equals_the_fp32_path(lib, res, shape):
    """dml_bn_apply with z == NULL (the output exists as fp16 planes only: bn_apply_planes8_kernel, eight channels per thread) against
    the same launch that also writes the fp32 tensor (bn_apply_cols_kernel): planes, ReLU mask bytes and max |z| bit for bit; and the
    residual operand given as the planes of a tensor (res_unscale, ABI 4) = the launch with that tensor's (hi + lo) / s as fp32."""
    M, Cc = shape
    g = torch.Generator(device="cpu").manual_seed(5)
    yd = (torch.randn(M, Cc, generator=g) * 2 + 0.3).cuda()
    sc, sh, mu = (torch.rand(Cc, generator=g) + 0.5).cuda(), (torch.randn(Cc, generator=g) * 0.1).cuda(), (torch.randn(Cc, generator=g) * 0.2).cuda()
    rfull = (torch.randn(M, Cc, generator=g) * 1.5).cuda()
    rp, rw = h2_planes(lib, rfull, 0)
    r_from_planes = ((rp[0].float() + rp[1].float()) * rw[1024]).view(M, Cc).contiguous()      # what the planes represent, exactly
    work = torch.zeros(1025, device="cuda")
    work[1024] = 2.0 ** -9
    out = {}
    for only in (False, True):
        pz = torch.zeros(2, M * Cc, device="cuda", dtype=torch.float16)
        zd = torch.full((M, Cc), float("nan"), device="cuda")
        mk = torch.zeros(M * Cc // 4, device="cuda", dtype=torch.uint8)
        amax = torch.zeros(1024, device="cuda")
        if res == "planes" and only:
            rarg, rps, run = rp.data_ptr(), M * Cc, rw.data_ptr() + 4096
        else:
            rsrc = None if res == "none" else (rfull if res == "f32" else r_from_planes)
            rarg, rps, run = (rsrc.data_ptr() if rsrc is not None else None), 0, None
        chk(lib.dml_bn_apply(yd.data_ptr(), rarg, None if only else zd.data_ptr(), sc.data_ptr(), sh.data_ptr(), mu.data_ptr(),
                             mk.data_ptr(), M, Cc, Cc, Cc, Cc, 1, 0, 0.0, 0, amax.data_ptr(), pz.data_ptr(), M * Cc, Cc,
                             work.data_ptr() + 4096, rps, run, st()))
        torch.cuda.synchronize()
        out[only] = (pz, mk, amax.max().item(), zd)
    ref = torch.relu((yd - mu) * sc + sh + (0 if res == "none" else (rfull if res == "f32" else r_from_planes)))
    assert (out[False][3] - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    assert torch.equal(out[True][0], out[False][0]) and torch.equal(out[True][1], out[False][1]) and out[True][2] == out[False][2]
    assert torch.isnan(out[True][3]).all()            # z really was not written


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("relu,res,drop", [(1, False, 0.0), (1, True, 0.0), (0, False, 0.0), (1, False, 0.25)])
def test_bn_train_fwd_bwd(lib, dname, relu, res, drop):
    """bn_stats -> finalize -> apply and the two-pass backward vs F.batch_norm autograd."""
    dt, tdt, tol = DT[dname]
    B, Hh, Ww, Cc = 3, 9, 11, 48
    M = B * Hh * Ww
    y = qz(rnd("bn.y", (B, Cc, Hh, Ww), 2.0) + 0.5, tdt).requires_grad_(True)
    r = qz(rnd("bn.r", (B, Cc, Hh, Ww)), tdt).requires_grad_(True) if res else None
    gamma = (rnd("bn.g", (Cc,)) * 0.2 + 1).requires_grad_(True)
    beta = (rnd("bn.b", (Cc,)) * 0.1).requires_grad_(True)
    rm0, rv0 = rnd("bn.rm", (Cc,)) * 0.1, rnd("bn.rv", (Cc,)).abs() + 0.5
    rm_ref, rv_ref = rm0.clone(), rv0.clone()
    o = F.batch_norm(y, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.01, eps=1e-5)
    if res:
        o = o + r
    if relu:
        o = F.relu(o)
    yd = nhwc(y.detach(), tdt)
    rd = nhwc(r.detach(), tdt) if res else None
    zd = torch.empty_like(yd)
    groups = (M + 63) // 64
    part = torch.zeros(max(groups, 1100) * Cc * 2, device="cuda")
    chk(lib.dml_bn_stats(yd.data_ptr(), part.data_ptr(), M, Cc, Cc, dt, st()))
    sc, sh, mu, inv = (torch.empty(Cc, device="cuda") for _ in range(4))
    rm, rv, g_d, b_d = rm0.cuda(), rv0.cuda(), gamma.detach().cuda(), beta.detach().cuda()
    chk(lib.dml_bn_finalize(part.data_ptr(), M, Cc, 64, g_d.data_ptr(), b_d.data_ptr(), rm.data_ptr(), rv.data_ptr(),
                            0.01, 1e-5, sc.data_ptr(), sh.data_ptr(), mu.data_ptr(), inv.data_ptr(), st()))
    # 1-bit ReLU mask, one byte per 16-byte vector (8 elements in bf16, 4 in fp32); every other fp32 case keeps reading z
    use_mask = dname == "bf16" or (res and relu)
    bitmask = torch.zeros(M * Cc // (8 if dname == "bf16" else 4), device="cuda", dtype=torch.uint8) if use_mask else None
    mk = bitmask.data_ptr() if bitmask is not None else None
    amax = torch.zeros(1024, device="cuda")
    # fp32: the output also as fp16 hi / lo planes, scaled from the bound dml_h2_bound_bn derives from gamma / beta / count
    # (+ max |res| from the residual's amax words, x the dropout's 1 / (1 - p))
    planes_ok = dname == "f32" and Cc % 4 == 0
    pz, zwork, rwork = None, torch.zeros(1025, device="cuda"), torch.zeros(1024, device="cuda")
    if planes_ok:
        pz = torch.zeros(2, M * Cc, device="cuda", dtype=torch.float16)
        if res:
            rwork[17] = rd.abs().max()
        chk(lib.dml_h2_bound_bn(g_d.data_ptr(), b_d.data_ptr(), Cc, M, 1.0 / (1.0 - drop), rwork.data_ptr() if res else None,
                                zwork.data_ptr(), st()))
    chk(lib.dml_bn_apply(yd.data_ptr(), rd.data_ptr() if res else None, zd.data_ptr(), sc.data_ptr(), sh.data_ptr(),
                         mu.data_ptr(), mk, M, Cc, Cc, Cc, Cc, relu, dt, drop, 1234, amax.data_ptr(),
                         pz.data_ptr() if planes_ok else None, M * Cc, Cc, zwork.data_ptr() + 4096 if planes_ok else None, 0, None, st()))
    torch.cuda.synchronize()
    if planes_ok:
        un = zwork[1024].item()
        bound = (gamma.detach().abs() * M ** 0.5 + beta.detach().abs()).max().item() / (1.0 - drop) + (rd.abs().max().item() if res else 0.0)
        zmax = zd.abs().max().item()
        assert np.log2(un) == np.round(np.log2(un)) and zmax / un < 2.0 ** 15, (un, zmax)
        assert 2.0 ** 14 <= bound * 1.002 / un and bound / un < 2.0 ** 15, (bound, un)
        rec = (pz[0].double() + pz[1].double()) * un
        err = (rec - zd.double().view(-1)).abs().max().item()
        # hi + lo resolves 2^-22 of an element down to ~2^-11 of the bound, 2^-33 of the bound below
        assert err <= 2.0 ** -21 * zmax + 2.0 ** -32 * bound, (err, zmax, bound)
        assert torch.isfinite(pz.float()).all()
    # (taken from the fp32 value before a bf16 store rounds it)
    assert abs(amax.max().item() - zd.float().abs().max().item()) <= 2.0 ** -7 * amax.max().item(), "amax side output of dml_bn_apply"
    relclose(rm.cpu(), rm_ref, 1e-4, "running_mean")
    relclose(rv.cpu(), rv_ref, 1e-4, "running_var")
    z = nchw(zd)
    gs = 1.0
    if drop > 0:
        keep = (z != 0) | (o.detach() == 0)
        frac = 1 - (z != 0).float().sum() / ((o.detach() != 0).float().sum() + 1e-9)
        assert abs(frac.item() - drop) < 0.03, frac
        gs = 1 / (1 - drop)
        relclose(z[z != 0], (o.detach() * gs)[z != 0], max(tol, 1e-5), "dropout kept values")
        mask = (z != 0).float()
    else:
        relclose(z, o.detach(), max(tol, 1e-5), "bn apply")
        mask = None
    # backward
    gz = qz(rnd("bn.gz", (B, Cc, Hh, Ww)), tdt)
    (o * (gz * (mask * gs if mask is not None else 1))).sum().backward()
    gzd = nhwc(gz, tdt)
    nblk = C.c_int(0)
    zarg = None if mk else zd.data_ptr()          # with the bitmask the backward never touches z
    gwork, dwork = torch.zeros(1024, device="cuda"), torch.zeros(1025, device="cuda")
    chk(lib.dml_bn_bwd_reduce(gzd.data_ptr(), yd.data_ptr(), zarg, mk, mu.data_ptr(), inv.data_ptr(),
                              part.data_ptr(), M, Cc, Cc, Cc, Cc, 1 if (relu or drop > 0) else 0, gs, dt,
                              C.byref(nblk), gwork.data_ptr(), st()))
    coef = torch.empty(4 * Cc, device="cuda")
    dg, db = torch.zeros(Cc, device="cuda"), torch.zeros(Cc, device="cuda")
    chk(lib.dml_bn_bwd_finalize(part.data_ptr(), nblk, M, Cc, g_d.data_ptr(), mu.data_ptr(), inv.data_ptr(),
                                dg.data_ptr(), db.data_ptr(), coef.data_ptr(), st()))
    dyd = torch.empty_like(yd)
    dres = torch.empty_like(yd) if res else None
    amax_dy = torch.zeros(1024, device="cuda")
    pdy = torch.zeros(2, M * Cc, device="cuda", dtype=torch.float16) if planes_ok else None
    if planes_ok:
        chk(lib.dml_h2_bound_bn_bwd(coef.data_ptr(), inv.data_ptr(), Cc, M, gwork.data_ptr(), dwork.data_ptr(), st()))
    chk(lib.dml_bn_bwd_apply(gzd.data_ptr(), yd.data_ptr(), zarg, mk, coef.data_ptr(), dyd.data_ptr(),
                             dres.data_ptr() if res else None, M, Cc, Cc, Cc, Cc, Cc, Cc,
                             1 if (relu or drop > 0) else 0, gs, 0, dt, amax_dy.data_ptr(),
                             pdy.data_ptr() if planes_ok else None, M * Cc, Cc, dwork.data_ptr() + 4096 if planes_ok else None, st()))
    torch.cuda.synchronize()
    if planes_ok:
        un, dmax = dwork[1024].item(), dyd.abs().max().item()
        assert np.log2(un) == np.round(np.log2(un)) and dmax / un < 2.0 ** 15, (un, dmax)
        assert dmax / un >= 2.0 ** 3, "the bound of dml_h2_bound_bn_bwd is more than 2^12 above max |dy|: %g" % (dmax / un)
        rec = (pdy[0].double() + pdy[1].double()) * un
        err = (rec - dyd.double().view(-1)).abs().max().item()
        assert err <= 2.0 ** -21 * dmax + 2.0 ** -32 * un * 2.0 ** 15, (err, dmax, un)
        # planes only (dy == NULL) writes the same planes
        # (with the bit mask: bn_bwd_apply_planes8_kernel, eight channels per thread; the identity branch's gradient rides along)
        pdy2 = torch.zeros_like(pdy)
        dres2 = torch.full_like(dres, float("nan")) if res else None
        chk(lib.dml_bn_bwd_apply(gzd.data_ptr(), yd.data_ptr(), zarg, mk, coef.data_ptr(), None, dres2.data_ptr() if res else None,
                                 M, Cc, Cc, Cc, Cc, Cc, Cc if res else 0,
                                 1 if (relu or drop > 0) else 0, gs, 0, dt, None, pdy2.data_ptr(), M * Cc, Cc, dwork.data_ptr() + 4096, st()))
        torch.cuda.synchronize()
        assert torch.equal(pdy2, pdy)
        if res:
            assert torch.equal(dres2, dres)
    assert abs(amax_dy.max().item() - dyd.float().abs().max().item()) <= 2.0 ** -7 * amax_dy.max().item()
    btol = 1e-4 if dname == "f32" else 2e-2
    relclose(dg.cpu(), gamma.grad, btol, "dgamma")
    relclose(db.cpu(), beta.grad, btol, "dbeta")
    relclose(nchw(dyd), y.grad, btol, "bn dy")
    if res:
        relclose(nchw(dres), r.grad, btol, "bn dres")


def test_bn_eval_coeffs(lib):
    Cc = 40
    g, b, rm, rv = (rnd("ev" + s, (Cc,)).cuda() for s in "gbmv")
    rv = rv.abs() + 0.3
    sc, sh = torch.empty(Cc, device="cuda"), torch.empty(Cc, device="cuda")
    chk(lib.dml_bn_eval_coeffs(g.data_ptr(), b.data_ptr(), rm.data_ptr(), rv.data_ptr(), 1e-5, sc.data_ptr(),
                               sh.data_ptr(), Cc, st()))
    torch.cuda.synchronize()
    x = rnd("ev.x", (2, Cc, 3, 3))
    ref = F.batch_norm(x, rm.cpu(), rv.cpu(), g.cpu(), b.cpu(), training=False, eps=1e-5)
    got = (x - rm.cpu().view(1, -1, 1, 1)) * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1)
    relclose(got, ref, 1e-5, "eval bn")


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(2, 13, 10, 16), (1, 12, 16, 64), (2, 7, 9, 8), (1, 1, 1, 8), (1, 2, 5, 8)])
def test_maxpool(lib, dname, geom):
    dt, tdt, tol = DT[dname]
    B, Hh, Ww, Cc = geom
    x = qz(rnd("mp.x", (B, Cc, Hh, Ww)), tdt).requires_grad_(True)
    y = F.max_pool2d(x, 3, 2, 1)
    gy = qz(rnd("mp.g", tuple(y.shape)), tdt)
    y.backward(gy)
    Ho, Wo = y.shape[2:]
    xd = nhwc(x.detach(), tdt)
    yd = torch.empty((B, Ho, Wo, Cc), device="cuda", dtype=tdt)
    am = torch.empty((B, Ho, Wo, Cc), device="cuda", dtype=torch.uint8)
    chk(lib.dml_maxpool3x3s2_fwd(xd.data_ptr(), yd.data_ptr(), am.data_ptr(), B, Hh, Ww, Cc, dt, st()))
    dxd = torch.empty_like(xd)
    gyd = nhwc(gy, tdt)
    chk(lib.dml_maxpool3x3s2_bwd(gyd.data_ptr(), am.data_ptr(), dxd.data_ptr(), B, Hh, Ww, Cc, dt, st()))
    torch.cuda.synchronize()
    assert torch.equal(nchw(yd), y.detach())
    relclose(nchw(dxd), x.grad, tol, "maxpool bwd")


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("geom", [(3, 35, 64, 96), (2, 301, 256, 1280), (2, 2304, 48, 64)])
def test_avgpool_broadcast_reduce(lib, dname, geom):
    dt, tdt, tol = DT[dname]
    B, HW, Cc, ld = geom
    x = qz(rnd("ap.x", (B, HW, ld)), tdt)
    xd = x.to("cuda", tdt)
    out = torch.empty((B, Cc), device="cuda", dtype=tdt)
    chk(lib.dml_global_avgpool_fwd(xd.data_ptr() + 16 * xd.element_size(), out.data_ptr(), B, HW, Cc, ld, dt, st()))
    red = torch.empty((B, Cc), device="cuda", dtype=tdt)
    chk(lib.dml_reduce_hw(xd.data_ptr() + 16 * xd.element_size(), red.data_ptr(), B, HW, Cc, ld, dt, st()))
    z = torch.zeros((B, HW, ld), device="cuda", dtype=tdt)
    chk(lib.dml_broadcast_hw(out.data_ptr(), z.data_ptr() + 8 * z.element_size(), B, HW, Cc, ld, dt, st()))
    acc = xd.clone()
    chk(lib.dml_avgpool_bwd_add(out.data_ptr(), acc.data_ptr() + 16 * acc.element_size(), B, HW, Cc, ld, dt, st()))
    torch.cuda.synchronize()
    ref = x[:, :, 16:16 + Cc].mean(1)
    relclose(out.float().cpu(), ref, max(tol, 1e-5), "avgpool")
    relclose(red.float().cpu(), ref * HW, max(tol, 1e-5), "reduce_hw")
    assert torch.equal(z[:, :, 8:8 + Cc].float().cpu(), out.float().cpu().unsqueeze(1).expand(B, HW, Cc))
    assert (z[:, :, :8] == 0).all() and (z[:, :, 8 + Cc:] == 0).all()
    relclose(acc[:, :, 16:16 + Cc].float().cpu(),
             x[:, :, 16:16 + Cc] + out.float().cpu().unsqueeze(1) / HW, max(tol, 1e-5), "avgpool bwd add")


@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("shape", [(5, 7, 20, 28), (4, 4, 16, 16), (5, 7, 13, 17), (1, 1, 6, 5)])
def test_bilinear(lib, dname, shape):
    dt, tdt, tol = DT[dname]
    h, w, Hh, Ww = shape
    B, Cc = 2, 8
    x = qz(rnd("bl.x", (B, Cc, h, w)), tdt).requires_grad_(True)
    y = F.interpolate(x, size=(Hh, Ww), mode="bilinear", align_corners=False)
    gy = qz(rnd("bl.g", tuple(y.shape)), tdt)
    y.backward(gy)
    xd = nhwc(x.detach(), tdt)
    yd = torch.empty((B, Hh, Ww, Cc), device="cuda", dtype=tdt)
    chk(lib.dml_bilinear_fwd(xd.data_ptr(), yd.data_ptr(), B, h, w, Hh, Ww, Cc, Cc, Cc, dt, 0, 0, st()))
    gyd = nhwc(gy, tdt)
    dxd = torch.empty_like(xd)
    chk(lib.dml_bilinear_bwd(gyd.data_ptr(), dxd.data_ptr(), B, h, w, Hh, Ww, Cc, Cc, Cc, dt, 0, 0, st()))
    # fp32 gradient in, storage-dtype gradient out (final upsample backward)
    gyf = gy.permute(0, 2, 3, 1).contiguous().cuda()
    dx2 = torch.empty_like(xd)
    chk(lib.dml_bilinear_bwd(gyf.data_ptr(), dx2.data_ptr(), B, h, w, Hh, Ww, Cc, Cc, Cc, dt, 1, 0, st()))
    torch.cuda.synchronize()
    relclose(nchw(yd), y.detach(), max(tol, 1e-6), "bilinear fwd")
    relclose(nchw(dxd), x.grad, max(tol, 1e-5), "bilinear bwd")
    relclose(nchw(dx2), x.grad, max(tol, 1e-5), "bilinear bwd f32 in")


def test_bilinear_golden_g6(lib):
    g = H.load_golden("g6_bilinear")
    a = torch.from_numpy(g["a"])
    xd = nhwc(a, torch.float32)
    yd = torch.empty((1, 20, 28, 3 + 1), device="cuda")          # C must be a multiple of 4: pad one channel
    xp = torch.zeros((1, 5, 7, 4), device="cuda")
    xp[..., :3] = xd
    chk(lib.dml_bilinear_fwd(xp.data_ptr(), yd.data_ptr(), 1, 5, 7, 20, 28, 4, 4, 4, 0, 0, 0, st()))
    torch.cuda.synchronize()
    relclose(nchw(yd[..., :3]), torch.from_numpy(g["ua"]), 1e-6, "G6 x4")


def test_distance_head_golden_g1(lib):
    g = H.load_golden("g1_distance_head")
    x = torch.from_numpy(g["x"]).cuda()
    B, Cc, Hh, Ww = x.shape
    for protos, key in ((torch.from_numpy(g["centers"]), "logits"), (torch.from_numpy(g["protos"]), "logits_general")):
        K = protos.shape[0]
        pr = protos.cuda().contiguous()
        lg = torch.empty((B, K, Hh, Ww), device="cuda")
        ft = torch.empty((B, Hh, Ww, Cc), device="cuda")
        am = torch.empty((B, Hh, Ww), device="cuda", dtype=torch.uint8)
        ds = torch.empty((B, Hh, Ww), device="cuda")
        chk(lib.dml_proto_dist_fwd(x.data_ptr(), pr.data_ptr(), lg.data_ptr(), ft.data_ptr(), am.data_ptr(),
                                   ds.data_ptr(), B, Cc, K, Hh, Ww, st()))
        torch.cuda.synchronize()
        ref = torch.from_numpy(g[key])
        relclose(lg.cpu(), ref, 1e-5, "G1 " + key)
        assert torch.equal(ft.cpu(), torch.from_numpy(g["features"]))
        assert torch.equal(am.cpu().long(), ref.argmax(1))
        relclose(ds.cpu(), -ref.sum(1), 1e-5, "dissum raw")


def test_upsample_dist_and_bwd(lib):
    B, h, w, K = 2, 6, 5, 16
    Hh, Ww = 24, 20
    e = rnd("ud.e", (B, K, h, w), 2.0).requires_grad_(True)
    up = F.interpolate(e, size=(Hh, Ww), mode="bilinear", align_corners=False)
    feats = up.permute(0, 2, 3, 1)
    protos = 3.0 * torch.eye(K)
    logits = -((feats.unsqueeze(3) - protos) ** 2).sum(-1).permute(0, 3, 1, 2)
    gl = rnd("ud.gl", (B, K, Hh, Ww))
    gf = rnd("ud.gf", (B, Hh, Ww, K))
    ((logits * gl).sum() + (feats * gf).sum()).backward()
    ed = e.detach().permute(0, 2, 3, 1).contiguous().cuda()
    pr = protos.cuda()
    lg = torch.empty((B, K, Hh, Ww), device="cuda")
    ft = torch.empty((B, Hh, Ww, K), device="cuda")
    chk(lib.dml_upsample_dist_fwd(ed.data_ptr(), pr.data_ptr(), lg.data_ptr(), ft.data_ptr(), None, None, B, h, w, K,
                                  K, Hh, Ww, st()))
    df = torch.empty((B, Hh, Ww, K), device="cuda")
    gld, gfd = gl.cuda(), gf.cuda()
    chk(lib.dml_proto_dist_bwd(gld.data_ptr(), gfd.data_ptr(), ft.data_ptr(), pr.data_ptr(), df.data_ptr(), B, K, K,
                               Hh, Ww, st()))
    de = torch.empty((B, h, w, K), device="cuda")
    chk(lib.dml_bilinear_bwd(df.data_ptr(), de.data_ptr(), B, h, w, Hh, Ww, K, K, K, 0, 1, 1, st()))
    torch.cuda.synchronize()
    relclose(lg.cpu(), logits.detach(), 1e-5, "fused upsample+dist logits")
    relclose(ft.cpu(), feats.detach(), 1e-6, "fused features")
    relclose(de.cpu().permute(0, 3, 1, 2), e.grad, 1e-5, "head backward to the embedding")


def test_loss_golden_g2(lib):
    import utils
    g = H.load_golden("g2_losses")
    lo = torch.from_numpy(g["logit"]).cuda().requires_grad_(True)
    loss = utils.DMLLoss(alpha=0.01, ignore_index=-1)(lo, torch.from_numpy(g["label"]).cuda())
    loss.backward()
    assert abs(loss.item() - float(g["loss"])) <= 1e-5 * abs(float(g["loss"]))
    relclose(lo.grad.cpu(), torch.from_numpy(g["grad"]), 1e-5, "DML grad")
    lo2 = torch.from_numpy(g["logit2"]).cuda().requires_grad_(True)
    l2 = utils.CrossEntropyLoss(ignore_index=255, alpha=0, beta=0, gamma=0)(lo2, torch.from_numpy(g["label2"]).cuda(),
                                                                            None)
    (l2 * 2.0).backward()
    assert abs(l2.item() - float(g["loss2"])) <= 1e-5 * abs(float(g["loss2"]))
    relclose(lo2.grad.cpu(), 2 * torch.from_numpy(g["grad2"]), 1e-5, "CE/n grad (gout = 2)")


def test_scores_golden_g7(lib):
    import utils
    g = H.load_golden("g7_scoring")
    lg = torch.from_numpy(g["logits"]).cuda()
    ft = torch.from_numpy(g["feats"]).cuda()
    relclose(utils.dissum_score(lg, 1000.0, False).cpu()[0], torch.from_numpy(g["dissum_deeplab"]), 1e-5, "dissum")
    relclose(utils.dissum_score(lg, 400.0, True).cpu()[0], torch.from_numpy(g["dissum_anomaly"]), 1e-5, "dissum400")
    preds, msp = utils.argmax_msp(lg)
    assert torch.equal(preds.cpu(), torch.from_numpy(g["preds"]))
    relclose(msp.cpu(), torch.from_numpy(g["msp"]), 1e-5, "msp")
    proto = utils.mean_prototype(g["shots"])
    assert np.allclose(proto, g["proto"])
    rel = utils.novel_relabel(preds, lg, ft, proto, -1.5, 16)
    assert torch.equal(rel.cpu(), torch.from_numpy(g["relabel"]))


def test_sgd_matches_torch(lib):
    n = 1000 * 4 + 3
    p0, g0 = rnd("sgd.p", (n,)), rnd("sgd.g", (n,))
    p_ref = p0.clone().requires_grad_(True)
    opt = torch.optim.SGD([p_ref], lr=0.05, momentum=0.9, weight_decay=1e-4)
    pd, vd = p0.cuda(), torch.zeros(n, device="cuda")
    for it in range(3):
        gi = g0 * (it + 1)
        p_ref.grad = gi.clone()
        opt.step()
        gd = gi.cuda()
        chk(lib.dml_sgd_step(pd.data_ptr(), gd.data_ptr(), vd.data_ptr(), n, 0.05, 0.9, 1e-4, 1.0, st()))
    torch.cuda.synchronize()
    relclose(pd.cpu(), p_ref.detach(), 1e-6, "sgd params")


@pytest.mark.parametrize("M,N", [(1000, 16), (70001, 16), (5003, 12)])
def test_bias_grad(lib, M, N):
    """atomics form and the workspace form (per-workgroup partials + fixed-order fold: bitwise reproducible); vector and scalar
    kernels (N = 12 is not a multiple of the 16-byte vector)"""
    dy = rnd("bg%d" % M, (M, N))
    for dt, tdt in ((0, torch.float32), (1, torch.bfloat16)):
        d = dy.to("cuda", tdt)
        db = torch.ones(N, device="cuda")
        chk(lib.dml_bias_grad(d.data_ptr(), db.data_ptr(), M, N, N, dt, st()))
        torch.cuda.synchronize()
        relclose(db.cpu(), 1 + qz(dy, tdt).sum(0), 1e-4, "bias grad")
        ws = torch.empty(1024 * N, device="cuda")
        outs = []
        for _ in range(3):
            db = torch.ones(N, device="cuda")
            ws.fill_(float("nan"))
            chk(lib.dml_bias_grad_ws(d.data_ptr(), db.data_ptr(), M, N, N, dt, ws.data_ptr(), ws.numel(), st()))
            torch.cuda.synchronize()
            outs.append(db.cpu())
        relclose(outs[0], 1 + qz(dy, tdt).sum(0), 1e-4, "bias grad (workspace)")
        assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert lib.dml_bias_grad_ws(d.data_ptr(), db.data_ptr(), M, N, N, dt, None, 0, st()) != 0


def test_rejects_bad_arguments(lib):
    from dmlnet._lib import ConvDesc
    d = ConvDesc()
    assert lib.dml_conv_igemm(C.byref(d), None) == -1
    x = torch.zeros(64, device="cuda")
    assert lib.dml_bn_apply(x.data_ptr(), None, x.data_ptr(), x.data_ptr(), x.data_ptr(), x.data_ptr(), None, 4, 6, 6, 6,
                            6, 1, 0, 0.0, 0, None, None, 0, 0, None, 0, None, st()) == -2
    assert lib.dml_proto_dist_fwd(x.data_ptr(), x.data_ptr(), None, None, None, None, 1, 64, 16, 1, 1, st()) == -3


@pytest.mark.parametrize("geom", [(2, 13, 20), (1, 7, 14), (3, 29, 5), (1, 192, 192)])
@pytest.mark.parametrize("dname", ["f32", "bf16"])
@pytest.mark.parametrize("proto_kind", ["general", "3I", "diagonal"])
def test_head_backward_fused_vs_plain_torch(lib, geom, dname, proto_kind):
    """dml_head_bwd_fused (loss gradient + distance-head gradient + transposed x4 upsample in one pass) against autograd
    through the reference's arithmetic (network/utils.py:88-118 + anomaly/models/models.py:42-78: interpolate ->
    distances -> CE/n + alpha VAR/n) on the CPU, and against the unfused kernel chain.  Tiles are 7 x 14 low-resolution
    pixels: the geometries cover ragged tiles in both directions, single-tile maps and every border clamp."""
    from dmlnet import _lib as L
    from oracle import dmlnet_ref as O
    dt, tdt, _ = DT[dname]
    B, h, w = geom
    Hh, Ww = 4 * h, 4 * w
    emb = rnd("hb.emb%d%d" % (h, w), (B, 16, h, w), 1.5).requires_grad_(True)
    lab = H.synth_labels(5, "hb.lab%d%d" % (h, w), (B, Hh, Ww), 16, 255, ignore_frac=0.1)
    # general matrix: the kernel's general path; 3 * I (the reference's centers) and another diagonal: its closed-form path
    if proto_kind == "general":
        protos = O.prototypes_3I(16) + 0.05 * rnd("hb.pr", (16, 16))
    elif proto_kind == "3I":
        protos = O.prototypes_3I(16)
    else:
        protos = torch.diag(3.0 + rnd("hb.prd", (16,)))
    up = F.interpolate(emb, size=(Hh, Ww), mode="bilinear", align_corners=False)
    logits, _, feats = O.distance_head(up, protos)
    loss = O.dml_loss(logits, lab, alpha=0.01, ignore_index=255) * 1.7            # gout = 1.7
    loss.backward()
    ref = emb.grad.permute(0, 2, 3, 1).contiguous()                                # [B,h,w,16]
    # device side: features / logits as the forward kernels leave them
    f_d = feats.detach().cuda().contiguous()
    lg_d = logits.detach().cuda().contiguous()
    lab_d, pr_d = lab.cuda(), protos.cuda().contiguous()
    sums = torch.zeros(5, dtype=torch.float64, device="cuda")
    part = torch.empty(L.LOSS_BLOCKS * 4, dtype=torch.float32, device="cuda")
    chk(lib.dml_loss_fwd(lg_d.data_ptr(), lab_d.data_ptr(), sums.data_ptr(), part.data_ptr(), B, 16, Hh, Ww, 255, st()))
    gout = torch.tensor(1.7, device="cuda")
    de = torch.empty(B, h, w, 16, dtype=tdt, device="cuda")
    chk(lib.dml_head_bwd_fused(f_d.data_ptr(), lab_d.data_ptr(), sums.data_ptr(), gout.data_ptr(), pr_d.data_ptr(),
                               de.data_ptr(), B, h, w, 16, 16, Hh, Ww, 255, 0.01, float(B), dt, st()))
    relclose(de.float().cpu(), ref, 2e-5 if dname == "f32" else 2.0 ** -8 * 1.05, "fused head backward vs autograd")
    # the unfused chain on the same inputs
    gl = torch.empty_like(lg_d)
    df = torch.empty_like(f_d)
    de2 = torch.empty_like(de)
    chk(lib.dml_loss_bwd(lg_d.data_ptr(), lab_d.data_ptr(), sums.data_ptr(), gout.data_ptr(), gl.data_ptr(), B, 16, Hh, Ww,
                         255, 0.01, float(B), st()))
    chk(lib.dml_proto_dist_bwd(gl.data_ptr(), None, f_d.data_ptr(), pr_d.data_ptr(), df.data_ptr(), B, 16, 16, Hh, Ww, st()))
    chk(lib.dml_bilinear_bwd(df.data_ptr(), de2.data_ptr(), B, h, w, Hh, Ww, 16, 16, 16, dt, 1, 0, st()))
    relclose(de.float().cpu(), de2.float().cpu(), 2e-5 if dname == "f32" else 2.0 ** -7, "fused vs unfused chain")
    # shapes it does not cover are refused, not mis-computed
    assert lib.dml_head_bwd_fused(f_d.data_ptr(), lab_d.data_ptr(), sums.data_ptr(), gout.data_ptr(), pr_d.data_ptr(),
                                  de.data_ptr(), B, h, w, 16, 16, Hh + 1, Ww, 255, 0.01, float(B), dt, st()) == -3


def test_conv_wgrad_group_matches_the_per_layer_launches(lib):
    """dml_conv_wgrad_group: the weight gradients of several layers in one launch (a few split-K slabs each instead of
    ~28) -- a bottleneck's three convolutions with different map sizes, the dropped-pad-channel case and a stride-2 job,
    against autograd on the CPU; an ineligible job makes the call fail, not fall back."""
    from dmlnet._lib import WgradDesc
    cases = [("g1x1a", 2, 16, 16, 256, 1024, 1, 1, 1, 0), ("g3x3", 2, 16, 16, 256, 256, 3, 1, 1, 0),
             ("g1x1b", 2, 16, 16, 1024, 256, 1, 1, 1, 0), ("g320", 1, 40, 44, 320, 256, 3, 1, 1, 304),
             ("gs2", 2, 40, 36, 128, 256, 3, 2, 1, 0), ("gd6", 3, 12, 16, 512, 256, 3, 1, 6, 0)]
    keep, descs, refs, outs = [], [], [], []
    for name, B, Hh, Ww, Cin, Cout, k, stride, dil, Cm in cases:
        x = qz(rnd("wgg.x" + name, (B, Cin, Hh, Ww)), torch.bfloat16)
        w = qz(rnd("wgg.w" + name, (Cout, Cin, k, k), scale=0.05), torch.bfloat16).requires_grad_(True)
        y, pad = conv_ref(x, w, k, stride, dil)
        gy = qz(rnd("wgg.gy" + name, tuple(y.shape)), torch.bfloat16)
        (y * gy).sum().backward()
        xd, gyd = nhwc(x, torch.bfloat16), nhwc(gy, torch.bfloat16)
        cm = Cm or Cin
        dw = torch.full((Cout, k, k, cm), 1.0, device="cuda")
        d = WgradDesc(x=xd.data_ptr(), dy=gyd.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hh, Wi=Ww, C=Cin, ldx=Cin, Ho=y.shape[2],
                      Wo=y.shape[3], N=Cout, ldy=Cout, R=k, S=k, stride=stride, dil=dil, pad=pad, dtype=1, splitk=0, Cm=Cm,
                      ws=None, ws_elems=0)
        assert lib.dml_conv_wgrad_group_eligible(C.byref(d)) == 1, name
        keep += [xd, gyd]
        descs.append(d)
        refs.append(w.grad[:, :cm])
        outs.append(dw)
    ws = torch.empty(48 << 20, device="cuda")
    arr = (C.c_void_p * len(descs))(*[C.addressof(d) for d in descs])
    chk(lib.dml_conv_wgrad_group(arr, len(descs), ws.data_ptr(), ws.numel(), st()))
    torch.cuda.synchronize()
    for (name, *_), dw, ref in zip(cases, outs, refs):
        relclose(dw.cpu().permute(0, 3, 1, 2) - 1.0, ref, 2e-3, "grouped wgrad " + name)
    # subsets and single jobs give the same result (other split choices)
    for sub in ([0], [1, 2], [3, 4, 5]):
        for i in sub:
            outs[i].fill_(0.0)
        arr2 = (C.c_void_p * len(sub))(*[C.addressof(descs[i]) for i in sub])
        chk(lib.dml_conv_wgrad_group(arr2, len(sub), ws.data_ptr(), ws.numel(), st()))
        torch.cuda.synchronize()
        for i in sub:
            relclose(outs[i].cpu().permute(0, 3, 1, 2), refs[i], 2e-3, "grouped wgrad subset %s" % cases[i][0])
    # a 64-channel job is not eligible: refused
    small = WgradDesc(x=keep[0].data_ptr(), dy=keep[1].data_ptr(), dw=outs[0].data_ptr(), B=2, Hi=16, Wi=16, C=256, ldx=256,
                      Ho=16, Wo=16, N=64, ldy=1024, R=1, S=1, stride=1, dil=1, pad=0, dtype=1, splitk=0, Cm=0, ws=None, ws_elems=0)
    assert lib.dml_conv_wgrad_group_eligible(C.byref(small)) == 0
    arr3 = (C.c_void_p * 1)(C.addressof(small))
    assert lib.dml_conv_wgrad_group(arr3, 1, ws.data_ptr(), ws.numel(), st()) == -3
    # too small a workspace is an error, not an overrun
    assert lib.dml_conv_wgrad_group(arr, len(descs), ws.data_ptr(), 1024, st()) == -1


@pytest.mark.parametrize("geom", [(2, 6, 5), (1, 1, 1), (2, 9, 70), (1, 3, 129), (1, 192, 192)])
def test_upsample_dist_exact_x4_staged_and_gathered(lib, geom, monkeypatch):
    """dml_upsample_dist_fwd at an exact x4 ratio runs the LDS-staged kernel (bands of four rows between two
    low-resolution rows, 256-column segments): single-pixel maps, segments that end inside a row, several segments, the
    benchmark's 192 -> 768 -- against F.interpolate + the distance formula on the CPU."""
    B, h, w = geom
    K = 16
    Hh, Ww = 4 * h, 4 * w
    e = rnd("ud4.e%d_%d" % (h, w), (B, K, h, w), 2.0)
    up = F.interpolate(e, size=(Hh, Ww), mode="bilinear", align_corners=False)
    feats = up.permute(0, 2, 3, 1).contiguous()
    protos = 3.0 * torch.eye(K) + 0.1 * rnd("ud4.p", (K, K))
    logits = -((feats.unsqueeze(3) - protos) ** 2).sum(-1).permute(0, 3, 1, 2)
    ed = e.permute(0, 2, 3, 1).contiguous().cuda()
    pr = protos.cuda()
    lg = torch.full((B, K, Hh, Ww), float("nan"), device="cuda")
    ft = torch.full((B, Hh, Ww, K), float("nan"), device="cuda")
    chk(lib.dml_upsample_dist_fwd(ed.data_ptr(), pr.data_ptr(), lg.data_ptr(), ft.data_ptr(), None, None, B, h, w, K, K,
                                  Hh, Ww, st()))
    torch.cuda.synchronize()
    relclose(ft.cpu(), feats, 1e-6, "staged x4 features")
    relclose(lg.cpu(), logits, 1e-5, "staged x4 logits")
    # outputs may be requested one at a time
    lg2 = torch.full_like(lg, float("nan"))
    chk(lib.dml_upsample_dist_fwd(ed.data_ptr(), pr.data_ptr(), lg2.data_ptr(), None, None, None, B, h, w, K, K, Hh, Ww, st()))
    torch.cuda.synchronize()
    assert torch.equal(lg2, lg)


TAIL_CASES = [
    # name, B, H, W, Cin, Cout, k, dil   -> tiles of 128 x 128, remainder beyond the last multiple of 256, parts
    ("3x3_288tiles", 2, 96, 96, 256, 256, 3, 1),          # 288 tiles: 32 remainder tiles x 8 parts
    ("1x1_k1024_320tiles", 2, 80, 128, 1024, 256, 1, 1),  # 320 tiles: 64 x 4
    ("3x3_d2_576", 4, 96, 96, 128, 256, 3, 2),            # 576 tiles: 64 x 4, K = 1152
    ("1x1_n512_360", 1, 96, 120, 512, 512, 1, 1),         # 90 x 4 = 360 tiles: 104 remainder x 2 parts, 16 K steps
    ("3x3_256rows_288", 4, 96, 96, 512, 256, 3, 1),       # K = 4608: 256-row tiles, 144 x 2 = 288 tiles: 32 x 8 parts
]


@pytest.mark.parametrize("case", TAIL_CASES, ids=[c[0] for c in TAIL_CASES])
def test_conv_tail_split_k(lib, case):
    """DmlConvDesc.tail_*: the tiles of a partially filled last round are computed by q workgroups each (split along K,
    fp32 partials parked in the workspace, the last arriver adds them in part order and runs the epilogue).  Forward with
    the fused BN statistics and the data gradient with accumulate, bf16, against F.conv2d on the same rounded operands;
    twice in a row (the counters reset themselves) and bit-identical to the unsplit launch's statistics path."""
    name, B, Hh, Ww, Cin, Cout, k, dil = case
    dt, tdt = 1, torch.bfloat16
    x = qz(rnd("tail.x" + name, (B, Cin, Hh, Ww)), tdt).requires_grad_(True)
    w = qz(rnd("tail.w" + name, (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5), tdt).requires_grad_(True)
    y_ref, pad = conv_ref(x, w, k, 1, dil)
    gy = qz(rnd("tail.gy" + name, tuple(y_ref.shape)), tdt)
    y_ref.backward(gy)
    xd = nhwc(x.detach(), tdt)
    wd = w.detach().permute(0, 2, 3, 1).contiguous().to("cuda", tdt)
    wtd = w.detach().permute(1, 2, 3, 0).contiguous().to("cuda", tdt)
    M = B * Hh * Ww
    ws = torch.full((512 * 128 * 128,), float("nan"), device="cuda")
    cnt = torch.zeros(128, dtype=torch.int32, device="cuda")

    def run(desc, out, use_tail):
        desc.tail_ws, desc.tail_ws_elems = (ws.data_ptr(), ws.numel()) if use_tail else (None, 0)
        desc.tail_counters, desc.tail_counters_len = (cnt.data_ptr(), cnt.numel()) if use_tail else (None, 0)
        out.fill_(float("nan"))
        chk(lib.dml_conv_igemm(C.byref(desc), st()))
        torch.cuda.synchronize()
        return out.clone()

    yd = torch.empty((B, Hh, Ww, Cout), device="cuda", dtype=tdt)
    stats = torch.zeros((M + 63) // 64 * Cout * 2, device="cuda")
    d = make_desc(lib, xd, wd, yd, B, Hh, Ww, Cin, Hh, Ww, Cout, k, 1, dil, pad, dt, stats=stats)
    y_plain = run(d, yd, False)
    st_plain = stats.clone()
    for rep in range(2):
        stats.zero_()
        y_tail = run(d, yd, True)
        assert int(cnt.abs().sum()) == 0                               # counters back to zero
        relclose(nchw(y_tail), y_ref.detach(), 1e-2, "tail fwd %s (run %d)" % (name, rep))
        # same accumulation order per K part => at most one bf16 ulp from the unsplit result, statistics alike
        relclose(y_tail.float(), y_plain.float(), 2.0 ** -7, "tail vs plain fwd " + name)
        relclose(stats, st_plain, 1e-3, "tail statistics " + name)
    import os
    # did the launch split?  256-row tiles when K >= 2304 (default rule), 128-row ones otherwise
    Kt = k * k * Cin
    rule = Kt >= 4608 or (Kt >= 2304 and (M + 255) // 256 * (Cout // 128) >= 512)
    bm = 256 if (rule and os.environ.get("DML_CONV_BM256", "1") == "1") or os.environ.get("DML_CONV_BM256") == "2" else 128
    tiles = (M + bm - 1) // bm * (Cout // 128)
    full, rem = tiles // 256 * 256, tiles % 256
    expect = full >= 256 and 0 < rem <= 128 and not os.environ.get("DML_CONV_V1") and os.environ.get("DML_CONV_TAIL", "1") != "0"
    if os.environ.get("DML_CONV_BM256", "1") in ("0", "1", "2"):
        assert bool(torch.isnan(ws[:128 * 128]).any()) != expect, (tiles, full, rem)     # the workspace was used iff it split
    # data gradient, accumulating into an initialised buffer
    gyd = nhwc(gy, tdt)
    dxd = torch.empty((B, Hh, Ww, Cin), device="cuda", dtype=tdt)
    dd = make_desc(lib, gyd, wtd, dxd, B, Hh, Ww, Cout, Hh, Ww, Cin, k, 1, dil, pad, dt, mode=1)
    g_tail = run(dd, dxd, True)
    relclose(nchw(g_tail), x.grad, 1e-2, "tail dgrad " + name)
    dd.accum = 1
    dd.tail_ws, dd.tail_ws_elems, dd.tail_counters, dd.tail_counters_len = ws.data_ptr(), ws.numel(), cnt.data_ptr(), cnt.numel()
    dxd.copy_(g_tail)
    chk(lib.dml_conv_igemm(C.byref(dd), st()))
    torch.cuda.synchronize()
    relclose(nchw(dxd), 2 * x.grad, 2e-2, "tail dgrad accum " + name)
    assert int(cnt.abs().sum()) == 0


WS_CASES = [
    # name, mode, B, H, W, C (K side), N (output side), k, stride (of the forward conv), dil, bnr, accum, res
    ("fwd_1x1_k256_n1024", 0, 4, 48, 48, 256, 1024, 1, 1, 1, 0, 0, 0),          # layer3 conv3: 8 K steps, 4 n-blocks
    ("fwd_3x3_n256", 0, 2, 48, 48, 256, 256, 3, 1, 1, 0, 0, 0),                 # layer3 conv2
    ("fwd_3x3_d2_ragged", 0, 2, 37, 29, 128, 256, 3, 1, 2, 0, 0, 0),            # rows not a multiple of 16 / 48 / 144
    ("fwd_1x1_odd_ksteps", 0, 3, 24, 40, 96, 256, 1, 1, 1, 0, 0, 0),            # 3 K steps (odd: the unrolled pair + a tail step)
    ("fwd_3x3_s2_n128", 0, 2, 96, 96, 128, 128, 3, 2, 1, 0, 0, 0),              # stride 2, the 288 x 128 tile
    ("fwd_1x1_n384", 0, 2, 48, 48, 512, 384, 1, 1, 1, 0, 0, 0),                 # N % 256 != 0: 128-wide tiles, three of them
    ("dgrad_1x1_bnr_res", 1, 4, 48, 48, 256, 1024, 1, 1, 1, 1, 0, 1),           # layer3 conv1 data gradient + bn3 sums + identity add
    ("dgrad_1x1_bnr_accum", 1, 2, 96, 96, 128, 512, 1, 1, 1, 1, 1, 0),
    ("dgrad_1x1_stride2", 1, 4, 48, 48, 512, 256, 1, 2, 1, 0, 0, 0),            # downsample conv: writes 96 x 96
    ("dgrad_3x3_d2_bnr", 1, 2, 48, 48, 512, 512, 3, 1, 2, 1, 0, 0),
    ("dgrad_3x3_s2", 1, 2, 48, 48, 256, 256, 3, 2, 1, 0, 0, 0),
]


@pytest.mark.parametrize("case", WS_CASES, ids=[c[0] for c in WS_CASES])
def test_wave_specialised_conv_equals_ring_kernel(lib, case):
    """conv_ws_kernel (one persistent workgroup per CU: loader waves feed an LDS ring, consumer waves only read fragments and
    issue MFMAs, 144-row tiles, statistics per 48 rows) against the ring kernel (conv_igemm_dma_kernel, 128-row tiles, 64-row
    statistics groups) on the same launch: same products in the same K order, so the OUTPUT must be bit-identical; the partial
    statistics / BN-backward partials are grouped differently and must agree after the merge.  The ring kernel is what the other
    tests check against torch.  DmlConvDesc.ws_min_tiles selects the kernel in-process."""
    from dmlnet._lib import PrepDesc
    name, mode, B, Hh, Ww, Cc, N, k, stride, dil, bnr, accum, res = case
    g = torch.Generator(device="cuda").manual_seed(5)
    pad = dil * (k // 2)
    if mode == 0:
        Hi, Wi = Hh, Ww
        Ho, Wo = (Hh + 2 * pad - dil * (k - 1) - 1) // stride + 1, (Ww + 2 * pad - dil * (k - 1) - 1) // stride + 1
    else:                          # data gradient: "input" is dy on the forward output grid, output on the input grid
        Hi, Wi, Ho, Wo = Hh, Ww, Hh * stride, Ww * stride
    x = (torch.randn(B, Hi, Wi, Cc, device="cuda", generator=g)).to(torch.bfloat16)
    master = (torch.randn(N, k, k, Cc, device="cuda", generator=g) * 0.05).contiguous()
    w = torch.empty(N * k * k * Cc, device="cuda", dtype=torch.bfloat16)
    arr = (PrepDesc * 1)(PrepDesc(master.data_ptr(), w.data_ptr(), None, N, k * k, Cc, Cc, 1, 0))
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    chk(lib.dml_prep_weights(tab.data_ptr(), 1, 1, st()))
    M = B * Ho * Wo
    y0 = (torch.randn(B, Ho, Wo, N, device="cuda", generator=g) * 0.3).to(torch.bfloat16)
    ybn = (torch.randn(M, N, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    bits = torch.randint(0, 256, (M * N // 8,), device="cuda", dtype=torch.uint8, generator=g)
    rdz = (torch.randn(M, N, device="cuda", generator=g)).to(torch.bfloat16)
    rbits = torch.randint(0, 256, (M * N // 8,), device="cuda", dtype=torch.uint8, generator=g)
    mean, invstd = torch.randn(N, device="cuda", generator=g) * 0.2, torch.rand(N, device="cuda", generator=g) + 0.5

    def run(ws):
        y = y0.clone()
        d = make_desc(lib, x, w, y, B, Hi, Wi, Cc, Ho, Wo, N, k, stride, dil, pad, 1, mode=mode, accum=accum)
        d.w_tiled, d.ws_min_tiles = 1, (1 if ws else 2 ** 31 - 1)
        rows = lib.dml_conv_stat_rows(C.byref(d))
        assert rows == (48 if ws else 64)
        G = (M + rows - 1) // rows
        stats = torch.full((G * N * 2,), 3.0, device="cuda")
        part = torch.full((G * N * 2,), 7.0, device="cuda")
        if mode == 0:
            d.stats = stats.data_ptr()
        if bnr:
            d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), bits.data_ptr(), mean.data_ptr(), invstd.data_ptr()
            d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), N, 1
        if res:
            d.res_dz, d.res_mask, d.res_ld = rdz.data_ptr(), rbits.data_ptr(), N
        for _ in range(2 if not accum else 1):           # a second launch over a warm cache changes the interleaving
            chk(lib.dml_conv_igemm(C.byref(d), st()))
        torch.cuda.synchronize()
        fin = None
        if mode == 0:
            sc, sh, mu, inv = (torch.empty(N, device="cuda") for _ in range(4))
            chk(lib.dml_bn_finalize(stats.data_ptr(), M, N, rows, None, None, None, None, 0.1, 1e-5, sc.data_ptr(), sh.data_ptr(),
                                    mu.data_ptr(), inv.data_ptr(), st()))
            torch.cuda.synchronize()
            fin = (mu, inv)
        return y, fin, part.view(G, N, 2).double().sum(0)

    y1, f1, p1 = run(True)
    y2, f2, p2 = run(False)
    assert torch.isfinite(y1.float()).all()
    assert (y1.float() - y0.float()).abs().max().item() > 0.1, "nothing was written"
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)), "%s: outputs differ at %d of %d elements" % (
        name, (y1.view(torch.int16) != y2.view(torch.int16)).sum().item(), y1.numel())
    if mode == 0:
        relclose(f1[0], f2[0], 1e-5, name + ": batch mean from the 48-row partials")
        relclose(f1[1], f2[1], 1e-5, name + ": 1/sigma from the 48-row partials")
    if bnr:
        relclose(p1, p2, 1e-5, name + ": BN-backward sums")
        assert (p1 != 7.0 * ((M + 47) // 48)).any()


@pytest.mark.parametrize("bnr", [0, 1])
def test_dgrad_adds_the_masked_residual_gradient_in_its_epilogue(lib, bnr):
    """DmlConvDesc.res_*: y = conv^T(dy) + res_dz (.) [mask bit] must be bit-identical to the accumulate path on a
    pre-masked copy (what the plans did before: dml_bn_bwd_apply wrote dz (.) mask, conv1's data gradient accumulated onto
    it), with and without the fused BN-backward sums; ragged row count."""
    B, Hh, Ww, Cin, Cout = 3, 21, 19, 256, 64                # data gradient of a 1x1 conv Cin -> Cout: writes [M][Cin]
    M = B * Hh * Ww
    g = torch.Generator(device="cuda").manual_seed(3)
    dy = torch.randn(B, Hh, Ww, Cout, device="cuda", generator=g).to(torch.bfloat16)
    wt = (torch.randn(Cin, 1, 1, Cout, device="cuda", generator=g) * 0.1).to(torch.bfloat16)
    dz = torch.randn(M, Cin, device="cuda", generator=g).to(torch.bfloat16)
    bits = torch.randint(0, 256, (M * Cin // 8,), device="cuda", dtype=torch.uint8, generator=g)
    mk = ((bits.view(M, Cin // 8, 1) >> torch.arange(8, device="cuda").view(1, 1, 8)) & 1).reshape(M, Cin).bool()
    masked = torch.where(mk, dz, torch.zeros_like(dz))
    ybn = (torch.randn(M, Cin, device="cuda", generator=g) * 1.5 + 0.3).to(torch.bfloat16)
    bits2 = torch.randint(0, 256, (M * Cin // 8,), device="cuda", dtype=torch.uint8, generator=g)
    mean, invstd = torch.randn(Cin, device="cuda", generator=g) * 0.2, torch.rand(Cin, device="cuda", generator=g) + 0.5
    G = (M + 63) // 64
    outs = []
    for fused in (True, False):
        y = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16) if fused else masked.clone()
        part = torch.full((G * Cin * 2,), 7.0, device="cuda")
        d = make_desc(lib, dy, wt, y, B, Hh, Ww, Cout, Hh, Ww, Cin, 1, 1, 1, 0, 1, mode=1, accum=0 if fused else 1)
        if fused:
            d.res_dz, d.res_mask, d.res_ld = dz.data_ptr(), bits.data_ptr(), Cin
        if bnr:
            d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), bits2.data_ptr(), mean.data_ptr(), invstd.data_ptr()
            d.bnr_partials, d.bnr_ldy, d.bnr_relu = part.data_ptr(), Cin, 1
        chk(lib.dml_conv_igemm(C.byref(d), st()))
        torch.cuda.synchronize()
        outs.append((y, part))
    (y1, p1), (y2, p2) = outs
    assert (y2.float() - masked.float()).abs().max().item() > 0.1
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)), "%d elements differ" % (y1.view(torch.int16) != y2.view(torch.int16)).sum().item()
    assert torch.equal(p1.view(torch.int32), p2.view(torch.int32))
    # the combination with accum is refused
    d = make_desc(lib, dy, wt, y1, B, Hh, Ww, Cout, Hh, Ww, Cin, 1, 1, 1, 0, 1, mode=1, accum=1)
    d.res_dz, d.res_mask, d.res_ld = dz.data_ptr(), bits.data_ptr(), Cin
    assert lib.dml_conv_igemm(C.byref(d), st()) != 0


@pytest.mark.parametrize("shape", [(3, 21, 19, 64, 256, 1, 1), (3, 21, 19, 256, 64, 1, 1), (1, 24, 24, 512, 128, 3, 2),
                                   (2, 17, 13, 96, 256, 3, 1)])
def test_dgrad_rounds_a_staged_gradient_once(lib, shape):
    """DmlConvDesc.acc32: a gradient with several producers (d(out) of the ASPP, a block input that also feeds a downsample
    branch) is summed in an fp32 staging tensor by the earlier producers (y_f32, accum) and the last producer writes
    bf16(conv + staging) -- one rounding of the fp32 total, as autograd does in the reference (network/utils.py:360,
    resnet.py:112-113).  Checked against torch on the bf16 operands: the result is the correctly rounded fp32 sum (up to
    the summation order of the fp32 accumulators), while rounding after every producer is measurably further away.
    Shapes: 64- and 128-wide LDS-DMA tiles, the 256-row variant (K = 4608), and a channel count the DMA kernels do not
    take (C = 96), which must be refused."""
    import torch.nn.functional as F
    B, Hh, Ww, Cout, Cin, k, dil = shape                      # data gradient of a conv Cin -> Cout: writes [M][Cin]
    M, pad = B * Hh * Ww, dil * (k - 1) // 2
    g = torch.Generator(device="cuda").manual_seed(11)
    dys = [torch.randn(B, Hh, Ww, Cout, device="cuda", generator=g).to(torch.bfloat16) for _ in range(3)]
    wts = [(torch.randn(Cin, k, k, Cout, device="cuda", generator=g) * (1.0 / (Cout * k * k)) ** 0.5).to(torch.bfloat16)
           for _ in range(3)]

    def ref(i):         # conv^T as a correlation with the flipped taps; fp64 on the bf16 operands
        w = wts[i].double().flip(1, 2).permute(0, 3, 1, 2)                   # [Cin][Cout][k][k]
        return F.conv2d(dys[i].double().permute(0, 3, 1, 2), w, None, 1, pad, dil).permute(0, 2, 3, 1).reshape(M, Cin)

    stage = torch.empty(M, Cin, device="cuda", dtype=torch.float32)
    y = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16)
    d = make_desc(lib, dys[2], wts[2], y, B, Hh, Ww, Cout, Hh, Ww, Cin, k, 1, dil, pad, 1, mode=1)
    d.acc32, d.acc32_ld = stage.data_ptr(), Cin
    if Cout % 32:
        assert lib.dml_conv_igemm(C.byref(d), st()) != 0      # no LDS-DMA kernel for this shape: refused, not mis-computed
        return
    for i in range(2):
        di = make_desc(lib, dys[i], wts[i], stage, B, Hh, Ww, Cout, Hh, Ww, Cin, k, 1, dil, pad, 1, mode=1, y_f32=1, accum=i)
        chk(lib.dml_conv_igemm(C.byref(di), st()))
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    # the old way: bf16 buffer, rounded after every producer
    y3 = torch.empty(M, Cin, device="cuda", dtype=torch.bfloat16)
    for i in range(3):
        di = make_desc(lib, dys[i], wts[i], y3, B, Hh, Ww, Cout, Hh, Ww, Cin, k, 1, dil, pad, 1, mode=1, accum=1 if i else 0)
        chk(lib.dml_conv_igemm(C.byref(di), st()))
    torch.cuda.synchronize()
    total = ref(0) + ref(1) + ref(2)
    want = total.float().to(torch.bfloat16)
    same = (y.view(torch.int16) == want.view(torch.int16)).float().mean().item()
    e1 = (y.double() - total).abs().max().item() / total.abs().max().item()
    e3 = (y3.double() - total).abs().max().item() / total.abs().max().item()
    r1 = ((y.double() - total) ** 2).mean().sqrt().item()
    r3 = ((y3.double() - total) ** 2).mean().sqrt().item()
    print("staged: %.4f of the elements are the correctly rounded sum, max err %.2e (three roundings %.2e), rms %.3e vs %.3e"
          % (same, e1, e3, r1, r3))
    assert same > 0.995 and e1 <= 2.0 ** -8
    assert r1 < 0.8 * r3
    # refused combinations
    d.accum = 1
    assert lib.dml_conv_igemm(C.byref(d), st()) != 0
    d.accum, d.y_f32 = 0, 1
    assert lib.dml_conv_igemm(C.byref(d), st()) != 0


@pytest.mark.parametrize("case", [c for c in CONV_CASES if c[4] % 32 == 0], ids=[c[0] for c in CONV_CASES if c[4] % 32 == 0])
def test_conv_f32_three_term_split_is_fp32_accurate(lib, case):
    """DmlConvDesc.f32_split: fp32 tensors, products on the bf16 matrix cores through hi + mid + lo of both operands (six bf16
    MFMAs per block).  Forward (with BN statistics) and data gradient against an fp64 torch convolution: the error must be at
    the level of the EXACT fp32 kernel's (same launch with f32_split = 0), far below the 1e-3 parity bar -- while a plain bf16
    rounding of the operands would sit at 4e-3."""
    name, B, Hh, Ww, Cin, Cout, k, stride, dil = case
    x = rnd(name + ".x", (B, Cin, Hh, Ww)).double().requires_grad_(True)
    w = rnd(name + ".w", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5).double().requires_grad_(True)
    y_ref, pad = conv_ref(x, w, k, stride, dil)
    Ho, Wo = y_ref.shape[2:]
    gy = rnd(name + ".gy", tuple(y_ref.shape)).double()
    y_ref.backward(gy)
    xd = nhwc(x.detach().float(), torch.float32)
    wd = w.detach().float().permute(0, 2, 3, 1).contiguous().cuda()
    wtd = w.detach().float().permute(1, 2, 3, 0).contiguous().cuda()
    gyd = nhwc(gy.float(), torch.float32)
    M = B * Ho * Wo
    errs = {}
    for split in (0, 1):
        yd = torch.empty((B, Ho, Wo, Cout), device="cuda")
        stats = torch.zeros((M + 63) // 64 * Cout * 2, device="cuda")
        d = make_desc(lib, xd, wd, yd, B, Hh, Ww, Cin, Ho, Wo, Cout, k, stride, dil, pad, 0, stats=stats)
        d.f32_split = split
        chk(lib.dml_conv_igemm(C.byref(d), st()))
        dxd = torch.empty((B, Hh, Ww, Cin), device="cuda")
        dd = make_desc(lib, gyd, wtd, dxd, B, Ho, Wo, Cout, Hh, Ww, Cin, k, stride, dil, pad, 0, mode=1)
        dd.f32_split = split
        chk(lib.dml_conv_igemm(C.byref(dd), st()))
        torch.cuda.synchronize()
        ef = (nchw(yd).double() - y_ref.detach()).abs().max().item() / y_ref.detach().abs().max().item()
        eg = (nchw(dxd).double() - x.grad).abs().max().item() / x.grad.abs().max().item()
        # statistics partials: sum over each 64-row group of the fp32 accumulators
        yflat = y_ref.detach().permute(0, 2, 3, 1).reshape(M, Cout)
        g0 = yflat[:min(64, M)].sum(0)
        es = (stats.view(-1, Cout, 2)[0, :, 0].double().cpu() - g0).abs().max().item() / (g0.abs().max().item() + 1e-30)
        # weight gradient: atomics path and workspace (plain stores + fold) path
        from dmlnet._lib import WgradDesc
        ew = []
        ws = torch.empty(6 * Cout * k * k * Cin, device="cuda")
        for use_ws, sk in ((0, 0), (0, 3), (1, 4)):
            dw = torch.zeros((Cout, k, k, Cin), device="cuda")
            wg = WgradDesc(x=xd.data_ptr(), dy=gyd.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hh, Wi=Ww, C=Cin, ldx=Cin, Ho=Ho, Wo=Wo,
                           N=Cout, ldy=Cout, R=k, S=k, stride=stride, dil=dil, pad=pad, dtype=0, splitk=sk, f32_split=split,
                           ws=ws.data_ptr() if use_ws else None, ws_elems=ws.numel() if use_ws else 0)
            chk(lib.dml_conv_wgrad(C.byref(wg), st()))
            torch.cuda.synchronize()
            ew.append((dw.cpu().permute(0, 3, 1, 2).double() - w.grad).abs().max().item() / w.grad.abs().max().item())
        errs[split] = (ef, eg, es, max(ew))
    print("%s: fwd / dgrad / stats / wgrad error vs fp64: exact fp32 MFMA %.2e %.2e %.2e %.2e | three-term split %.2e %.2e %.2e %.2e"
          % ((name,) + errs[0] + errs[1]))
    for a_, b_ in zip(errs[1], errs[0]):
        assert a_ <= max(4.0 * b_, 2e-6), errs


def test_split_products_keep_a_non_finite_input_visible(lib):
    """f32_split = 1: an Inf in the activations becomes hi = Inf, mid = Inf - Inf = NaN (include/dmlnet_hip.h): every output the Inf
    reaches is non-finite (NaN where exact fp32 gives Inf), every output it does not reach is unaffected."""
    B, Hh, Ww, Cin, Cout = 1, 8, 8, 64, 64
    x = torch.randn(B, Hh, Ww, Cin, device="cuda")
    x[0, 3, 4, 7] = float("inf")
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.1).contiguous()
    ys = {}
    for split in (0, 1):
        y = torch.zeros(B, Hh, Ww, Cout, device="cuda")
        d = make_desc(lib, x, w, y, B, Hh, Ww, Cin, Hh, Ww, Cout, 1, 1, 1, 0, 0)
        d.f32_split = split
        chk(lib.dml_conv_igemm(C.byref(d), st()))
        torch.cuda.synchronize()
        ys[split] = y
    hit = torch.zeros(B, Hh, Ww, dtype=torch.bool, device="cuda")
    hit[0, 3, 4] = True
    for split in (0, 1):
        assert not torch.isfinite(ys[split][hit]).any(), split
        assert torch.isfinite(ys[split][~hit]).all(), split
    assert (ys[1][~hit] - ys[0][~hit]).abs().max().item() <= 1e-5 * ys[0][~hit].abs().max().item()


def test_two_plane_forward_with_accumulate_falls_back_and_accumulates(lib):
    """mode 0 + accum + planes: the two-plane forward epilogue has no accumulate path, so the launch is NOT eligible for that kernel
    (dml_conv_stat_rows says 64) and runs the three-term split on x / w -- the sum is still what comes out."""
    B, Hh, Ww, Cin, Cout = 2, 12, 12, 64, 128
    x = torch.randn(B, Hh, Ww, Cin, device="cuda")
    w = (torch.randn(Cout, 1, 1, Cin, device="cuda") * 0.1).contiguous()
    y0 = torch.randn(B, Hh, Ww, Cout, device="cuda")
    y = y0.clone()
    d = make_desc(lib, x, w, y, B, Hh, Ww, Cin, Hh, Ww, Cout, 1, 1, 1, 0, 0, accum=1)
    xp, xw = h2_planes(lib, x.view(-1, Cin), 0)
    wp, ww = h2_planes(lib, w.view(Cout, -1), 1)
    d.f32_split = 2
    d.x_planes, d.x_unscale, d.x_plane_stride = xp.data_ptr(), xw.data_ptr() + 4096, xp.shape[1]
    d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
    assert lib.dml_conv_stat_rows(C.byref(d)) == 64
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    torch.cuda.synchronize()
    ref = y0.double() + torch.einsum("bhwc,nc->bhwn", x.double(), w.view(Cout, Cin).double())
    assert (y.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    d.accum = 0
    assert lib.dml_conv_stat_rows(C.byref(d)) == 144             # (forward on 144-row wave tiles: one statistics partial per wave tile)


def h2_planes(lib, t2d, layout=0, amax=None):
    """dml_h2_split of an fp32 [rows][C] cuda tensor -> (planes fp16 [2][rows * C], work) ; work[1024] = 1 / scale"""
    rows, Cc = t2d.shape
    planes = torch.empty((2, rows * Cc), device="cuda", dtype=torch.float16)
    work = torch.zeros(1025, device="cuda")
    if amax is not None:
        work[77] = amax                 # anywhere in the first 1024 words
    chk(lib.dml_h2_split(t2d.data_ptr(), rows, Cc, Cc, planes.data_ptr(), rows * Cc, Cc, layout, work.data_ptr(),
                         0 if amax is None else 1, st()))
    return planes, work


@pytest.mark.parametrize("scale", [1.0, 3e-5, 7e4, 0.0])
def test_h2_split_reconstructs_the_tensor(lib, scale):
    """dml_h2_split: (hi + lo) / s reproduces x to 2^-21 of its magnitude (elements far below the maximum: to 2^-36 of the
    maximum), s is the power of two that puts max|x| into [2^14, 2^15), both layouts hold the same values, zero tensors work."""
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.randn(192, 96, device="cuda", generator=g) * scale).contiguous()
    x[5, 7] *= 1e-6                                      # a tiny element next to large ones
    pl, work = h2_planes(lib, x, 0)
    torch.cuda.synchronize()
    un = work[1024].item()
    amax = x.abs().max().item()
    if amax > 0:
        assert 2.0 ** 14 <= amax / un < 2.0 ** 15 and np.log2(un) == np.floor(np.log2(un))
    else:
        assert un == 1.0
    rec = (pl[0].double() + pl[1].double()).view(192, 96) * un
    err = (rec - x.double()).abs()
    assert (err <= x.double().abs() * 2.0 ** -21 + amax * 2.0 ** -36).all(), err.max().item()
    assert torch.equal(pl[0].view(192, 96).float(), (x / un).half().float())        # hi = fp16(s x), round to nearest even
    plt, workt = h2_planes(lib, x, 1)
    torch.cuda.synchronize()
    for p_ in (0, 1):
        assert torch.equal(plt[p_].view(-1), _tile_major(pl[p_].view(192, 96)).view(-1))
    assert workt[1024].item() == un
    plk, workk = h2_planes(lib, x, 0, amax=amax)            # the producer already knows max |x|: one pass, same planes
    torch.cuda.synchronize()
    assert torch.equal(plk, pl) and workk[1024].item() == un
    # an Inf poisons the scale instead of being clipped silently
    x[1, 1] = float("inf")
    _, w2 = h2_planes(lib, x, 0)
    torch.cuda.synchronize()
    v = w2[1024].item()
    assert v == 0.0 or not np.isfinite(v)


def test_h2_split_table_equals_the_single_tensor_calls(lib):
    """dml_h2_split_table (all weight copies of a plan in two launches) writes the planes and scales of per-tensor dml_h2_split
    calls bit for bit: both layouts, tensors from 64 x 32 to 512 x 2304, one all-zero."""
    from dmlnet._lib import H2Desc
    g = torch.Generator(device="cuda").manual_seed(11)
    shapes = [(64, 32, 1, 1.0), (256, 2304, 1, 0.05), (192, 96, 0, 3e3), (512, 256, 1, 1e-4), (64, 64, 0, 0.0)]
    xs = [(torch.randn(r, c, device="cuda", generator=g) * sc).contiguous() for r, c, _, sc in shapes]
    ref = [h2_planes(lib, x, lay) for x, (_, _, lay, _) in zip(xs, shapes)]
    pls = [torch.zeros_like(p) for p, _ in ref]
    works = [torch.zeros(1025, device="cuda") for _ in ref]
    arr = (H2Desc * len(xs))(*[H2Desc(x.data_ptr(), p.data_ptr(), w.data_ptr(), x.shape[0], x.numel(), x.shape[1], x.shape[1],
                                     x.shape[1], lay) for x, p, w, (_, _, lay, _) in zip(xs, pls, works, shapes)])
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
    chk(lib.dml_h2_split_table(tab.data_ptr(), len(xs), st()))
    torch.cuda.synchronize()
    for (p0, w0), p1, w1 in zip(ref, pls, works):
        assert torch.equal(p0, p1) and w0[1024].item() == w1[1024].item()


H2_CASES = [("h2_1x1", 2, 24, 20, 64, 128, 1, 1, 1), ("h2_3x3", 2, 19, 23, 64, 256, 3, 1, 1), ("h2_3x3_d2", 2, 16, 16, 128, 256, 3, 1, 2),
            ("h2_3x3_s2", 2, 22, 18, 128, 128, 3, 2, 1), ("h2_rows", 3, 40, 40, 256, 256, 3, 1, 1), ("h2_1x1_n384", 2, 13, 29, 96, 384, 1, 1, 1),
            ("h2_3x3_n64", 2, 21, 17, 64, 64, 3, 1, 1), ("h2_1x1_tile_stats", 5, 64, 61, 64, 256, 1, 1, 1),
            # K order of the two-plane kernel (64-channel groups outermost, taps inside): a last group of 32 channels (C = 96), five
            # groups (C = 320); three column blocks, the last one half empty, walked fastest (N = 320: forward of the first case and
            # data gradient of the second, the shape of the decoder's 3x3)
            ("h2_3x3_c96_n320", 2, 17, 21, 96, 320, 3, 1, 1), ("h2_3x3_c320_n192", 2, 15, 14, 320, 192, 3, 1, 1),
            # dilated 3x3 on a 48 x 48 map (the ASPP branches): whole filter rows are padding for the tiles at the top and the bottom of an
            # image and their K steps are skipped (ws_live_taps) -- 144-row tiles forward and data gradient, 192-row tiles (64 channels)
            ("h2_3x3_d12_48", 2, 48, 48, 256, 256, 3, 1, 12), ("h2_3x3_d18_48", 1, 48, 48, 64, 256, 3, 1, 18),
            ("h2_3x3_d6_40x48", 2, 40, 48, 128, 128, 3, 1, 6)]


@pytest.mark.parametrize("case", H2_CASES, ids=lambda c: c[0])
def test_conv_f16_two_plane_split_is_fp32_accurate(lib, case):
    """DmlConvDesc.f32_split = 2 with x_planes / w_planes (dml_h2_split): forward (with BN statistics) and data gradient of fp32
    tensors with the products of two fp16 planes per operand on the matrix cores (conv_ws_kernel<.., 2>).  Against fp64 the
    error must stay within 4x the exact fp32 MFMA kernel's own (the two-term split carries 22 bits, the accumulation is fp32
    either way); operands of very different magnitude (activations ~1e3, weights ~1e-3) check the two scales."""
    name, B, Hh, Ww, Cin, Cout, k, stride, dil = case
    x = (rnd(name + ".x", (B, Cin, Hh, Ww)).double() * 1e3).requires_grad_(True)
    w = (rnd(name + ".w", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5).double() * 1e-3).requires_grad_(True)
    y_ref, pad = conv_ref(x, w, k, stride, dil)
    Ho, Wo = y_ref.shape[2:]
    gy = rnd(name + ".gy", tuple(y_ref.shape)).double() * 1e-4
    y_ref.backward(gy)
    xd = nhwc(x.detach().float(), torch.float32)
    wd = w.detach().float().permute(0, 2, 3, 1).contiguous().cuda()            # N R S C
    wtd = w.detach().float().permute(1, 2, 3, 0).contiguous().cuda()           # C R S N
    gyd = nhwc(gy.float(), torch.float32)
    M = B * Ho * Wo
    errs = {}
    for split in (0, 2):
        yd = torch.empty((B, Ho, Wo, Cout), device="cuda")
        d = make_desc(lib, xd, wd, yd, B, Hh, Ww, Cin, Ho, Wo, Cout, k, stride, dil, pad, 0)
        dxd = torch.empty((B, Hh, Ww, Cin), device="cuda")
        dd = make_desc(lib, gyd, wtd, dxd, B, Ho, Wo, Cout, Hh, Ww, Cin, k, stride, dil, pad, 0, mode=1)
        keep = []
        if split:
            for desc, act, wmat, rows_w in ((d, xd, wd, Cout), (dd, gyd, wtd, Cin)):
                if rows_w % 64:                      # not a shape of the planes kernel: the three-term split takes it
                    desc.f32_split = 1
                    continue
                ap, aw = h2_planes(lib, act.view(-1, act.shape[-1]), 0)
                wp, ww = h2_planes(lib, wmat.view(rows_w, -1), 1)
                keep += [ap, aw, wp, ww]
                desc.f32_split = 2
                desc.x_planes, desc.x_unscale, desc.x_plane_stride = ap.data_ptr(), aw.data_ptr() + 4096, ap.shape[1]
                desc.w_planes, desc.w_unscale, desc.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
        rows = lib.dml_conv_stat_rows(C.byref(d))
        # two-plane forward: one partial per 144-row wave tile -- 48 rows on the 48-row wave tiles (64 output channels; 48 x 256 tiles of
        # launches with at most 128 tiles of 144 x 256)
        short = Cout % 256 == 0 and ((M + 143) // 144) * (Cout // 256) * 2 <= 256
        assert rows == ((48 if (Cout == 64 or short) else 144) if (split and Cout % 64 == 0) else 64)
        stats = torch.zeros((M + rows - 1) // rows * Cout * 2, device="cuda")
        d.stats = stats.data_ptr()
        chk(lib.dml_conv_igemm(C.byref(d), st()))
        chk(lib.dml_conv_igemm(C.byref(dd), st()))
        sc, sh, mu, inv = (torch.empty(Cout, device="cuda") for _ in range(4))
        chk(lib.dml_bn_finalize(stats.data_ptr(), M, Cout, rows, None, None, None, None, 0.1, 1e-5, sc.data_ptr(), sh.data_ptr(),
                                mu.data_ptr(), inv.data_ptr(), st()))
        torch.cuda.synchronize()
        yr = y_ref.detach()
        ef = (nchw(yd).double() - yr).abs().max().item() / yr.abs().max().item()
        eg = (nchw(dxd).double() - x.grad).abs().max().item() / x.grad.abs().max().item()
        em = (mu.cpu().double() - yr.mean(dim=(0, 2, 3))).abs().max().item() / yr.abs().max().item()
        # weight gradient (workspace path): both operands as planes
        from dmlnet._lib import WgradDesc
        ws = torch.empty(8 * Cout * k * k * Cin, device="cuda")
        dw = torch.zeros((Cout, k, k, Cin), device="cuda")
        wg = WgradDesc(x=xd.data_ptr(), dy=gyd.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hh, Wi=Ww, C=Cin, ldx=Cin, Ho=Ho, Wo=Wo,
                       N=Cout, ldy=Cout, R=k, S=k, stride=stride, dil=dil, pad=pad, dtype=0, splitk=5, f32_split=split,
                       ws=ws.data_ptr(), ws_elems=ws.numel())
        if split:
            xp, xw = h2_planes(lib, xd.view(-1, Cin), 0)
            yp, yw = h2_planes(lib, gyd.view(-1, Cout), 0)
            keep += [xp, xw, yp, yw]
            wg.x_planes, wg.x_unscale, wg.x_plane_stride = xp.data_ptr(), xw.data_ptr() + 4096, xp.shape[1]
            wg.dy_planes, wg.dy_unscale, wg.dy_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
        chk(lib.dml_conv_wgrad(C.byref(wg), st()))
        torch.cuda.synchronize()
        ew = (dw.cpu().permute(0, 3, 1, 2).double() - w.grad).abs().max().item() / w.grad.abs().max().item()
        errs[split] = (ef, eg, em, ew)
    print("%s: fwd / dgrad / batch-mean / wgrad error vs fp64: exact fp32 MFMA %.2e %.2e %.2e %.2e | two fp16 planes %.2e %.2e %.2e %.2e"
          % ((name,) + errs[0] + errs[2]))
    for a_, b_ in zip(errs[2], errs[0]):
        assert a_ <= max(4.0 * b_, 2e-6), errs


@pytest.mark.parametrize("case", [("s2cls_3x3", 2, 24, 20, 128, 128, 3, 1), ("s2cls_3x3_ragged", 3, 14, 10, 64, 192, 3, 1),
                                  ("s2cls_1x1", 2, 24, 20, 128, 256, 1, 0)], ids=lambda c: c[0])
def test_stride2_data_gradient_by_parity_class_equals_the_direct_launch(lib, case):
    """DmlConvDesc.sub_grid (round 6): the data gradient of a stride-2 convolution as one stride-1 launch per pixel-parity class on the
    taps that class sees (3x3: 1 / 2 / 2 / 4 taps; 1x1: the even pixels only, accumulating).  Every dX element is the same sum over
    the same K steps in the same order as in the direct stride-2 launch (whose other taps contribute exact zeros): BIT-EQUAL, with
    the identity / accumulate operand and with the fused BatchNorm-backward sums (their totals over all partial groups agree to
    fp32 summation noise: the grouping of the rows differs)."""
    from dmlnet._lib import ConvDesc
    name, B, Hh, Ww, Cin, Cout, k, pad = case             # forward conv: x [B, Hh, Ww, Cin] -> y [B, Hh / 2, Ww / 2, Cout]
    Ho, Wo = Hh // 2, Ww // 2
    gy = nhwc(rnd(name + ".gy", (B, Cout, Ho, Wo)) * 1e-3, torch.float32)
    w = rnd(name + ".w", (Cout, Cin, k, k), scale=(2.0 / (Cin * k * k)) ** 0.5)
    wt = w.permute(1, 2, 3, 0).contiguous().cuda()           # C R S N: the data gradient's operand
    old = nhwc(rnd(name + ".old", (B, Cin, Hh, Ww)) * 1e-3, torch.float32)       # earlier producers' sum (accumulate)
    ybn = nhwc(rnd(name + ".ybn", (B, Cin, Hh, Ww)), torch.float32)
    mask = torch.randint(0, 16, (B * Hh * Ww * (Cin // 4),), device="cuda", dtype=torch.uint8)
    mean, inv = rnd(name + ".mu", (Cin,)).cuda(), (rnd(name + ".is", (Cin,)).abs() + 0.5).cuda()
    M = B * Hh * Ww
    yp, yw = h2_planes(lib, gy.view(-1, Cout), 0)
    keep = []

    def run(classes, accum, bnr):
        dx = old.clone() if accum else torch.full_like(old, float("nan"))
        G = (M + 47) // 48 + 8
        part = torch.zeros(G * Cin * 2, device="cuda")
        gmx = torch.zeros(1025, device="cuda")
        descs = []
        if not classes:
            wp, ww = h2_planes(lib, wt.view(Cin, -1), 1)
            d = make_desc(lib, gy, wt, dx, B, Ho, Wo, Cout, Hh, Ww, Cin, k, 2, 1, pad, 0, mode=1, accum=accum)
            d.f32_split = 2
            d.x_planes, d.x_unscale, d.x_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
            d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
            keep.extend([wp, ww])
            descs.append((d, (M + 47) // 48))
        else:
            for cy in (0, 1):
                for cx in (0, 1):
                    rs = [r for r in range(k) if (r - cy - pad) % 2 == 0]
                    ss = [t for t in range(k) if (t - cx - pad) % 2 == 0]
                    if not rs or not ss:
                        continue
                    taps = [r * k + t for r in rs for t in ss]
                    sub = torch.empty(Cin * len(taps) * Cout, device="cuda")
                    chk(lib.dml_gather_taps(wt.data_ptr(), sub.data_ptr(), Cin, k * k, Cout, len(taps), *(taps + [0] * (4 - len(taps))), st()))
                    wp, ww = h2_planes(lib, sub.view(Cin, -1), 1)
                    d = ConvDesc(x=gy.data_ptr(), w=sub.data_ptr(), y=dx.data_ptr(), bias=None, stats=None, B=B, Hi=Ho, Wi=Wo, C=Cout,
                                 ldx=Cout, Ho=Ho, Wo=Wo, N=Cin, ldy=Cin, R=len(rs), S=len(ss), stride=1, dil=1,
                                 pad=(cy + pad - rs[0]) // 2, dtype=0, y_f32=0, accum=accum, mode=1)
                    d.pad_w_set, d.pad_w = 1, (cx + pad - ss[0]) // 2
                    d.sub_grid, d.sub_y, d.sub_x = 1, cy, cx
                    d.f32_split = 2
                    d.x_planes, d.x_unscale, d.x_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
                    d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
                    keep.extend([sub, wp, ww])
                    descs.append((d, (B * Ho * Wo + 47) // 48))
        g0 = 0
        for d, gc in descs:
            d.ws_min_tiles = 1
            if bnr:
                d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), mask.data_ptr(), mean.data_ptr(), inv.data_ptr()
                d.bnr_partials, d.bnr_ldy, d.bnr_relu, d.bnr_gmax = part.data_ptr() + g0 * Cin * 8, Cin, 1, gmx.data_ptr()
            g0 += gc
            assert lib.dml_conv_stat_rows(C.byref(d)) == 48
            chk(lib.dml_conv_igemm(C.byref(d), st()))
        torch.cuda.synchronize()
        return dx, part.view(-1, Cin, 2)[:g0].double().sum(0), float(gmx[:1024].max()), len(descs)

    for accum, bnr in ((1, False), (1, True)) + (((0, False), (0, True)) if k == 3 else ()):
        a, pa, ga, na = run(False, accum, bnr)
        b, pb, gb, nb = run(True, accum, bnr)
        assert na == 1 and nb == (4 if k == 3 else 1)
        assert torch.equal(a, b), "%s accum=%d bnr=%d: %d of %d elements differ (max %.3e)" % (
            name, accum, bnr, int((a != b).sum()), a.numel(), (a - b).abs().max().item())
        if bnr and k == 3:                 # (1x1: the class launch visits a quarter of the rows -- the plan does not fuse the sums there)
            assert ga == gb
            assert (pa - pb).abs().max().item() <= 1e-5 * pa.abs().max().item()
    if k == 1:
        # DmlConvDesc.bnr_inc: the class launch's sums over its INCREMENT + the sums of the earlier producers' share (here: `old`, in
        # fp64 on the host) = the sums of the stored total, which the direct launch (all pixels, accumulate + fused sums) computes
        a, pa, ga, _ = run(False, 1, True)
        rs = [0]
        sub = torch.empty(Cin * Cout, device="cuda")
        chk(lib.dml_gather_taps(wt.data_ptr(), sub.data_ptr(), Cin, 1, Cout, 1, 0, 0, 0, 0, st()))
        wp, ww = h2_planes(lib, sub.view(Cin, -1), 1)
        dx = old.clone()
        part = torch.zeros(((B * Ho * Wo + 47) // 48) * Cin * 2, device="cuda")
        gmx = torch.zeros(1025, device="cuda")
        d = ConvDesc(x=gy.data_ptr(), w=sub.data_ptr(), y=dx.data_ptr(), bias=None, stats=None, B=B, Hi=Ho, Wi=Wo, C=Cout, ldx=Cout,
                     Ho=Ho, Wo=Wo, N=Cin, ldy=Cin, R=1, S=1, stride=1, dil=1, pad=0, dtype=0, y_f32=0, accum=1, mode=1)
        d.pad_w_set, d.pad_w, d.sub_grid, d.sub_y, d.sub_x, d.f32_split, d.ws_min_tiles, d.bnr_inc = 1, 0, 1, 0, 0, 2, 1, 1
        d.x_planes, d.x_unscale, d.x_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
        d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
        d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), mask.data_ptr(), mean.data_ptr(), inv.data_ptr()
        d.bnr_partials, d.bnr_ldy, d.bnr_relu, d.bnr_gmax = part.data_ptr(), Cin, 1, gmx.data_ptr()
        chk(lib.dml_conv_igemm(C.byref(d), st()))
        torch.cuda.synchronize()
        assert torch.equal(dx, a)
        bits = (mask.view(M, Cin // 4, 1) >> torch.arange(4, device="cuda", dtype=torch.uint8).view(1, 1, 4)) & 1
        g_old = old.view(M, Cin).double() * bits.reshape(M, Cin).double()
        xhat = (ybn.view(M, Cin).double() - mean.double()) * inv.double()
        p_old = torch.stack([g_old.sum(0), (g_old * xhat).sum(0)], 1)
        p_inc = part.view(-1, Cin, 2).double().sum(0)
        err = ((p_old + p_inc) - pa).abs().max().item() / pa.abs().max().item()
        print("%s: sums of the increment + sums of the old share vs sums of the total: rel %.2e" % (name, err))
        assert err <= 1e-5
        # max |g| is still taken from the stored total of the rows the launch visits
        tot_even = (a.view(B, Hh, Ww, Cin)[:, ::2, ::2].reshape(-1, Cin) *
                    bits.reshape(B, Hh, Ww, Cin)[:, ::2, ::2].reshape(-1, Cin).float()).abs().max().item()
        assert abs(float(gmx[:1024].max()) - tot_even) <= 1e-6 * tot_even
    # shapes / modes the mapping does not exist for are refused, never mis-executed
    sub = torch.empty(Cin * Cout, device="cuda")
    d = ConvDesc(x=gy.data_ptr(), w=sub.data_ptr(), y=old.data_ptr(), bias=None, stats=None, B=B, Hi=Ho, Wi=Wo, C=Cout, ldx=Cout, Ho=Ho,
                 Wo=Wo, N=Cin, ldy=Cin, R=1, S=1, stride=1, dil=1, pad=0, dtype=0, y_f32=0, accum=1, mode=1)
    d.sub_grid = 1
    assert lib.dml_conv_igemm(C.byref(d), st()) == -3                # no planes: DML_EUNSUPPORTED
    d.mode = 0
    assert lib.dml_conv_igemm(C.byref(d), st()) == -1                # forward launches have no sub-grid: DML_EINVAL


def test_planes_only_operand_is_refused_where_the_planes_kernels_cannot_take_it(lib):
    """x == x_planes (dy == dy_planes) declares an operand that exists as fp16 planes ONLY (the plan drops the fp32 copy of tensors only
    convolutions read).  A launch the planes kernels cannot take must then fail with DML_EUNSUPPORTED -- the fallback kernels would read
    the planes as floats -- while the same launch with a real fp32 tensor behind `x` still falls back silently."""
    from dmlnet._lib import WgradDesc
    B, Hh, Ww, Cin, Cout, k = 2, 9, 7, 64, 24, 1                 # 24 output channels: no shape of the planes kernel
    xd = torch.randn(B, Hh, Ww, Cin, device="cuda")
    wd = torch.randn(Cout, k, k, Cin, device="cuda") * 0.1
    yd = torch.empty(B, Hh, Ww, Cout, device="cuda")
    ap, aw = h2_planes(lib, xd.view(-1, Cin), 0)
    wp, ww = h2_planes(lib, torch.randn(64, Cin, device="cuda"), 1)          # (any valid weight planes: never read)
    d = make_desc(lib, xd, wd, yd, B, Hh, Ww, Cin, Hh, Ww, Cout, k, 1, 1, 0, 0)
    d.f32_split = 2
    d.x_planes, d.x_unscale, d.x_plane_stride = ap.data_ptr(), aw.data_ptr() + 4096, ap.shape[1]
    d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
    chk(lib.dml_conv_igemm(C.byref(d), st()))                  # fp32 tensor behind x: the three-term kernel takes it
    torch.cuda.synchronize()
    ref = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2).double(), wd.permute(0, 3, 1, 2).double()).permute(0, 2, 3, 1)
    assert (yd.double() - ref).abs().max().item() <= 1e-5 * ref.abs().max().item()
    d.x = ap.data_ptr()                                        # planes only
    assert lib.dml_conv_igemm(C.byref(d), st()) == -3          # DML_EUNSUPPORTED
    gyd = torch.randn(B, Hh, Ww, 20, device="cuda")            # 20 channels: no fp16 vectors of 8 -> the planes weight gradient declines
    yp, yw = h2_planes(lib, torch.randn(B * Hh * Ww, 24, device="cuda"), 0)
    ws = torch.empty(8 * 20 * Cin, device="cuda")
    dw = torch.zeros(20, Cin, device="cuda")
    wg = WgradDesc(x=ap.data_ptr(), dy=gyd.data_ptr(), dw=dw.data_ptr(), B=B, Hi=Hh, Wi=Ww, C=Cin, ldx=Cin, Ho=Hh, Wo=Ww, N=20, ldy=20,
                   R=1, S=1, stride=1, dil=1, pad=0, dtype=0, splitk=2, f32_split=2, ws=ws.data_ptr(), ws_elems=ws.numel())
    wg.x_planes, wg.x_unscale, wg.x_plane_stride = ap.data_ptr(), aw.data_ptr() + 4096, ap.shape[1]
    wg.dy_planes, wg.dy_unscale, wg.dy_plane_stride = yp.data_ptr(), yw.data_ptr() + 4096, yp.shape[1]
    assert lib.dml_conv_wgrad(C.byref(wg), st()) == -3
    torch.cuda.synchronize()


@pytest.mark.parametrize("accum,relu,Cin,bnr", [(0, 1, 128, 1), (1, 1, 128, 1), (1, 0, 128, 1), (2, 1, 128, 1),
                                                 # the other tile configurations of conv_epilogue_rows_ops: 144 x 256 (Cin = 256: three
                                                 # 48-row sub-tiles per wave tile, the operand ring crosses them), 192 x 64 (Cin = 64: one
                                                 # sub-tile), a half-empty last 128-wide block (Cin = 320); operands without the BN sums
                                                 (2, 1, 256, 1), (1, 0, 256, 1), (0, 1, 256, 1), (2, 1, 64, 1), (0, 0, 64, 1), (2, 1, 320, 1),
                                                 (1, 1, 256, 0), (2, 1, 256, 0), (2, 1, 64, 0), (1, 1, 320, 0)])
def test_two_plane_dgrad_emits_bn_backward_partials(lib, accum, relu, Cin, bnr):
    """DmlConvDesc.bnr_* on fp32 tensors (the two-plane kernel's row epilogue): the data gradient writes the BN-backward sums of
    the tensor it stores per 48 rows -- what dml_bn_bwd_reduce computes from that tensor -- and raises max |g| in bnr_gmax; a
    ragged last group, the accumulate path, ReLU mask of one byte per four channels.  Other fp32 kernels refuse the request."""
    B, Hh, Ww, Cout, k = 2, 13, 11, 64, 3          # dgrad output: M = 286 rows (5.96 groups of 48; 1.99 wave tiles of 144) x Cin channels
    M = B * Hh * Ww
    gyd = (torch.randn(B, Hh, Ww, Cout, device="cuda") * 1e-2).contiguous()
    wt = (torch.randn(Cin, k, k, Cout, device="cuda") * 0.05).contiguous()          # wt[Cin][R][S][Cout]
    dx0 = torch.randn(B, Hh, Ww, Cin, device="cuda") * 3e-3
    dx = dx0.clone()
    ybn = torch.randn(M, Cin, device="cuda") * 1.5 + 0.3
    bits = torch.randint(0, 16, (M * Cin // 4,), device="cuda", dtype=torch.uint8)
    mean, invstd = torch.randn(Cin, device="cuda") * 0.2, torch.rand(Cin, device="cuda") + 0.5
    G = (M + 47) // 48
    part = torch.full((G * Cin * 2,), 7.0, device="cuda")
    gmax = torch.zeros(1024, device="cuda")
    d = make_desc(lib, gyd, wt, dx, B, Hh, Ww, Cout, Hh, Ww, Cin, k, 1, 1, 1, 0, mode=1, accum=1 if accum == 1 else 0)
    rbits = torch.randint(0, 16, (M * Cin // 4,), device="cuda", dtype=torch.uint8)
    if accum == 2:      # DmlConvDesc.res_*: the identity branch's gradient (dx0 under its own ReLU mask) added in the epilogue
        d.res_dz, d.res_mask, d.res_ld = dx0.data_ptr(), rbits.data_ptr(), Cin
    ap, aw = h2_planes(lib, gyd.view(-1, Cout), 0)
    wp, ww = h2_planes(lib, wt.view(Cin, -1), 1)
    d.f32_split = 2
    d.x_planes, d.x_unscale, d.x_plane_stride = ap.data_ptr(), aw.data_ptr() + 4096, ap.shape[1]
    d.w_planes, d.w_unscale, d.w_plane_stride = wp.data_ptr(), ww.data_ptr() + 4096, wp.shape[1]
    assert lib.dml_conv_stat_rows(C.byref(d)) == 48
    if bnr:
        d.bnr_y, d.bnr_mask, d.bnr_mean, d.bnr_invstd = ybn.data_ptr(), bits.data_ptr(), mean.data_ptr(), invstd.data_ptr()
        d.bnr_partials, d.bnr_ldy, d.bnr_relu, d.bnr_gmax = part.data_ptr(), Cin, relu, gmax.data_ptr()
    guard = dx0.clone() if accum == 2 else None
    chk(lib.dml_conv_igemm(C.byref(d), st()))
    torch.cuda.synchronize()
    if guard is not None:
        assert torch.equal(guard, dx0)                 # (the identity gradient is an input)
    # the stored tensor itself: conv (+ the earlier value)
    ref = torch.nn.functional.conv_transpose2d(gyd.permute(0, 3, 1, 2).double(), wt.permute(3, 0, 1, 2).double(), padding=1)
    ref = ref.permute(0, 2, 3, 1)
    if accum == 1:
        ref = ref + dx0.double()
    elif accum == 2:
        rmk = ((rbits.view(M, Cin // 4, 1) >> torch.arange(4, device="cuda").view(1, 1, 4)) & 1).reshape(B, Hh, Ww, Cin).double()
        ref = ref + dx0.double() * rmk
    assert (dx.double() - ref).abs().max().item() <= 2e-6 * ref.abs().max().item()
    if not bnr:
        assert torch.all(part == 7.0) and gmax.max().item() == 0.0
        return
    g = dx.view(M, Cin).double()
    if relu:
        mk = ((bits.view(M, Cin // 4, 1) >> torch.arange(4, device="cuda").view(1, 1, 4)) & 1).reshape(M, Cin).double()
        g = g * mk
    xh = (ybn.double() - mean.double()) * invstd.double()
    pad = G * 48 - M
    gp = torch.cat([g, torch.zeros(pad, Cin, device="cuda", dtype=torch.float64)]).view(G, 48, Cin)
    xp = torch.cat([xh, torch.zeros(pad, Cin, device="cuda", dtype=torch.float64)]).view(G, 48, Cin)
    want = torch.stack([gp.sum(1), (gp * xp).sum(1)], dim=-1)
    relclose(part.view(G, Cin, 2).cpu(), want.cpu(), 1e-5, "per-group partials")
    assert abs(gmax.max().item() - g.abs().max().item()) <= 1e-6 * g.abs().max().item()
    # against the stand-alone reduce on the stored tensor
    part2 = torch.zeros(4096 * Cin * 2, device="cuda")
    nb = C.c_int(0)
    chk(lib.dml_bn_bwd_reduce(dx.data_ptr(), ybn.data_ptr(), None, bits.data_ptr(), mean.data_ptr(), invstd.data_ptr(),
                              part2.data_ptr(), M, Cin, Cin, Cin, Cin, relu, 1.0, 0, C.byref(nb), None, st()))
    torch.cuda.synchronize()
    a = part.view(G, Cin, 2).double().sum(0)
    b = part2[: nb.value * Cin * 2].view(nb.value, Cin, 2).double().sum(0)
    relclose(a.cpu(), b.cpu(), 2e-6, "sums vs dml_bn_bwd_reduce")
    # without planes (exact fp32 / three-term kernels) the request is refused, not ignored
    d.f32_split, d.x_planes = 1, None
    assert lib.dml_conv_igemm(C.byref(d), st()) == -3


def _tile_major(mat):
    """[rows][K] -> [rows / 64][K / 32][64][32] (DmlPrepDesc.w_tiled)"""
    rows, K = mat.shape
    return mat.reshape(rows // 64, 64, K // 32, 32).permute(0, 2, 1, 3).contiguous()


@pytest.mark.parametrize("case", [("t1x1", 2, 24, 20, 64, 128, 1, 1, 1), ("t1x1_n64", 3, 17, 13, 256, 64, 1, 1, 1),
                                  ("t3x3", 2, 19, 23, 64, 192, 3, 1, 1), ("t3x3_d2", 2, 16, 16, 128, 256, 3, 1, 2),
                                  ("t3x3_s2", 2, 22, 18, 128, 128, 3, 2, 1), ("t_rows256", 4, 40, 40, 256, 256, 3, 1, 1)],
                         ids=lambda c: c[0])
def test_tile_major_weights_are_bit_identical(lib, case):
    """DmlPrepDesc.w_tiled / wt_tiled + DmlConvDesc.w_tiled: dml_prep_weights writes the tile-major copies ([rows / 64][K / 32]
    [64][32]); forward (with BN statistics) and data gradient through the LDS-DMA kernels must equal the plain layout bit for
    bit (same products, same order; only the addresses of the weight tile's DMA change).  Kernels without the layout refuse."""
    from dmlnet._lib import PrepDesc, ConvDesc
    name, B, Hh, Ww, Cin, Cout, k, stride, dil = case
    bf = torch.bfloat16
    master = rnd(name + ".w", (Cout, k, k, Cin), scale=(2.0 / (Cin * k * k)) ** 0.5).cuda().contiguous()
    RS = k * k
    bufs = {}
    for tiled in (0, 1):
        w = torch.empty(Cout * RS * Cin, device="cuda", dtype=bf)
        wt = torch.empty(Cout * RS * Cin, device="cuda", dtype=bf)
        arr = (PrepDesc * 1)(PrepDesc(master.data_ptr(), w.data_ptr(), wt.data_ptr(), Cout, RS, Cin, Cin, tiled, tiled))
        tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).cuda()
        chk(lib.dml_prep_weights(tab.data_ptr(), 1, 1, st()))
        torch.cuda.synchronize()
        bufs[tiled] = (w, wt)
    # the layout itself
    w_plain = bufs[0][0].view(Cout, RS * Cin)
    wt_plain = bufs[0][1].view(Cin, RS * Cout)
    assert torch.equal(bufs[1][0].view(-1), _tile_major(w_plain).view(-1))
    assert torch.equal(bufs[1][1].view(-1), _tile_major(wt_plain).view(-1))
    x = nhwc(rnd(name + ".x", (B, Cin, Hh, Ww)), bf)
    pad = dil * (k // 2)
    Ho, Wo = (Hh + 2 * pad - dil * (k - 1) - 1) // stride + 1, (Ww + 2 * pad - dil * (k - 1) - 1) // stride + 1
    M = B * Ho * Wo
    gy = nhwc(rnd(name + ".gy", (B, Cout, Ho, Wo)), bf)
    tws = torch.empty(512 * 128 * 128, device="cuda")
    tcnt = torch.zeros(128, dtype=torch.int32, device="cuda")
    outs = {}
    for tiled in (0, 1):
        y = torch.empty((B, Ho, Wo, Cout), device="cuda", dtype=bf)
        stats = torch.zeros((M + 63) // 64 * Cout * 2, device="cuda")
        d = make_desc(lib, x, bufs[tiled][0], y, B, Hh, Ww, Cin, Ho, Wo, Cout, k, stride, dil, pad, 1, stats=stats)
        d.w_tiled = tiled
        d.tail_ws, d.tail_ws_elems, d.tail_counters, d.tail_counters_len = tws.data_ptr(), tws.numel(), tcnt.data_ptr(), 128
        chk(lib.dml_conv_igemm(C.byref(d), st()))
        dx = torch.empty((B, Hh, Ww, Cin), device="cuda", dtype=bf)
        dd = make_desc(lib, gy, bufs[tiled][1], dx, B, Ho, Wo, Cout, Hh, Ww, Cin, k, stride, dil, pad, 1, mode=1)
        dd.w_tiled = tiled
        dd.tail_ws, dd.tail_ws_elems, dd.tail_counters, dd.tail_counters_len = tws.data_ptr(), tws.numel(), tcnt.data_ptr(), 128
        chk(lib.dml_conv_igemm(C.byref(dd), st()))
        torch.cuda.synchronize()
        outs[tiled] = (y, stats, dx)
    for a_, b_, what in zip(outs[0], outs[1], ("forward", "statistics", "data gradient")):
        assert torch.equal(a_, b_), "%s: %s differs with tile-major weights" % (name, what)
    # refused where no LDS-DMA kernel would read the layout: fp32, C % 32 != 0, N % 64 != 0
    bad = make_desc(lib, x.float(), master, torch.empty((B, Ho, Wo, Cout), device="cuda"), B, Hh, Ww, Cin, Ho, Wo, Cout, k, stride,
                    dil, pad, 0)
    bad.w_tiled = 1
    assert lib.dml_conv_igemm(C.byref(bad), st()) != 0
    d.N = Cout - 16
    assert lib.dml_conv_igemm(C.byref(d), st()) != 0
