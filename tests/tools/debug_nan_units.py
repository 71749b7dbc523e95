"""Debug (GPU box): after one bf16 train step, list the units whose stored tensors hold NaN / Inf (forward and backward
order) and compare each forward conv output with an fp32 torch conv on the GPU.  usage: python tests/tools/debug_nan_units.py 2x3x768x768 77"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, torch.nn.functional as F
import helpers as H
import test_gpu_bf16_parity as P
import utils
shape = tuple(int(v) for v in sys.argv[1].split("x")); seed = int(sys.argv[2])
img = H.synth_tensor(seed, "unit.img", shape)
lab = H.synth_labels(seed, "unit.lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
m = P._build_hip(torch.bfloat16, seed)
lg, _, ft = m(img.cuda())
utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft).backward()
torch.cuda.synchronize()
plan = next(p for k, p in m._engine.plans.items() if k[4])
names = {id(mod): n for n, mod in m.named_modules()}
def bad(a):
    if a is None: return "-"
    off = (a.ptr - a.t.data_ptr()) // a.es
    v = a.t.view(-1)[off: off + (a.M - 1) * a.ld + a.C].float()
    v = v.view(-1)
    n = int((~torch.isfinite(v)).sum())
    return "%d" % n
def gpu(a):
    off = (a.ptr - a.t.data_ptr()) // a.es
    flat = a.t.view(-1)
    idx = off + torch.arange(a.M, device=flat.device).unsqueeze(1) * a.ld + torch.arange(a.C, device=flat.device).unsqueeze(0)
    return flat[idx].float().view(a.B, a.H, a.W, a.C).permute(0, 3, 1, 2)
print("%-40s %8s %8s %8s %8s  conv-vs-torch  mean-err" % ("unit", "nan y", "nan z", "nan dz", "nan dy"))
for u in plan.units:
    n = names[id(u.conv)]
    conv = u.conv
    x = gpu(u.x)[:, :conv.in_channels]
    w = conv.weight.detach().float()
    if u.dtype == torch.bfloat16: w = w.to(torch.bfloat16).float()
    yr = F.conv2d(x, w, None, conv.stride, conv.padding, conv.dilation)
    y = gpu(u.y)
    e = ((y - yr).abs().max() / (yr.abs().max() + 1e-30)).item()
    me = ((u.mean - yr.mean((0, 2, 3))).abs().max() / (yr.var((0, 2, 3), unbiased=False).sqrt().max() + 1e-30)).item()
    flag = "  <<<" if (e > 8e-3 or me > 1e-3 or bad(u.y) != "0" or bad(getattr(u, "dy", None)) not in ("0", "-")) else ""
    print("%-40s %8s %8s %8s %8s  %.2e  %.2e%s" % (n, bad(u.y), bad(u.z), bad(getattr(u, "dz", None)), bad(getattr(u, "dy", None)), e, me, flag))
