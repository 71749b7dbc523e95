"""Debug helper (GPU box): in-situ self-consistency of every conv+BN unit of a backward pass.
For each unit recompute, in fp64 torch on the CPU from the HIP path's OWN buffers, what the BN backward and the
weight gradient should be, and compare with what the kernels wrote."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, torch.nn.functional as F
import helpers as H
import network, utils

shape, seed = (2, 3, 64, 96), 9
if os.environ.get("DBG_SHAPE"):
    shape = tuple(int(v) for v in os.environ["DBG_SHAPE"].split(","))
    seed = int(os.environ.get("DBG_SEED", "9"))
dtype = torch.bfloat16 if (len(sys.argv) > 1 and sys.argv[1] == "bf16") else torch.float32
torch.set_num_threads(32)
tag = os.environ.get("DBG_TAG", "fresh")
img = H.synth_tensor(seed, tag + ".img", shape)
lab = H.synth_labels(seed, tag + ".lab", (shape[0],) + shape[2:], 16, 255, ignore_frac=0.05)
m = network.deeplabv3plus_embedding_resnet101(16, 16, False)
m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=seed))
m.cuda().train(); m.classifier.aspp.project[3].eval(); m.set_compute_dtype(dtype)
lg, _, ft = m(img.cuda())
loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
loss.backward()
torch.cuda.synchronize()
plan = next(iter(m._engine.plans.values()))
names = {id(mod): n for n, mod in m.named_modules()}

def act(a):     # Act -> [M, C] double on cpu
    es = a.es
    off = (a.ptr - a.t.data_ptr()) // es
    flat = a.t.view(-1)
    M = a.M
    idx = off + torch.arange(M, device=flat.device).unsqueeze(1) * a.ld + torch.arange(a.C, device=flat.device).unsqueeze(0)
    return flat[idx].double().cpu()

print("%-40s %10s %10s %10s %10s" % ("unit", "bn dy", "dgamma", "wgrad", "dgrad"))
from collections import Counter
consumers = Counter(id(u.x.root) for u in plan.units)
for u in plan.units:
    n = names[id(u.conv)]
    dz, y, z, dy = act(u.dz), act(u.y), act(u.z), act(u.dy)
    g = dz * (z > 0) if u.relu else dz          # (the kernels read the same predicate from the bitmask in bf16)
    mu, inv = u.mean.double().cpu(), u.invstd.double().cpu()
    xhat = (y - mu) * inv
    gam = u.bn.weight.detach().double().cpu()
    M = y.shape[0]
    dbeta, dgamma = g.sum(0), (g * xhat).sum(0)
    dy_ref = gam * inv * (g - dbeta / M - xhat * dgamma / M)
    e_dy = (dy - dy_ref).abs().max().item() / (dy_ref.abs().max().item() + 1e-30)
    e_dg = (u.bn.weight.grad.double().cpu() - dgamma).abs().max().item() / (dgamma.abs().max().item() + 1e-30)
    # weight gradient from the HIP path's own x and dy
    x = act(u.x).view(u.x.B, u.x.H, u.x.W, u.x.C).permute(0, 3, 1, 2)
    cin = u.conv.in_channels
    x = x[:, :cin]
    dyn = dy.view(u.y.B, u.y.H, u.y.W, -1).permute(0, 3, 1, 2)
    xr = x.clone().requires_grad_(False)
    w = u.conv.weight.detach().double().cpu().contiguous().requires_grad_(True)
    out = F.conv2d(xr, w, None, u.conv.stride, u.conv.padding, u.conv.dilation)
    out.backward(dyn)
    e_w = (u.conv.weight.grad.double().cpu() - w.grad).abs().max().item() / (w.grad.abs().max().item() + 1e-30)
    e_dx = float("nan")
    if consumers[id(u.x.root)] == 1 and u.x.root.grad is not None and u.x is u.x.root and "layer" in n and "conv1" not in n:
        xin = torch.zeros((u.x.B, cin, u.x.H, u.x.W), dtype=torch.float64, requires_grad=True)
        F.conv2d(xin, u.conv.weight.detach().double().cpu().contiguous(), None, u.conv.stride, u.conv.padding,
                 u.conv.dilation).backward(dyn)
        got = act(u.x.grad).view(u.x.B, u.x.H, u.x.W, -1).permute(0, 3, 1, 2)[:, :cin]
        e_dx = (got - xin.grad).abs().max().item() / (xin.grad.abs().max().item() + 1e-30)
    flag = " <<<" if max(e_dy, e_dg, e_w, e_dx if e_dx == e_dx else 0) > (1e-4 if dtype == torch.float32 else 2e-2) else ""
    print("%-40s %10.2e %10.2e %10.2e %10.2e%s" % (n, e_dy, e_dg, e_w, e_dx, flag))
