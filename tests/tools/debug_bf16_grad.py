"""Debug (GPU box): where does a bf16-mode parameter gradient come from?  For chosen BatchNorm layers print the HIP
gradient, the same sums recomputed in fp64 from the plan's stored dz / y / mask (kernel-level consistency), the fp32
oracle and the bf16-storage oracle.  usage: python tests/tools/debug_bf16_grad.py 4x3x256x256 9"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import numpy as np, torch
import helpers as H
import test_gpu_bf16_parity as P
import utils

shape = tuple(int(v) for v in sys.argv[1].split("x")); seed = int(sys.argv[2]); tag = sys.argv[3] if len(sys.argv) > 3 else "bf16.256"
torch.set_num_threads(64)
img = H.synth_tensor(seed, tag + ".img", shape)
lab = H.synth_labels(seed, tag + ".lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
m = P._build_hip(torch.bfloat16, seed)
lg, _, ft = m(img.cuda())
utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft).backward()
torch.cuda.synchronize()
_, _, g_true, _ = P._oracle_grads(seed, shape, tag, emulate=False)
_, _, g_emu, _ = P._oracle_grads(seed, shape, tag, emulate=True)
plan = next(p for k, p in m._engine.plans.items() if k[4])
names = {id(mod): n for n, mod in m.named_modules()}
want = sys.argv[4].split(",") if len(sys.argv) > 4 else ["backbone.layer4.2.bn3", "backbone.layer4.1.bn3", "backbone.layer4.0.bn3",
                                                       "backbone.layer3.22.bn3", "backbone.layer3.8.bn1", "classifier.aspp.convs.0.1"]
def nrm(t): return float(t.double().norm())
for u in plan.units:
    n = names[id(u.bn)]
    if n not in want: continue
    dz = P._nchw(P._act(u.dz), u.z.B, u.z.H, u.z.W).double(); z = P._nchw(P._act(u.z), u.z.B, u.z.H, u.z.W)
    y = P._nchw(P._act(u.y), u.y.B, u.y.H, u.y.W).double()
    g = dz * (z > 0) if u.relu else dz
    mu, inv = u.mean.double().cpu(), u.invstd.double().cpu()
    xhat = (y - mu.view(1, -1, 1, 1)) * inv.view(1, -1, 1, 1)
    db_loc, dg_loc = g.sum((0, 2, 3)), (g * xhat).sum((0, 2, 3))
    db_hip, dg_hip = u.bn.bias.grad.double().cpu(), u.bn.weight.grad.double().cpu()
    for what, hip, loc in (("bias", db_hip, db_loc), ("weight", dg_hip, dg_loc)):
        k = n + "." + what
        t, e = g_true[k], g_emu[k]
        print("%-36s |true| %.3e |emu| %.3e |hip| %.3e |local| %.3e  hip-local %.2e  hip-true %.2e emu-true %.2e  (frac relu on %.3f, |dz| rms %.3e, M %d)"
              % (k, nrm(t), nrm(e), nrm(hip), nrm(loc), nrm(hip - loc) / (nrm(loc) + 1e-30), nrm(hip - t) / nrm(t), nrm(e - t) / nrm(t),
                 float((z > 0).float().mean()), float(dz.pow(2).mean().sqrt()), dz.numel() // dz.shape[1]))
