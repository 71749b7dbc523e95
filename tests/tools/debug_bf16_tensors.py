"""Debug (GPU box): bf16 gradient noise per ACTIVATION-gradient tensor.  For chosen conv + BN units: 1 - cos between
the HIP bf16 plan's stored gradients (dz (.) ReLU mask at the BatchNorm output, dy at the conv output) and the fp32
oracle's / the bf16-storage oracle's autograd gradients of the same tensors.  Locates the op where the plan's noise
departs from the emulation's.  usage: python tests/tools/debug_bf16_tensors.py 2x3x768x768 77 bf16.768 [prefix,prefix,...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import numpy as np, torch, torch.nn as nn
import helpers as H
import test_gpu_bf16_parity as P
import utils
from oracle import dmlnet_ref as O

shape = tuple(int(v) for v in sys.argv[1].split("x")); seed = int(sys.argv[2]); tag = sys.argv[3]
want = sys.argv[4].split(",") if len(sys.argv) > 4 else ["backbone.layer4", "classifier.aspp", "backbone.layer3.22", "backbone.layer3.21",
                                                       "classifier.classifier", "classifier.project", "backbone.layer3.0", "backbone.layer2.0"]
torch.set_num_threads(64)
img = H.synth_tensor(seed, tag + ".img", shape)
lab = H.synth_labels(seed, tag + ".lab", (shape[0], shape[2], shape[3]), 16, 255, ignore_frac=0.05)
m = P._build_hip(torch.bfloat16, seed)
lg, _, ft = m(img.cuda())
utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft).backward()
torch.cuda.synchronize()
plan = next(p for k, p in m._engine.plans.items() if k[4])
names = {id(mod): n for n, mod in m.named_modules()}


def oracle_tensor_grads(emulate):
    o = P._build_oracle(seed, emulate=emulate)
    store = {}
    for n, mod in o.named_modules():
        if isinstance(mod, (nn.Conv2d, nn.BatchNorm2d)) and any(n.startswith(w) for w in want):
            def hook(mod, inp, out, n=n):
                if out.requires_grad:
                    out.register_hook(lambda g, n=n: store.__setitem__(n, g.detach().clone()))
            mod.register_forward_hook(hook)
    olg, _, _ = o(img)
    O.dml_loss(olg, lab, alpha=0.01, ignore_index=255).backward()
    return store


g_true = oracle_tensor_grads(False)
g_emu = oracle_tensor_grads(True)


def omc(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return 1.0 - (a @ b).item() / (a.norm().item() * b.norm().item() + 1e-30)


print("%-40s %-8s %10s %10s %10s %7s   |hip|/|true| |emu|/|true|" % ("unit (backward order)", "tensor", "hip-true", "emu-true", "hip-emu", "ratio"))
for u in reversed(plan.units):
    cn, bn_n = names[id(u.conv)], names[id(u.bn)]
    if cn not in g_true or getattr(u, "dz", None) is None:
        continue
    dz = P._nchw(P._act(u.dz), u.z.B, u.z.H, u.z.W)
    z = P._nchw(P._act(u.z), u.z.B, u.z.H, u.z.W)
    g = dz * (z > 0) if u.relu else dz
    dy = P._nchw(P._act(u.dy), u.y.B, u.y.H, u.y.W)
    for what, hip, key in (("bn out", g, bn_n), ("conv out", dy, cn)):
        if key not in g_true or key not in g_emu:
            continue
        t, e = g_true[key], g_emu[key]
        if what == "conv out" and u.dtype == torch.bfloat16:
            e = e.to(torch.bfloat16).float()          # the emulation stores this gradient rounded (_QGrad runs upstream of the hook)
        a, b, c = omc(hip, t), omc(e, t), omc(hip, e)
        print("%-40s %-8s %10.3e %10.3e %10.3e %7.2f   %.3f %.3f" % (cn, what, a, b, c, a / (b + 1e-30),
                                                                   hip.double().norm() / t.double().norm(), e.double().norm() / t.double().norm()))
