"""Where does the decoder's 3x3 conv weight gradient deviate from the fp64 oracle?  Compare x, dz, dy of that unit."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
import torch, helpers as H, network, utils
from oracle import dmlnet_ref as O
torch.set_num_threads(32)
shape = tuple(int(v) for v in os.environ.get("DBG_SHAPE", "2,3,64,80").split(","))
seed = int(os.environ.get("DBG_SEED", "21")); tag = os.environ.get("DBG_TAG", "var")
img = H.synth_tensor(seed, tag + ".img", shape); lab = H.synth_labels(seed, tag + ".lab", (shape[0],) + shape[2:], 16, 255, ignore_frac=0.05)
m = network.deeplabv3plus_embedding_resnet101(16, 16, False); m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=seed))
m.cuda().train(); m.classifier.aspp.project[3].eval(); m.set_compute_dtype(torch.float32)
lg, _, ft = m(img.cuda()); loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft); loss.backward(); torch.cuda.synchronize()
plan = next(iter(m._engine.plans.values()))
def act(a):
    off = (a.ptr - a.t.data_ptr()) // a.es; flat = a.t.view(-1)
    idx = off + torch.arange(a.M, device=flat.device).unsqueeze(1) * a.ld + torch.arange(a.C, device=flat.device).unsqueeze(0)
    return flat[idx].double().cpu().view(a.B, a.H, a.W, a.C).permute(0, 3, 1, 2)
o = O.deeplabv3plus_embedding_resnet101(16, 16); o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=seed)); o = o.double(); o.train(); o.classifier.aspp.project[3].eval()
cap = {}
cls = o.classifier.classifier
def fwd_hook(name):
    def h(mod, inp, out):
        cap[name + ".in"] = inp[0].detach()
        out.register_hook(lambda g: cap.__setitem__(name + ".gout", g.detach()))
        if inp[0].requires_grad: inp[0].register_hook(lambda g: cap.__setitem__(name + ".gin", g.detach()))
    return h
for i in range(4): cls[i].register_forward_hook(fwd_hook("c%d" % i))
olg, _, oft = o(img.double()); ol = O.dml_loss(olg, lab, alpha=0.01, ignore_index=255); ol.backward()
u = [u for u in plan.units if u.conv is m.classifier.classifier[0]][0]
def rel(a, b, what):
    print("%-28s max|d| %.3e  scale %.3e  rel %.3e" % (what, (a - b).abs().max().item(), b.abs().max().item(), (a - b).abs().max().item() / b.abs().max().item()))
x = act(u.x)[:, :304]
rel(x, cap["c0.in"], "cat2 (conv input)")
rel(x[:, :48], cap["c0.in"][:, :48], "  low-level part")
rel(x[:, 48:], cap["c0.in"][:, 48:], "  upsampled ASPP part")
rel(act(u.y), cap["c1.in"], "conv output y")
rel(act(u.z), cap["c3.in"], "z (after BN+ReLU)")
rel(act(u.dz), cap["c2.gout"], "dz (grad at ReLU output)")
rel(act(u.dy), cap["c0.gout"], "dy (grad at conv output)")
gw = m.classifier.classifier[0].weight.grad.double().cpu(); ow = cls[0].weight.grad
rel(gw, ow, "weight grad")
d = (gw - ow).abs()
print("weight-grad error by input-channel block: low-level %.3e, upsampled %.3e; by tap:" % (d[:, :48].max().item(), d[:, 48:].max().item()), [float("%.2e" % d[:, :, r, s].max().item()) for r in range(3) for s in range(3)])
