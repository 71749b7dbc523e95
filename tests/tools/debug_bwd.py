"""Debug helper (GPU box): gradients of the head's intermediate tensors, HIP vs fp64 oracle vs fp32 oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, torch.nn.functional as F
import helpers as H
import network, utils
from oracle import dmlnet_ref as O

shape, seed = (2, 3, 64, 96), 9
torch.set_num_threads(32)
img = H.synth_tensor(9, "fresh.img", shape)
lab = H.synth_labels(9, "fresh.lab", (shape[0],) + shape[2:], 16, 255, ignore_frac=0.05)

def oracle(dtype):
    o = O.deeplabv3plus_embedding_resnet101(16, 16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=seed))
    o = o.to(dtype); o.train(); o.classifier.aspp.project[3].eval()
    keep = {}
    def hook(name):
        def f(m_, i_, out):
            out.retain_grad(); keep[name] = out
        return f
    o.classifier.classifier[3].register_forward_hook(hook("e"))
    o.classifier.classifier[2].register_forward_hook(hook("zc"))
    o.classifier.classifier[0].register_forward_hook(hook("yc"))
    o.classifier.aspp.register_forward_hook(hook("aspp"))
    o.classifier.project.register_forward_hook(hook("lowproj"))
    x = img.to(dtype)
    e_up = o.embed(x); e_up.retain_grad()
    lg, _, ft = O.distance_head(e_up)
    lg.retain_grad()
    loss = O.dml_loss(lg, lab, alpha=0.01, ignore_index=255)
    loss.backward()
    g = {k: v.grad.detach().double() for k, v in keep.items()}
    g["e_up"] = e_up.grad.detach().double(); g["lg"] = lg.grad.detach().double()
    return g

g64, g32 = oracle(torch.float64), oracle(torch.float32)
m = network.deeplabv3plus_embedding_resnet101(16, 16, False)
m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=seed))
m.cuda().train(); m.classifier.aspp.project[3].eval()
lg, _, ft = m(img.cuda())
lg.retain_grad()
loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
loss.backward()
torch.cuda.synchronize()
plan = next(iter(m._engine.plans.values()))
names = {id(mod): n for n, mod in m.named_modules()}
units = {names[id(u.conv)]: u for u in plan.units}
def nchw(act):
    return act.t.float().view(act.B, act.H, act.W, -1)[..., :act.C].permute(0, 3, 1, 2).cpu().double()
def rep(name, got, key):
    ref = g64[key]; sc = ref.abs().max().item()
    print("%-22s hip %.2e  ref32 %.2e  (scale %.2e)" % (name, (got - ref).abs().max().item() / sc, (g32[key] - ref).abs().max().item() / sc, sc))
rep("d logits", lg.grad.cpu().double(), "lg")
B, Hh, Ww = shape[0], shape[2], shape[3]
rep("d features (df)", plan.heads[0].df.view(B, Hh, Ww, 16).permute(0, 3, 1, 2).cpu().double(), "e_up")
ucls = units["classifier.classifier.0"]
fin_x = ucls.z
# de is the dy of the final conv: find through the grad chain
rep("d zc (dgrad final)", nchw(fin_x.grad), "zc")
cat2g = plan_cat2 = ucls.x.grad
rep("d lowproj (cat2[:48])", nchw(cat2g.slice(0, 48)) if False else cat2g.t.float().view(B, ucls.x.H, ucls.x.W, -1)[..., :48].permute(0, 3, 1, 2).cpu().double(), "lowproj")
