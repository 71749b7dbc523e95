"""Debug helper (GPU box): per-parameter gradient error of the HIP path vs the fp64 oracle, next to the
fp32 oracle's own error, to tell real bugs from fp32 conditioning."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
import helpers as H
import network, utils
from oracle import dmlnet_ref as O

shape = tuple(int(v) for v in (sys.argv[1:5] or (3, 3, 96, 128)))
seed = 9
torch.set_num_threads(32)

def oracle(dtype):
    o = O.deeplabv3plus_embedding_resnet101(16, 16)
    o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=seed))
    o = o.to(dtype); o.train(); o.classifier.aspp.project[3].eval()
    return o

img = H.synth_tensor(9, "fresh.img", shape)
lab = H.synth_labels(9, "fresh.lab", (shape[0],) + shape[2:], 16, 255, ignore_frac=0.05)
res = {}
for name, dt in (("f64", torch.float64), ("f32", torch.float32)):
    o = oracle(dt)
    lg, _, ft = o(img.to(dt))
    loss = O.dml_loss(lg, lab, alpha=0.01, ignore_index=255)
    loss.backward()
    res[name] = (lg.detach().double(), float(loss), {k: p.grad.double() for k, p in o.named_parameters()})

m = network.deeplabv3plus_embedding_resnet101(16, 16, False)
m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=seed))
m.cuda().train(); m.classifier.aspp.project[3].eval()
lg, _, ft = m(img.cuda())
loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft)
loss.backward()
torch.cuda.synchronize()
t = res["f64"]
print("loss hip %.6f f32 %.6f f64 %.6f" % (loss.item(), res["f32"][1], t[1]))
print("logits rel err: hip %.2e  ref32 %.2e" % (((lg.detach().cpu().double() - t[0]).abs().max() / t[0].abs().max()).item(),
      ((res["f32"][0] - t[0]).abs().max() / t[0].abs().max()).item()))
rows = []
for k, p in m.named_parameters():
    g64 = t[2][k]
    sc = g64.abs().max().item() + 1e-30
    eh = (p.grad.detach().cpu().double() - g64).abs().max().item() / sc
    er = (res["f32"][2][k] - g64).abs().max().item() / sc
    rows.append((eh, er, k, sc))
print("%-52s %10s %10s %10s" % ("param", "hip", "ref32", "scale"))
for eh, er, k, sc in rows:
    flag = " <<<" if eh > 10 * er + 1e-4 else ""
    print("%-52s %10.2e %10.2e %10.2e%s" % (k, eh, er, sc, flag))
