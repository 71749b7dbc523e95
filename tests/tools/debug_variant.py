import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import helpers as H, network, utils
from oracle import dmlnet_ref as O
K, OS = int(sys.argv[1]), int(sys.argv[2])
torch.set_num_threads(32)
m = network.deeplabv3plus_embedding_resnet101(num_classes=K, output_stride=OS, pretrained_backbone=False)
m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=21)); m.cuda(); m.set_compute_dtype(torch.float32); m.train()
m.classifier.aspp.project[3].eval(); utils.set_bn_momentum(m.backbone, 0.01)
img = H.synth_tensor(21, "var.img", (2, 3, 64, 80)); lab = H.synth_labels(21, "var.lab", (2, 64, 80), K, 255, ignore_frac=0.05)
lg, _, ft = m(img.cuda()); loss = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg, lab.cuda(), ft); loss.backward()
o = O.deeplabv3plus_embedding_resnet101(num_classes=K, output_stride=OS)
o.load_state_dict(H.synth_state_dict(H.shapes_of(o), seed=21)); o = o.double(); o.train(); o.classifier.aspp.project[3].eval()
olg, _, oft = o(img.double()); oloss = O.dml_loss(olg, lab, alpha=0.01, ignore_index=255); oloss.backward()
print("loss", loss.item(), float(oloss), "logits err", (lg.cpu().double() - olg).abs().max().item() / olg.abs().max().item())
errs = []
for (k, p), (k2, q) in zip(m.named_parameters(), o.named_parameters()):
    g = q.grad; sc = g.abs().max().item() + 1e-30
    errs.append(((p.grad.cpu().double() - g).abs().max().item() / sc, k, sc))
errs.sort(reverse=True)
for e in errs[:12]: print("%.3e  %-50s scale %.3e" % e)
print("median", np.median([e[0] for e in errs]))
# last params in forward order
for e in [x for x in errs if x[1].startswith("classifier.classifier")]: print("  head:", "%.3e %s" % (e[0], e[1]))
