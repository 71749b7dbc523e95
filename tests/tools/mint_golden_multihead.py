"""Mint tests/golden/g12_multihead.npz from the reference's self-distillation model (authoring container only):
network.deeplabv3plus_embedding_self_distillation_resnet101 (modeling.py:150-158, utils.py:120-193) -- shared backbone,
heads with 16 and 17 prototypes, lists out -- one train-mode step with the loss on the LAST head only, as
main_self_distillation.py:447-507 does.  Also checks that the oracle restatement is identical."""
import os, sys
from collections import OrderedDict
import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tools")]
import helpers as H  # noqa: E402
import mint_golden as MG  # noqa: E402
from oracle import dmlnet_ref as O  # noqa: E402

torch.set_num_threads(8)
MG.install_shims()
sys.path.insert(0, os.path.join(MG.REF, "DeepLabV3Plus-Pytorch"))
import network as R  # noqa: E402  (the reference package)

ref = R.deeplabv3plus_embedding_self_distillation_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
orc = O.deeplabv3plus_embedding_self_distillation_resnet101(output_stride=16)
assert list(ref.state_dict().keys()) == list(orc.state_dict().keys()), "state_dict keys differ"
shapes = H.shapes_of(ref)
assert shapes == H.shapes_of(orc)
sd = H.synth_state_dict(shapes, seed=12)
for m in (ref, orc):
    m.load_state_dict(sd)
    m.train()
    m.classifier.aspp.project[3].eval()
    m.classifier_1.aspp.project[3].eval()
img = H.synth_tensor(12, "g12.img", (2, 3, 64, 64))
lab = H.synth_labels(12, "g12.lab", (2, 64, 64), 17, 255, ignore_rows=3)
out = {}
res = {}
for name, m in (("ref", ref), ("orc", orc)):
    logits, centers, feats = m(img)
    loss = O.ce_over_n(logits[-1], lab, 255)              # utils/loss.py:34-42 with alpha = 0, on the last head
    loss.backward()
    res[name] = (logits, centers, feats, loss, OrderedDict((k, p.grad) for k, p in m.named_parameters()))
(lg, ctr, ft, loss, g), (olg, octr, oft, oloss, og) = res["ref"], res["orc"]
for h in range(2):
    MG.assert_close(olg[h], lg[h].detach(), 1e-6, "logits head %d" % h)
    MG.assert_close(oft[h], ft[h].detach(), 1e-6, "features head %d" % h)
    assert torch.equal(octr[h], ctr[h])
assert abs(float(loss) - float(oloss)) < 1e-6
none_ref = sorted(k for k, v in g.items() if v is None)
none_orc = sorted(k for k, v in og.items() if v is None)
assert none_ref == none_orc and all(k.startswith("classifier.") for k in none_ref) and len(none_ref) > 0
for k in g:
    if g[k] is not None:
        MG.assert_close(og[k], g[k], 1e-5, "grad " + k) if k in ("backbone.conv1.weight", "classifier_1.classifier.3.weight") else None
keys = ["backbone.conv1.weight", "backbone.layer3.5.conv2.weight", "backbone.layer4.2.bn3.weight",
        "classifier_1.aspp.convs.1.0.weight", "classifier_1.classifier.0.weight", "classifier_1.classifier.3.weight",
        "classifier_1.classifier.3.bias"]
save = dict(loss=float(loss), n_keys=len(sd), keys=np.array(list(sd.keys())[-4:]),
            logits0_sub=lg[0][:, :, ::4, ::4], logits1_sub=lg[1][:, :, ::4, ::4], feats1_sub=ft[1][:, ::4, ::4, :],
            logits0_checksum=H.checksum(lg[0]), logits1_checksum=H.checksum(lg[1]),
            grad_keys=np.array(keys), untouched=np.array(none_ref[:3]))
for i, k in enumerate(keys):
    save["grad_%d" % i] = g[k] if g[k].numel() <= 70000 else g[k].reshape(-1)[::97]
    save["grad_%d_checksum" % i] = H.checksum(g[k])
MG.save("g12_multihead", **save)
print("state_dict entries:", len(sd), "loss", float(loss))
