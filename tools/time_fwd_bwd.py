"""GPU box: wall time of the forward (incl. loss) and of the backward + optimizer of the headline train step (f16x2, 16 x 768 x 768),
from CUDA events on the main stream.    python3 tools/time_fwd_bwd.py [dtype=f16x2]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch
import network, utils
from dmlnet.optim import FusedSGD
mode = sys.argv[1] if len(sys.argv) > 1 else "f16x2"
dev = torch.device("cuda", 0)
torch.manual_seed(1)
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).to(dev)
if mode == "bf16":
    m.set_compute_dtype(torch.bfloat16)
else:
    m.set_compute_dtype(torch.float32, fp32_products=mode)
m.train()
utils.set_bn_momentum(m.backbone, momentum=0.01)
opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.001}, {"params": m.classifier.parameters(), "lr": 0.01}],
               lr=0.01, momentum=0.9, weight_decay=1e-4).bind(m)
crit = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=True)
g = torch.Generator().manual_seed(1234)
img = torch.randn(16, 3, 768, 768, generator=g).to(dev)
lab = torch.randint(0, 16, (16, 768, 768), generator=g)
lab[:, :38] = 255
lab = lab.to(dev)
ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
tf = tb = 0.0
N = 20
for it in range(N + 4):
    ev[0].record()
    opt.zero_grad()
    lg, _, ft = m(img)
    loss = crit(lg, lab, ft)
    ev[1].record()
    loss.backward()
    opt.step()
    ev[2].record()
    torch.cuda.synchronize()
    if it >= 4:
        tf += ev[0].elapsed_time(ev[1]); tb += ev[1].elapsed_time(ev[2])
print("%s: forward + loss %.2f ms, backward + SGD %.2f ms, step %.2f ms (synchronised every step)" % (mode, tf / N, tb / N, (tf + tb) / N))
