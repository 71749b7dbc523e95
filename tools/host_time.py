"""How long does the host take to ENQUEUE one train step (vs the GPU's 45 ms)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
import torch, network, utils
from dmlnet.optim import FusedSGD
dev = torch.device("cuda")
torch.manual_seed(1)
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).to(dev)
m.set_compute_dtype(torch.bfloat16); m.train()
opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.001}, {"params": m.classifier.parameters(), "lr": 0.01}], lr=0.01, momentum=0.9, weight_decay=1e-4).bind(m)
crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
g = torch.Generator().manual_seed(1234)
img = torch.randn(16, 3, 768, 768, generator=g).to(dev); lab = torch.randint(0, 16, (16, 768, 768), generator=g).to(dev)
def step():
    opt.zero_grad(); lg, c, f = m(img); loss = crit(lg, lab, f); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
host = []
t_all = time.perf_counter()
for _ in range(10):
    t0 = time.perf_counter(); step(); host.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tot = (time.perf_counter() - t_all) / 10
print("host enqueue per step: %s ms; wall per step %.2f ms" % (", ".join("%.1f" % (h * 1e3) for h in host), tot * 1e3))
