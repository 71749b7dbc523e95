"""Locate the host code that issues D2D copies during a train step (torch profiler with stacks)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
import torch, network, utils
from dmlnet.optim import FusedSGD
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda")
torch.manual_seed(1)
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).to(dev)
m.set_compute_dtype(torch.bfloat16); m.train()
opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.001}, {"params": m.classifier.parameters(), "lr": 0.01}],
               lr=0.01, momentum=0.9, weight_decay=1e-4).bind(m)
crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
B_, S_ = int(os.environ.get("D2D_BATCH", "2")), int(os.environ.get("D2D_SIZE", "128"))
img = torch.randn(B_, 3, S_, S_, device=dev); lab = torch.randint(0, 16, (B_, S_, S_), device=dev)
def step():
    opt.zero_grad(); lg, c, f = m(img); loss = crit(lg, lab, f); loss.backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
import collections
names = collections.Counter(e.name for e in prof.events())
for n, k in names.most_common(60):
    if k >= 50: print(k, n[:100])
print("---- copy-like events with stacks")
seen = collections.Counter()
for e in prof.events():
    if "copy" in e.name.lower() or "memcpy" in e.name.lower():
        st = tuple(s for s in (e.stack or [])[:6])
        seen[(e.name[:60], st)] += 1
for (n, st), k in seen.most_common(12):
    print(k, n)
    for s in st:
        print("      ", s[:140])
