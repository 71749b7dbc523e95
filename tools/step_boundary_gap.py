"""GPU box: is the GPU idle at the step boundary (between the fused SGD of step k and the first launch of step k + 1)?  Event pairs on
the main stream: elapsed(end of step k, start of step k + 1) is ~0 when the host is ahead of the device.   python3 tools/step_boundary_gap.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, network, utils
from dmlnet.optim import FusedSGD
dev = torch.device("cuda", 0)
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).to(dev)
m.set_compute_dtype(torch.float32, fp32_products="f16x2")
m.train()
utils.set_bn_momentum(m.backbone, momentum=0.01)
opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 1e-3}, {"params": m.classifier.parameters(), "lr": 1e-2}], lr=1e-2, momentum=0.9,
               weight_decay=1e-4).bind(m)
sched = utils.PolyLR(opt, 30000, power=0.9)
crit = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=True)
img = torch.randn(16, 3, 768, 768, device=dev)
lab = torch.randint(0, 16, (16, 768, 768), device=dev)
N = 12
ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(N)]
host = []
for k in range(N):
    h0 = time.perf_counter()
    ev[k][0].record()
    opt.zero_grad()
    lg, ctr, ft = m(img)
    ev[k][1].record()
    loss = crit(lg, lab, ft)
    loss.backward()
    ev[k][2].record()
    opt.step()
    sched.step()
    ev[k][3].record()
    host.append((time.perf_counter() - h0) * 1e3)
torch.cuda.synchronize()
for k in range(4, N):
    print("step %2d: boundary gap %.3f ms | forward %.2f  backward %.2f  sgd %.2f | host enqueue %.2f ms" % (
        k, ev[k - 1][3].elapsed_time(ev[k][0]), ev[k][0].elapsed_time(ev[k][1]), ev[k][1].elapsed_time(ev[k][2]),
        ev[k][2].elapsed_time(ev[k][3]), host[k]))
