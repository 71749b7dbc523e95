"""GPU box: where does the host wait when the data-parallel path is active (one forced RCCL rank)?  Per-phase host time of a drained-queue step.
    DML_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29533 RANK=0 LOCAL_RANK=0 WORLD_SIZE=1 python3 tools/host_time_dist.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, torch.distributed as dist
from dmlnet import parallel
import network, utils
from dmlnet.optim import FusedSGD
rank, local, world = parallel.init_from_env()
dev = torch.device("cuda", local)
dp = dist.is_initialized()
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).to(dev)
m.set_compute_dtype(torch.float32, fp32_products="f16x2")
m.train()
opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 1e-3}, {"params": m.classifier.parameters(), "lr": 1e-2}], lr=1e-2, momentum=0.9,
               weight_decay=1e-4).bind(m)
crit = utils.DMLLoss(alpha=0.01, ignore_index=255, sync=True if dp else None, fused_backward=True)
if dp:
    m._engine.store.bind(dev)
    m._engine.reducer = parallel.GradReducer(m._engine.store, bucket_mb=32.0, average=False)
img = torch.randn(16, 3, 768, 768, device=dev)
lab = torch.randint(0, 16, (16, 768, 768), device=dev)
for it in range(6):
    torch.cuda.synchronize()
    t = [time.perf_counter()]
    opt.zero_grad(); t.append(time.perf_counter())
    lg, ctr, ft = m(img); t.append(time.perf_counter())
    loss = crit(lg, lab, ft); t.append(time.perf_counter())
    loss.backward(); t.append(time.perf_counter())
    opt.step(); t.append(time.perf_counter())
    torch.cuda.synchronize(); t.append(time.perf_counter())
    if it >= 3:
        print("dist=%s  zero_grad %.2f  forward %.2f  loss %.2f  backward %.2f  opt.step %.2f  | drain %.2f ms" % (
            dp, *[(t[i + 1] - t[i]) * 1e3 for i in range(6)]))
