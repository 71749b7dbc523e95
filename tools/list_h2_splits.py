"""GPU box: which tensors of the f16x2 train plan (16 x 768 x 768) still go through dml_h2_split (an amax pass + a split pass each)
instead of having their fp16 planes written by their producer.    python3 tools/list_h2_splits.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, network, utils
from dmlnet import engine as E
orig = E.Plan.h2_of
def h2_of(self, a, ops):
    root = a.root
    fresh = root.h2 is None
    r = orig(self, a, ops)
    if fresh:
        print("h2_split: root M=%d C=%d ld=%d (%.0f MB fp32) amax_known=%s list=%s" % (root.M, root.C, root.ld, root.M * root.ld * 4 / 1e6, root.amax is not None, "fwd" if ops is self.fwd else "bwd"))
    return r
E.Plan.h2_of = h2_of
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).cuda()
m.set_compute_dtype(torch.float32, fp32_products="f16x2")
m.train()
x = torch.randn(16, 3, 768, 768, device="cuda")
lab = torch.randint(0, 16, (16, 768, 768), device="cuda")
lg, _, ft = m(x)
loss = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=True)(lg, lab, ft)
loss.backward()
