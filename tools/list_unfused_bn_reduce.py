"""GPU box: which BatchNorm backward passes of the f16x2 train plan (16 x 768 x 768) still run a stand-alone dml_bn_bwd_reduce (a pass over
dz, y and the mask) instead of taking their sums from the epilogue of the data gradient that wrote dz, and which tensors still go through
dml_h2_split.    python3 tools/list_unfused_bn_reduce.py"""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, network, utils
m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).cuda()
m.set_compute_dtype(torch.float32, fp32_products="f16x2")
m.train()
x = torch.randn(16, 3, 768, 768, device="cuda")
lab = torch.randint(0, 16, (16, 768, 768), device="cuda")
lg, _, ft = m(x)
loss = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=True)(lg, lab, ft)
loss.backward()
plan = next(p for p in m._engine.plans.values() if p.training)
lib = plan.lib
names = {id(mod): n for n, mod in m.named_modules()}
tot = 0.0
for fn, args in plan.bwd:
    if fn is lib.dml_bn_bwd_reduce:
        M, N = args[7], args[8]
        mb = M * N * 8 / 1e6
        tot += mb
        u = next((u for u in plan.units if u.y is not None and u.y.ptr == args[1]), None)
        print("stand-alone bn_bwd_reduce: M=%d N=%d reads %.0f MB   %s" % (M, N, mb, names.get(id(u.bn), "?") if u is not None else "?"))
print("total %.0f MB per step (~%.2f ms at 5.5 TB/s)" % (tot, tot / 5.5e6))
for fn, args in list(plan.fwd) + list(plan.bwd):
    if fn is lib.dml_h2_split:
        print("h2_split: rows %d C %d ld %d (%.0f MB fp32), amax known %s" % (args[1], args[2], args[3], args[1] * args[3] * 4 / 1e6, bool(args[9])))
