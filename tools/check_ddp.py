"""Debug/validation (GPU box): 2+ ranks (gloo on one GPU is fine) -- after synchronised steps every rank must hold
bit-identical parameters, and the reduced gradient must equal the sum of the per-rank gradients."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch, torch.distributed as dist
import helpers as H, network, utils
from dmlnet import parallel
from dmlnet.optim import FusedSGD

rank, local, world = parallel.init_from_env()
dev = torch.device("cuda", local)
m = network.deeplabv3plus_embedding_resnet101(16, 16, False)
m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=1))
m.to(dev).train(); m.set_compute_dtype(torch.float32); m.classifier.aspp.project[3].eval()
st = m._engine.store
st.bind(dev)
opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 1e-4}, {"params": m.classifier.parameters(), "lr": 1e-3}],
               lr=1e-3, momentum=0.9, weight_decay=1e-4).bind(m)
crit = utils.DMLLoss(alpha=0.01, ignore_index=255, sync=True)
img = H.synth_tensor(100 + rank, "ddp.img", (2, 3, 64, 64)).to(dev)
lab = H.synth_labels(100 + rank, "ddp.lab", (2, 64, 64), 16, 255, ignore_rows=2).to(dev)
# reference: local (un-reduced) gradient of this rank
lg, _, ft = m(img); loss = crit(lg, lab, ft); loss.backward(); torch.cuda.synchronize()
g_local = st.flat_g.clone()
gs = [torch.zeros_like(g_local) for _ in range(world)]
dist.all_gather(gs, g_local)
g_sum = sum(gs)
# now with the overlapped reducer
m._engine.reducer = parallel.GradReducer(st, bucket_mb=8.0, average=False)
opt.zero_grad()
lg, _, ft = m(img); loss2 = crit(lg, lab, ft); loss2.backward(); torch.cuda.synchronize()
err = (st.flat_g - g_sum).abs().max().item() / g_sum.abs().max().item()
opt.step(); torch.cuda.synchronize()
ps = [torch.zeros_like(st.flat_p) for _ in range(world)]
dist.all_gather(ps, st.flat_p)
same = all(torch.equal(ps[0], p) for p in ps)
if rank == 0:
    print("loss %.6f (both passes %.6f), reduced-grad rel err vs sum of local grads %.2e, params identical across ranks: %s, buckets %d"
          % (loss.item(), loss2.item(), err, same, len(m._engine.reducer.buckets)))
    assert err < 1e-5 and same
dist.barrier(); dist.destroy_process_group()
