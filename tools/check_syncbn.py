"""2-rank check of synchronised BatchNorm (run under torch.distributed.run, any backend; several ranks may share a GPU):
each rank trains on half of the batch with set_sync_batchnorm(True); logits, loss and the reduced gradients must
equal a single-process run on the whole batch.  Weights and batch are the CONDITIONED g5l fixture's (tests/tools/mint_golden_large.py:
no ReLU input of the single-process run within 64 eps32 sum|terms| of zero) -- the two-rank run merges the statistics in another
order, and on an unconditioned 64 x 64 batch one flipped ReLU of a 32-sample BatchNorm moves the gradients by percent (round 5: the
old fixture failed at 4e-3 in the two-plane mode once the residual stream was read from planes; it is a 1e-4 match here)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
import torch, torch.distributed as dist
import helpers as H, network, utils
from dmlnet import parallel

rank, local, world = parallel.init_from_env()
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
assert world == 2

G5L = H.load_golden("g5l_full_train")

def build():
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.load_state_dict(H.conditioned_state_dict(H.shapes_of(m), 1, G5L["beta_idx"], G5L["beta_val"]))
    m.to(dev); m.set_compute_dtype(torch.float32); m.train(); m.classifier.aspp.project[3].eval()
    utils.set_bn_momentum(m.backbone, 0.01)
    return m

img = H.synth_tensor(5, "g5l.img", (2, 3, 128, 128)).to(dev)
lab = H.synth_labels(5, "g5l.lab", (2, 128, 128), 16, 255, ignore_rows=5).to(dev)
per = img.shape[0] // world
# reference: one process, whole batch
ref = build()
lg_ref, _, ft_ref = ref(img)
loss_ref = utils.DMLLoss(alpha=0.01, ignore_index=255)(lg_ref, lab, ft_ref)
loss_ref.backward()
g_ref = ref._engine.store.flat_g.clone()
rm_ref = ref.backbone.layer3[5].bn2.running_var.clone()
# two ranks, half a batch each, synchronised statistics
m = build()
m.set_sync_batchnorm(True)
m._engine.store.bind(dev)
m._engine.reducer = parallel.GradReducer(m._engine.store, bucket_mb=32.0, average=False)
lo = per * rank
lg, _, ft = m(img[lo:lo + per])
loss = utils.DMLLoss(alpha=0.01, ignore_index=255, sync=True)(lg, lab[lo:lo + per], ft)
loss.backward()
torch.cuda.synchronize()
g = m._engine.store.flat_g

def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()

e_lg, e_loss, e_g = rel(lg, lg_ref[lo:lo + per]), abs(loss.item() - loss_ref.item()) / abs(loss_ref.item()), rel(g, g_ref)
e_rv = rel(m.backbone.layer3[5].bn2.running_var, rm_ref)
print("rank %d: logits %.2e loss %.2e grads %.2e running_var %.2e" % (rank, e_lg, e_loss, e_g, e_rv), flush=True)
ok = e_lg < 1e-4 and e_loss < 1e-5 and e_g < 5e-4 and e_rv < 1e-5
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if ok else 1)
