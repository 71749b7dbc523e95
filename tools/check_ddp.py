"""Data-parallel check on the GPU box (run under torch.distributed.run; gloo ranks sharing one GPU are fine, nccl = RCCL
with one GPU per rank).  Exercises the PRODUCT reducer path -- ParamStore buckets, Plan.param_last_op scheduling,
GradReducer.run_backward with its comm-stream waits, the weight-gradient side stream -- not a stand-in:

  1. overlapped bucketed reduction == sum over ranks of the un-reduced per-rank gradients (fp32 and bf16 plans);
  2. UNEVEN shards (rank r holds 3 - r % 2 ... images): loss and gradients are those of the gathered global batch, i.e.
     every rank normalises by the global image count (utils/loss.py:38-41 of the reference on DataParallel's gather);
  3. after optimizer steps every rank holds bit-identical parameters;
  4. a second backward without zero_grad() raises (in-place reduction cannot accumulate);
  5. dropout seeds differ between ranks.
Exit code 0 only if every rank passes.  Replaces nn.DataParallel, main_embedding.py:425,438 of the reference.
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "open-world-semantic-segmentation_amd")]
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import helpers as H  # noqa: E402
import network  # noqa: E402
import utils  # noqa: E402
from dmlnet import parallel  # noqa: E402
from dmlnet.optim import FusedSGD  # noqa: E402

rank, local, world = parallel.init_from_env()
assert world >= 2, "run under torch.distributed.run with >= 2 ranks"
dev = torch.device("cuda", local)
torch.cuda.set_device(dev)
ok = True


def say(msg):
    print("rank %d: %s" % (rank, msg), flush=True)


def build(dtype):
    m = network.deeplabv3plus_embedding_resnet101(16, 16, False)
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=1))
    m.to(dev).train()
    m.set_compute_dtype(dtype)
    m.classifier.aspp.project[3].eval()
    m._engine.store.bind(dev)
    return m


def rel(a, b):
    return ((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30)).item()


# global batch of 2 * world + world // 2 images, sharded unevenly by parallel.shard_range (first ranks get one more)
n_global = 2 * world + max(1, world // 2)
lo, hi = parallel.shard_range(n_global, rank, world)
img_all = H.synth_tensor(100, "ddp.img", (n_global, 3, 64, 64)).to(dev)
lab_all = H.synth_labels(100, "ddp.lab", (n_global, 64, 64), 16, 255, ignore_rows=2).to(dev)
img, lab = img_all[lo:hi], lab_all[lo:hi]
sizes = [parallel.shard_range(n_global, r, world) for r in range(world)]
assert len({b - a for a, b in sizes}) > 1, "shards are meant to be uneven"

# DDP_BUCKET_MBS: bucket sizes to run (comma list, default 8); DDP_DTYPES: f32,bf16 (default both)
BUCKET_MBS = [float(v) for v in os.environ.get("DDP_BUCKET_MBS", "8").split(",")]
DTYPES = [{"f32": torch.float32, "bf16": torch.bfloat16}[v] for v in os.environ.get("DDP_DTYPES", "f32,bf16").split(",")]
for dtype, bucket_mb in [(d, b) for d in DTYPES for b in BUCKET_MBS]:
    m = build(dtype)
    st = m._engine.store
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255, sync=True)
    # (a) no reducer: local gradient of the GLOBAL-batch loss; summed by hand over the ranks
    lg, _, ft = m(img)
    loss = crit(lg, lab, ft)
    loss.backward()
    torch.cuda.synchronize()
    g_local = st.flat_g.clone()
    gs = [torch.zeros_like(g_local) for _ in range(world)]
    dist.all_gather(gs, g_local)
    g_sum = sum(gs)
    # (b) the overlapped reducer on a fresh model (same weights)
    m2 = build(dtype)
    st2 = m2._engine.store
    m2._engine.reducer = parallel.GradReducer(st2, bucket_mb=bucket_mb, average=False)
    opt = FusedSGD([{"params": m2.backbone.parameters(), "lr": 1e-4}, {"params": m2.classifier.parameters(), "lr": 1e-3}],
                   lr=1e-3, momentum=0.9, weight_decay=1e-4).bind(m2)
    opt.zero_grad()
    lg2, _, ft2 = m2(img)
    loss2 = crit(lg2, lab, ft2)
    loss2.backward()
    torch.cuda.synchronize()
    e_red = rel(st2.flat_g, g_sum)
    plan = next(p for k, p in m2._engine.plans.items() if k[4])
    sched = m2._engine.reducer._schedule(plan)
    n_early = sum(1 for i in sched if i < len(plan.bwd) - 1)
    n_b = len(m2._engine.reducer.buckets)
    say("%s, %g MB buckets: loss %.6f, reduced gradient vs sum of local gradients %.2e, %d buckets (%d launched before the "
        "last backward op)" % (str(dtype).split(".")[-1], bucket_mb, loss2.item(), e_red, n_b, n_early))
    ok &= e_red < 1e-5 and abs(loss.item() - loss2.item()) < 1e-6 * abs(loss.item()) and n_early >= min(n_b - 1, max(1, (3 * n_b) // 4))
    # (c) global-batch semantics: one process on the whole batch with per-shard BatchNorm statistics is not expressible,
    # so compare what IS shard-independent: the loss value and d(loss)/d(logits) of this rank's images
    if dtype == torch.float32:
        ref_crit = utils.DMLLoss(alpha=0.01, ignore_index=255)
        # uneven shapes: gather through a padded buffer
        pad = torch.zeros((max(b - a for a, b in sizes), 16, 64, 64), device=dev)
        pad[: hi - lo] = lg.detach()
        pads = [torch.zeros_like(pad) for _ in range(world)]
        dist.all_gather(pads, pad)
        lg_all = torch.cat([p[: b - a] for p, (a, b) in zip(pads, sizes)]).requires_grad_(True)
        loss_ref = ref_crit(lg_all, lab_all, None)
        loss_ref.backward()
        lgl = lg.detach().clone().requires_grad_(True)
        loss_l = crit(lgl, lab, None)
        loss_l.backward()
        e_loss = abs(loss_l.item() - loss_ref.item()) / abs(loss_ref.item())
        e_dl = rel(lgl.grad, lg_all.grad[lo:hi])
        say("uneven shards %s: loss vs single-process loss on the gathered logits %.2e, d loss / d logits %.2e"
            % ([b - a for a, b in sizes], e_loss, e_dl))
        ok &= e_loss < 1e-6 and e_dl < 1e-5
    # (d) steps keep the replicas bit-identical
    for _ in range(2):
        opt.step()
        opt.zero_grad()
        lg2, _, ft2 = m2(img)
        crit(lg2, lab, ft2).backward()
    opt.step()
    torch.cuda.synchronize()
    ps = [torch.zeros_like(st2.flat_p) for _ in range(world)]
    dist.all_gather(ps, st2.flat_p)
    same = all(torch.equal(ps[0], p) for p in ps)
    say("parameters identical across ranks after 3 steps: %s" % same)
    ok &= same
    if dtype == torch.float32:
        # (e) accumulate over two backward passes: refused with a reducer attached
        lg2, _, ft2 = m2(img)
        try:
            crit(lg2, lab, ft2).backward()
            say("second backward without zero_grad() did NOT raise")
            ok = False
        except RuntimeError as e:
            ok &= "zero_grad" in str(e)
        # (f) dropout seeds
        seeds = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(seeds, torch.tensor([m2._engine.rank_seed() & 0x7FFFFFFFFFFFFFFF], dtype=torch.int64, device=dev))
        ok &= len({int(s) for s in seeds}) == world
    del m, m2, opt

flag = torch.tensor([1 if ok else 0], device=dev)
dist.all_reduce(flag, op=dist.ReduceOp.MIN)
dist.barrier()
dist.destroy_process_group()
sys.exit(0 if int(flag) == 1 else 1)
