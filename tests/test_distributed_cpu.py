"""CPU, world_size 2 and 4 (uneven shards: 6 images over 4 ranks), gloo: the N>1 path -- bucketed gradient reduction over the flat buffer and the
global-batch loss normalisation -- gives the gradients of the single-process run on the concatenated batch.
The compute on each rank is plain torch (allowed in tests); the small net has no BatchNorm because the
reference's DataParallel replicas (and this design) keep BatchNorm statistics per rank."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import helpers as H


def _small_model():
    torch.manual_seed(3)
    return torch.nn.Sequential(torch.nn.Conv2d(3, 8, 3, padding=1), torch.nn.ReLU(), torch.nn.Conv2d(8, 5, 1))


class _Store:
    """Minimal stand-in for engine.ParamStore: flat gradient buffer + offsets."""
    ALIGN = 64

    def __init__(self, params):
        self.params = list(params)
        self.offsets, off = [], 0
        for p in self.params:
            self.offsets.append(off)
            off += (p.numel() + 63) // 64 * 64
        self.total = off
        self.flat_g = torch.zeros(off)


def _worker(rank, world, port, out):
    sys.path[:0] = [H.ROOT, H.PKG, os.path.join(H.ROOT, "tests")]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from dmlnet import parallel
    r, _, w = parallel.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    torch.set_num_threads(1)
    model = _small_model()
    x = H.synth_tensor(21, "ddp.x", (6, 3, 10, 12))
    y = H.synth_labels(21, "ddp.y", (6, 10, 12), 5, 255, ignore_frac=0.2)
    lo, hi = parallel.shard_range(6, rank, world)
    logits = model(x[lo:hi])
    # rank-local sums -> all-reduce -> global normalisation (what utils.DMLLoss(sync=True) does on the device)
    valid = y[lo:hi] != 255
    nll = torch.nn.functional.cross_entropy(logits, y[lo:hi], ignore_index=255, reduction="sum")
    own = logits.gather(1, y[lo:hi].clamp(0, 4).unsqueeze(1)).squeeze(1)
    var = -(own * valid).sum() / (10 * 12)
    sums = torch.tensor([float(valid.sum())], dtype=torch.float64)
    dist.all_reduce(sums)
    loss = (nll / sums[0] + 0.01 * var) / 6.0          # n = GLOBAL batch
    loss.backward()
    store = _Store(model.parameters())
    for p, off in zip(store.params, store.offsets):
        store.flat_g[off:off + p.numel()] = p.grad.flatten()
    red = parallel.GradReducer(store, bucket_mb=0.0002, average=False)
    assert len(red.buckets) >= 2
    red.reduce_all()
    if rank == 0:
        torch.save(store.flat_g, out)
    dist.barrier()
    dist.destroy_process_group()


import pytest


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_gradients_equal_single_process(tmp_path, world):
    from oracle import dmlnet_ref as O
    out = str(tmp_path / "g.pt")
    port = 29500 + (os.getpid() % 500) + world
    mp.start_processes(_worker, args=(world, port, out), nprocs=world, join=True, start_method="spawn")
    flat = torch.load(out)
    model = _small_model()
    x = H.synth_tensor(21, "ddp.x", (6, 3, 10, 12))
    y = H.synth_labels(21, "ddp.y", (6, 10, 12), 5, 255, ignore_frac=0.2)
    O.dml_loss(model(x), y, alpha=0.01, ignore_index=255).backward()
    store = _Store(model.parameters())
    for p, off in zip(store.params, store.offsets):
        assert torch.allclose(flat[off:off + p.numel()], p.grad.flatten(), rtol=1e-5, atol=1e-7)
