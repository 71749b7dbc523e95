"""Batch-sharded data parallelism: one process per GPU, gradients reduced with RCCL all-reduce over xGMI.

Replaces the single-process nn.DataParallel of the reference (main_embedding.py:425,438: scatter /
replicate / gather / reduce-add to GPU 0 every step) by a bucketed all-reduce of the flat fp32 gradient
buffer, launched on a side stream as soon as the backward plan has finished the last weight gradient of a
bucket, so the ~235 MB exchange overlaps the remaining backward kernels.  BatchNorm statistics stay per
rank, as in the reference's DataParallel replicas.
"""
from __future__ import annotations

import os
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of a global batch for `rank` (remainder spread over the first ranks)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def make_buckets(offsets: List[int], sizes: List[int], total: int, bucket_elems: int) -> List[Tuple[int, int, List[int]]]:
    """Partition [0,total) into contiguous buckets walking parameters from the END of the flat buffer (their
    gradients are produced first).  Returns (lo, hi, param indices) per bucket, in launch order."""
    buckets, hi, members = [], total, []
    for i in range(len(offsets) - 1, -1, -1):
        members.append(i)
        if hi - offsets[i] >= bucket_elems or i == 0:
            buckets.append((offsets[i], hi, members))
            hi, members = offsets[i], []
    return buckets


class GradReducer:
    def __init__(self, store, bucket_mb: float = 32.0, group=None, average: bool = True):
        self.store, self.group, self.average = store, group, average
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        # DML_FORCE_DIST=1: run the collectives with a single rank too -- the only way to execute the RCCL path (communicator
        # setup, comm-stream ordering, the nccl backend's in-place all-reduce on slices of the flat buffer) on a 1-GPU box
        self.active = self.world > 1 or (dist.is_initialized() and os.environ.get("DML_FORCE_DIST") == "1")
        sizes = [p.numel() for p in store.params]
        self.buckets = make_buckets(store.offsets, sizes, store.total, int(bucket_mb * (1 << 20) / 4))
        self._sched = {}
        self.comm_stream = None

    # --- synchronous path (CPU/gloo tests, or no overlap wanted)
    def reduce_all(self, flat_g: Optional[torch.Tensor] = None):
        flat_g = self.store.flat_g if flat_g is None else flat_g
        if not self.active:
            return
        for lo, hi, _ in self.buckets:
            seg = flat_g[lo:hi]
            dist.all_reduce(seg, group=self.group)
            if self.average:
                seg.div_(self.world)

    # --- overlapped path used by Engine.backward
    def _schedule(self, plan):
        key = id(plan)
        if key not in self._sched:
            sched = {}
            for b, (lo, hi, members) in enumerate(self.buckets):
                ready = max(plan.param_last_op.get(i, -1) for i in members)
                ready = len(plan.bwd) - 1 if ready < 0 else ready
                sched.setdefault(ready, []).append(b)
            self._sched[key] = sched
        return self._sched[key]

    def run_backward(self, plan, stream):
        if not self.active:
            plan.run_backward()
            return
        dev = self.store.flat_g.device
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=dev)
        sched = self._schedule(plan)
        cur = torch.cuda.current_stream(dev)
        flat_g = self.store.flat_g

        def hook(i, ev_main=None, ev_side=None):
            # the communication stream picks up the backward at bucket point i: through the event pair the native replay recorded
            # there (deferred: the whole backward has been enqueued by now), else through the streams' present tails
            if ev_main is not None:
                self.comm_stream.wait_event(ev_main)
                if ev_side is not None:
                    self.comm_stream.wait_event(ev_side)                       # weight gradients run on the side stream
            else:
                self.comm_stream.wait_stream(cur)
                if plan.e.overlap_wgrad:
                    self.comm_stream.wait_stream(plan.e.side_stream(dev))
            with torch.cuda.stream(self.comm_stream):
                for b in sched.get(i, ()):
                    lo, hi, _ = self.buckets[b]
                    seg = flat_g[lo:hi]
                    dist.all_reduce(seg, group=self.group)
                    if self.average:
                        seg.div_(self.world)

        hook.points = set(sched.keys())      # the bucket points: op indices after which a bucket's gradients are complete
        # one return to Python per backward instead of one per bucket (dml_plan_run_marks).  DML_REDUCER_DEFERRED=0: the hook runs
        # at each point, between two native segments (A/B; the Python replay always does)
        hook.deferred = os.environ.get("DML_REDUCER_DEFERRED", "1") != "0"
        plan.run_backward(hook=hook)
        cur.wait_stream(self.comm_stream)


def init_from_env(backend: Optional[str] = None):
    """torchrun-style bootstrap: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if (world > 1 or os.environ.get("DML_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = os.environ.get("DML_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if torch.cuda.is_available():
            local = local % max(1, torch.cuda.device_count())     # (gloo debugging: several ranks on one GPU)
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if torch.cuda.is_available():
        local = local % max(1, torch.cuda.device_count())
    return rank, local, world
