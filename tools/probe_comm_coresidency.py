"""GPU box (one GPU): does a communication kernel co-run with the f16x2 backward?  

No second GPU exists on this pool, so RCCL's all-reduce cannot run; what CAN be measured on one GPU is the question DESIGN.md
section 6 rests on: the backward's convolution kernels are persistent workgroups that hold 150 of a CU's 160 KB of LDS and two
256-register waves on three of its four SIMDs -- does a kernel with an all-reduce kernel's LOCAL shape (tens of 512-thread
workgroups, tens of KB of LDS, streaming a 32 MB bucket in place) start and progress beside them, and what does it do to the step?

The probe drives bench.py's train step (16 x 768 x 768, f16x2, fused loss backward, SGD) through the product's GradReducer with
its collective replaced by the stand-in (tools/probe_comm_standin.hip) on the reducer's comm stream at the reducer's own bucket
launch points, and reports per bucket: ready -> start delay, duration beside the backward, duration on an idle GPU; and the step
time without / with the stand-ins (+ with the stand-ins run serially after the backward).

    python3 tools/probe_comm_coresidency.py [nblocks=32] [lds_kb=32] [passes=2]
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "open-world-semantic-segmentation_amd"), os.path.join(ROOT, "tests")]
import torch  # noqa: E402

import network  # noqa: E402
import utils  # noqa: E402
from dmlnet import parallel  # noqa: E402
from dmlnet.optim import FusedSGD  # noqa: E402

arg = dict(a.split("=") for a in sys.argv[1:])
NBLK, LDS, PASSES = int(arg.get("nblocks", 32)), int(arg.get("lds_kb", 32)) * 1024, int(arg.get("passes", 2))
lib = C.CDLL(os.path.join(ROOT, "tools", "build", "libcomm_standin.so"))
lib.standin_launch.argtypes = [C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)


class ProbeReducer(parallel.GradReducer):
    """the product's reducer (buckets, launch points, stream waits) with the collective replaced by the stand-in kernel"""

    def __init__(self, store, mode):
        super().__init__(store, bucket_mb=32.0, average=False)
        self.active, self.mode = True, mode          # mode: "overlap" (as the product), "serial" (all buckets after the backward)
        self.events = []

    def run_backward(self, plan, stream):
        if self.comm_stream is None:
            self.comm_stream = torch.cuda.Stream(device=dev)
        sched = self._schedule(plan)
        cur = torch.cuda.current_stream(dev)
        flat_g = self.store.flat_g
        self.events = []

        def launch(b):
            lo, hi, _ = self.buckets[b]
            e_ready, e0, e1 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
            e_ready.record(cur)
            self.comm_stream.wait_stream(cur)
            if plan.e.overlap_wgrad:
                self.comm_stream.wait_stream(plan.e.side_stream(dev))
            with torch.cuda.stream(self.comm_stream):
                e0.record(self.comm_stream)
                seg = flat_g[lo:hi]
                lib.standin_launch(seg.data_ptr(), seg.numel() * 4 // 16 * 16, NBLK, LDS, PASSES, self.comm_stream.cuda_stream)
                e1.record(self.comm_stream)
            self.events.append((b, (hi - lo) * 4, e_ready, e0, e1))

        if self.mode == "overlap":
            def hook(i):
                for b in sched.get(i, ()):
                    launch(b)
            hook.points = set(sched.keys())
            plan.run_backward(hook=hook)
        else:
            plan.run_backward()
            for b in range(len(self.buckets)):
                launch(b)
        cur.wait_stream(self.comm_stream)


def build(mode):
    torch.manual_seed(1)
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False).to(dev)
    m.set_compute_dtype(torch.float32, fp32_products="f16x2")
    m.train()
    utils.set_bn_momentum(m.backbone, momentum=0.01)
    opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.001}, {"params": m.classifier.parameters(), "lr": 0.01}],
                   lr=0.01, momentum=0.9, weight_decay=1e-4).bind(m)
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255, fused_backward=True)
    red = None
    if mode != "none":
        m._engine.store.bind(dev)
        red = ProbeReducer(m._engine.store, mode)
        m._engine.reducer = red
    return m, opt, crit, red


g = torch.Generator().manual_seed(1234)
img = torch.randn(16, 3, 768, 768, generator=g).to(dev)
lab = torch.randint(0, 16, (16, 768, 768), generator=g)
lab[:, :38] = 255
lab = lab.to(dev)

# the stand-in alone on an idle GPU, per bucket size
buf = torch.zeros(64 << 20, device=dev)
idle = {}
for mb in (16, 24, 32, 36, 40):
    n = mb << 20
    for _ in range(3):
        lib.standin_launch(buf.data_ptr(), n, NBLK, LDS, PASSES, torch.cuda.current_stream().cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        lib.standin_launch(buf.data_ptr(), n, NBLK, LDS, PASSES, torch.cuda.current_stream().cuda_stream)
    e1.record()
    torch.cuda.synchronize()
    idle[mb] = e0.elapsed_time(e1) / 10
print("stand-in: %d workgroups x 512 threads, %d KB LDS, %d in-place passes; idle GPU: %s"
      % (NBLK, LDS // 1024, PASSES, ", ".join("%d MB %.3f ms (%.0f GB/s r+w)" % (k, v, 2 * PASSES * (k << 20) / v / 1e6) for k, v in idle.items())))
del buf

res = {}
for mode in ("none", "overlap", "serial"):
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    m, opt, crit, red = build(mode)

    def step():
        opt.zero_grad()
        lg, _, ft = m(img)
        loss = crit(lg, lab, ft)
        loss.backward()
        opt.step()
        return loss

    for _ in range(4):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 12
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / K * 1e3
    res[mode] = ms
    print("step, stand-in %-8s: %.3f ms (%.1f images/s)" % (mode, ms, 16e3 / ms))
    if red is not None:
        step()
        torch.cuda.synchronize()
        tot = 0.0
        for b, nbytes, er, e0, e1 in red.events:
            d = e0.elapsed_time(e1)
            tot += d
            ref_ms = idle[min(idle, key=lambda k: abs((k << 20) - nbytes))] * nbytes / (min(idle, key=lambda k: abs((k << 20) - nbytes)) << 20)
            print("   bucket %d (%5.1f MB): ready -> start %.3f ms, duration %.3f ms (idle GPU ~%.3f ms, x %.2f)"
                  % (b, nbytes / 1048576, er.elapsed_time(e0), d, ref_ms, d / ref_ms))
        print("   sum of stand-in durations %.3f ms" % tot)
    del m, opt, crit, red
print("overlap costs the step %.3f ms; run serially after the backward the same kernels cost %.3f ms"
      % (res["overlap"] - res["none"], res["serial"] - res["none"]))
