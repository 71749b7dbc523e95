#!/usr/bin/env python3
"""Headline benchmark: images/sec of one DMLNet train step (DeepLabV3+/ResNet-101 OS16 + pixel-prototype
distance head + DML loss + SGD), synthetic 768x768 crops, 16 images per GPU (BASELINE.json configs[2];
configs[3] when launched on 8 ranks).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--dtype bf16|f32] [--batch 16] [--size 768]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0.  `value` = global images / (max-over-ranks wall time of exactly K steps),
inputs resident in HBM.  `roofline` = the dominant kernel class (implicit-GEMM convolution on MFMA):
algorithmic conv FLOPs of one step / summed duration of its conv launches, measured with HIP events on the
launch stream in a separate profiled pass right after the timed region.  `cpu_baseline` = the CPU oracle
(a PyTorch restatement of the reference, kind "port") timed on this box's host cores on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "open-world-semantic-segmentation_amd")
for p in (ROOT, PKG, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK = {"bf16": 2.5e15, "f32": 157.3e12}      # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK = 8.0e12


def synth_batch(batch, size, rank, device):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img = torch.randn(batch, 3, size, size, generator=g)
    lab = torch.randint(0, 16, (batch, size, size), generator=g)
    lab[:, : max(1, size * 38 // 768)] = 255            # ~5 % ignored pixels
    return img.to(device), lab.to(device)


def conv_flops_of_plan(plan):
    """Algorithmic FLOPs (2*MAC on the un-padded channel counts) of every conv launch of one train step."""
    from dmlnet._lib import ConvDesc, WgradDesc
    import ctypes as C
    lib = plan.lib
    total = 0.0
    per_op = {}
    for name, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        for i, (fn, args) in enumerate(ops):
            if fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad:
                d = args[0]._obj
                # undo the padding of the stem (3->8) and of the decoder concat (304->320)
                if fn is lib.dml_conv_igemm and d.mode == 1:
                    # data gradient: same MACs as the forward conv = dY pixels x Cout x taps x Cin
                    cin = 304 if d.N == 320 else d.N
                    fl = 2.0 * d.B * d.Hi * d.Wi * d.C * d.R * d.S * cin
                else:
                    cin = 3 if d.C == 8 else (304 if d.C == 320 else d.C)
                    fl = 2.0 * d.B * d.Ho * d.Wo * d.N * d.R * d.S * cin
                per_op[(name, i)] = fl
                total += fl
    return total, per_op


def profile_convs(model, engine, step_fn, n_steps):
    """Re-run `n_steps` steps with a HIP event pair around every conv launch (same stream)."""
    from dmlnet import engine as E
    lib = engine.lib
    records = []
    orig_run = E.Plan.run

    def timed_run(ops, stream, start=0, stop=None, hook=None, skipped=()):
        stop = len(ops) if stop is None else stop
        for i in range(start, stop):
            if i in skipped:
                if hook is not None:
                    hook(i)
                continue
            fn, args = ops[i]
            is_conv = fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad
            if is_conv:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn(*args, stream)
            if rc:
                raise RuntimeError("kernel failed rc=%d" % rc)
            if is_conv:
                e1.record()
                records.append((id(ops), i, "wgrad" if fn is lib.dml_conv_wgrad else "igemm", e0, e1))
            if hook is not None:
                hook(i)

    E.Plan.run = staticmethod(timed_run)
    try:
        for _ in range(n_steps):
            step_fn()
        torch.cuda.synchronize()
    finally:
        E.Plan.run = orig_run
    t = {"igemm": 0.0, "wgrad": 0.0}
    cnt = {"igemm": 0, "wgrad": 0}
    per_op = {}
    for oid, i, kind, e0, e1 in records:
        dt = e0.elapsed_time(e1) * 1e-3
        t[kind] += dt
        cnt[kind] += 1
        per_op[(oid, i)] = per_op.get((oid, i), 0.0) + dt / n_steps
    profile_convs.per_op = per_op
    return {k: t[k] / n_steps for k in t}, {k: cnt[k] // n_steps for k in cnt}


def dump_conv_table(plan, path):
    """Per-launch table (shape, ms, TFLOP/s) of the profiled pass, for kernel tuning."""
    lib = plan.lib
    _, fl = conv_flops_of_plan(plan)
    rows = []
    for name, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        for i, (fn, args) in enumerate(ops):
            if fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad:
                d = args[0]._obj
                sec = profile_convs.per_op.get((id(ops), i), 0.0)
                kind = "wgrad" if fn is lib.dml_conv_wgrad else ("dgrad" if d.mode == 1 else "fwd")
                rows.append(dict(kind=kind, B=d.B, Hi=d.Hi, Wi=d.Wi, C=d.C, Ho=d.Ho, Wo=d.Wo, N=d.N, R=d.R,
                                 stride=d.stride, dil=d.dil, ms=sec * 1e3,
                                 tflops=fl[(name, i)] / sec / 1e12 if sec > 0 else 0.0, gflop=fl[(name, i)] / 1e9))
    with open(path, "w") as f:
        json.dump(rows, f)


def bench_distance_kernel(batch, size, device):
    """Standalone pixel->prototype distance kernel (192 B/px algorithmic: 64 read + 64 logits + 64 features)."""
    from dmlnet import _lib
    lib = _lib.load()
    x = torch.randn(batch, 16, size, size, device=device)
    protos = 3.0 * torch.eye(16, device=device)
    lg = torch.empty_like(x)
    ft = torch.empty(batch, size, size, 16, device=device)
    st = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), protos.data_ptr(), lg.data_ptr(), ft.data_ptr(), None, None, batch, 16, 16, size, size, st)
    for _ in range(3):
        lib.dml_proto_dist_fwd(*args)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 20
    e0.record()
    for _ in range(n):
        lib.dml_proto_dist_fwd(*args)
    e1.record()
    torch.cuda.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / n
    bytes_ = 192.0 * batch * size * size
    return {"kernel": "proto_dist_fwd", "bound": "hbm", "achieved": bytes_ / sec / 1e9, "peak": HBM_PEAK / 1e9,
            "unit": "GB/s", "frac": bytes_ / sec / HBM_PEAK, "bytes_per_px": 192, "ms": sec * 1e3}


def bench_input_pipeline(batch, size, device):
    """Device input pipeline: 1024 x 2048 uint8 frames -> size x size crops with colour jitter and
    flip -> normalised fp32 NCHW + int64 labels.  Algorithmic bytes per output pixel: 3 (contrast-sum pass) + 3 + 1
    read, 12 + 8 written = 27."""
    import random
    import utils
    et = utils.ext_transforms
    g = torch.Generator().manual_seed(7)
    frames = torch.randint(0, 256, (batch, 1024, 2048, 3), generator=g, dtype=torch.uint8).to(device)
    labels = torch.randint(0, 19, (batch, 1024, 2048), generator=g, dtype=torch.uint8).to(device)
    tf = et.ExtCompose([et.ExtRandomCrop(size=(size, size)), et.ExtColorJitter(brightness=0.5, contrast=0.5, saturation=0.5),
                        et.ExtRandomHorizontalFlip(), et.ExtToTensor(),
                        et.ExtNormalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
    random.seed(0)
    for _ in range(3):
        tf(frames, labels)
    torch.cuda.synchronize()
    n = 20
    t0 = time.perf_counter()
    for _ in range(n):
        tf(frames, labels)
    torch.cuda.synchronize()
    sec = (time.perf_counter() - t0) / n           # host parameter draw + H2D of 40 B/sample + both kernels
    # the two kernels alone (HIP events on the launch stream, fixed parameters)
    from dmlnet import _lib
    lib = _lib.load()
    params = tf.last_params
    arr = (_lib.AugSample * batch)()
    for b, p in enumerate(params):
        arr[b].i, arr[b].j, arr[b].flip, arr[b].n_ops = p["i"], p["j"], int(p["flip"]), len(p["ops"])
        for k, (code, f) in enumerate(p["ops"]):
            arr[b].op[k], arr[b].factor[k] = code, f
    dp = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    lsum = torch.empty(batch, dtype=torch.int32, device=device)
    oi = torch.empty(batch, 3, size, size, device=device)
    ol = torch.empty(batch, size, size, dtype=torch.int64, device=device)
    st = torch.cuda.current_stream().cuda_stream
    def kern():
        lib.dml_aug_contrast_sum(frames.data_ptr(), dp.data_ptr(), lsum.data_ptr(), batch, 1024, 2048, size, size, st)
        lib.dml_aug_apply(frames.data_ptr(), labels.data_ptr(), dp.data_ptr(), lsum.data_ptr(), oi.data_ptr(), ol.data_ptr(),
                          batch, 1024, 2048, size, size, 0.485, 0.456, 0.406, 0.229, 0.224, 0.225, st)
    for _ in range(3):
        kern()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        kern()
    e1.record()
    torch.cuda.synchronize()
    ksec = e0.elapsed_time(e1) * 1e-3 / n
    bytes_ = 27.0 * batch * size * size
    return {"kernels": "aug_contrast_sum + aug_apply", "bound": "hbm", "achieved": bytes_ / ksec / 1e9,
            "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_ / ksec / HBM_PEAK, "bytes_per_px": 27,
            "kernel_ms_per_batch": ksec * 1e3, "ms_per_batch_with_host": sec * 1e3, "images_per_sec": batch / sec}


def cpu_input_pipeline(size):
    """the oracle's restatement of the reference pipeline (numpy, one core) on two frames"""
    import random
    import numpy as np
    from oracle import transforms_ref as TR
    rs = np.random.RandomState(7)
    img = (rs.rand(1024, 2048, 3) * 256).astype(np.uint8)
    lbl = (rs.rand(1024, 2048) * 19).astype(np.uint8)
    rng = random.Random(0)
    t0 = time.perf_counter()
    n = 2
    for _ in range(n):
        p = TR.sample_params(rng, 1024, 2048, (size, size))
        TR.apply(img, lbl, p, (size, size), [0.485, 0.456, 0.406], [0.229, 0.224, 0.225])
    return {"value": n / (time.perf_counter() - t0), "unit": "images/sec", "cores": 1, "kind": "port",
            "sample": "2 frames 1024x2048 -> %dx%d, numpy restatement of the Pillow arithmetic" % (size, size)}


def cpu_baseline(size, threads):
    """The CPU oracle (port of the reference's PyTorch path) on the host cores: 768x768 bs=2 train step."""
    import helpers as H
    from oracle import dmlnet_ref as O
    torch.set_num_threads(threads)
    bs = 2
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.train()
    O.set_bn_momentum(o.backbone, 0.01)
    opt = O.make_optimizer(o, lr=0.01, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(bs, 3, size, size, generator=g)
    lab = torch.randint(0, 16, (bs, size, size), generator=g)
    lab[:, :38] = 255
    times = []
    for it in range(3):
        t0 = time.perf_counter()
        O.train_step(o, opt, img, lab, it, 100, [0.001, 0.01], lambda a, b: O.dml_loss(a, b, 0.01, 255))
        times.append(time.perf_counter() - t0)
    sec = min(times[1:])
    return {"value": bs / sec, "unit": "images/sec", "cores": threads, "kind": "port",
            "sample": "oracle (PyTorch CPU fp32 restatement of the reference) train step, %dx%d bs=%d, "
                      "best of 2 after 1 warm-up, %.2f s/step" % (size, size, bs, sec)}


def infer_bench(args):
    """Config #5 (not the headline line): eval-mode forward + argmax / max-softmax + dissum score + novel-prototype
    relabel (test_embedding.py:328-350,428-445 of the reference), images sharded over ranks, no collective."""
    from dmlnet import parallel
    rank, local, world = parallel.init_from_env()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    import network
    import utils
    import numpy as np
    torch.manual_seed(1)
    model = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    model.to(device).eval()
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    batch = args.batch if args.batch != 16 else 1
    g = torch.Generator().manual_seed(4321 + rank)
    img = torch.randn(batch, 3, args.height, args.width, generator=g).to(device)
    proto = np.full((16,), 0.1)

    def step():
        with torch.no_grad():
            logits, centers, feats = model(img)
            preds, msp = utils.argmax_msp(logits)
            score = utils.dissum_score(logits, clip=1000.0, inclusive=False)
            return utils.novel_relabel(preds, logits, feats, proto, -1.5, 16), score

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds, score = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    if rank == 0:
        print(json.dumps({
            "metric": "images/sec open-world inference (eval forward + argmax/msp + dissum + novel relabel), DeepLabV3+R101",
            "value": batch * world * args.steps / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "open-world inference %dx%d, %d image(s)/GPU/step, 16 prototypes, random-init weights"
                                   % (args.height, args.width, batch), "parallelism": "dp%d" % world}}))
    if world > 1:
        dist.destroy_process_group()


def ood_bench(args):
    """SURVEY 8(f) rank 2 (not the headline line): the open-set evaluation of anomaly/eval_ood_traditional.py:190-305 for
    one StreetHazards-sized frame -- five resized copies (short side 300..600, long side <= 1000, padded to multiples of
    8: config imgSizes / imgMaxSize / padding_constant) through ResNet-50-dilated + pyramid-pooling embedding decoder, scores
    averaged at 720x1280 inside the upsample kernel, argmax, dissum score (clip 400), AUROC / AUPR / FPR95 on the device."""
    torch.cuda.set_device(0)
    device = torch.device("cuda", 0)
    import anom_utils
    import models
    import utils
    torch.manual_seed(1)
    enc = models.ModelBuilder.build_encoder("resnet50dilated", fc_dim=2048)
    dec = models.ModelBuilder.build_decoder("ppm_deepsup_embedding", fc_dim=2048, num_class=13, use_softmax=True)
    m = models.SegmentationModuleOOD(enc, dec, None).to(device).eval()
    m.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    Hs, Ws = 720, 1280
    g = torch.Generator().manual_seed(99)
    imgs = []
    for short in (300, 375, 450, 525, 600):
        sc = min(short / float(min(Hs, Ws)), 1000.0 / float(max(Hs, Ws)))
        h, w = (int(Hs * sc) + 7) // 8 * 8, (int(Ws * sc) + 7) // 8 * 8
        imgs.append(torch.randn(1, 3, h, w, generator=g).to(device))
    label = torch.randint(0, 14, (Hs, Ws), generator=g).to(device)          # 13 = the anomaly class

    def step():
        scores, ft = models.evaluate_multiscale(m, imgs, (Hs, Ws))
        preds, _ = utils.argmax_msp(scores)
        conf = utils.dissum_score(scores, clip=400.0, inclusive=True)
        return preds, anom_utils.eval_ood_measure(conf[0], label, (13,))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds, res = step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    print(json.dumps({
        "metric": "images/sec open-set evaluation (5-scale ResNet-50-dilated + PPM embedding decoder, dissum, AUROC/AUPR/FPR95)",
        "value": args.steps / elapsed, "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": "one 720x1280 frame, inputs %s, segSize 720x1280, 13 prototypes, random-init weights"
                               % ", ".join("%dx%d" % (t.shape[2], t.shape[3]) for t in imgs), "parallelism": "dp1"}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32"])
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--dump-conv", default=None, help="write a per-launch conv table (json) from the profiled pass")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "ood"],
                    help="train = the headline metric (default); infer = SURVEY 8(d) config #5: open-world inference "
                         "at --height x --width, --batch images per step (default 1), scores on the device")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    args = ap.parse_args()
    if args.mode == "ood":
        return ood_bench(args)
    if args.mode == "infer":
        return infer_bench(args)

    from dmlnet import parallel
    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    import network
    import utils
    from dmlnet.optim import FusedSGD

    torch.manual_seed(1)
    model = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    model.to(device)
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32)
    model.train()
    utils.set_bn_momentum(model.backbone, momentum=0.01)                     # main_embedding.py:379
    lr = 0.01
    opt = FusedSGD([{"params": model.backbone.parameters(), "lr": 0.1 * lr},
                    {"params": model.classifier.parameters(), "lr": lr}],
                   lr=lr, momentum=0.9, weight_decay=1e-4).bind(model)       # main_embedding.py:385-388
    sched = utils.PolyLR(opt, 30000, power=0.9)
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255, sync=True if world > 1 else None)
    if world > 1:
        model._engine.store.bind(device)
        model._engine.reducer = parallel.GradReducer(model._engine.store, bucket_mb=32.0, average=False)
    img, lab = synth_batch(args.batch, args.size, rank, device)

    def step():
        opt.zero_grad()
        logits, centers, feats = model(img)
        loss = crit(logits, lab, feats)
        loss.backward()
        opt.step()
        sched.step()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss.item())

    out = None
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = args.batch * world * args.steps / elapsed
        out = {"metric": "images/sec train-step, DeepLabV3+R101 768x768 bs=16; % HBM & MFMA roofline",
               "value": value, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": ms, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "DMLNet train step: DeepLabV3+/ResNet-101 OS16 fwd+bwd, prototype-distance "
                                      "head, DML loss (DCE+VL), SGD; %dx%d crops, %d images/GPU, 16 prototypes, "
                                      "random-init weights" % (args.size, args.size, args.batch),
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "final_loss": final_loss}}
    if world == 1 and not args.no_profile:
        plan = next(p for k, p in model._engine.plans.items() if k[4])
        flops, _ = conv_flops_of_plan(plan)
        model._engine.overlap_wgrad = False          # profiled pass: every conv launch alone on one stream
        tsec, counts = profile_convs(model, model._engine, step, 2)
        model._engine.overlap_wgrad = True
        if args.dump_conv:
            dump_conv_table(plan, args.dump_conv)
        conv_sec = tsec["igemm"] + tsec["wgrad"]
        n_launch = counts["igemm"] + counts["wgrad"]
        peak = PEAK[args.dtype]
        # HBM bytes per conv launch from the committed PMC passes of this same command (tools/run_traffic.sh:
        # separate --pmc FETCH_SIZE / WRITE_SIZE runs, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024); null when absent
        traffic, traffic_src = None, None
        tpath = os.path.join(ROOT, "profiles", "r01_traffic_pmc.json")
        if os.path.exists(tpath) and args.batch == 16 and args.size == 768 and args.dtype == "bf16":
            with open(tpath) as fh:
                tj = json.load(fh)
            traffic = tj["conv_GB_per_step"] * 1e9 / tj["conv_launches_per_step"]
            traffic_src = "profiles/r01_traffic_pmc.json (rocprofv3 --pmc, %.1f GB per step over the conv launches)" % tj["conv_GB_per_step"]
        out["roofline"] = {"bound": "mfma", "kernel": "conv_igemm_dma_kernel / conv_igemm_kernel + conv_wgrad_big_kernel / conv_wgrad_kernel (all conv launches of a step)",
                           "achieved": flops / conv_sec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                           "frac": flops / conv_sec / peak, "traffic": traffic, "traffic_unit": "bytes per launch (mean)",
                           "traffic_source": traffic_src,
                           "flops_per_step": flops, "launches_per_step": n_launch,
                           "avg_launch_ms": conv_sec / n_launch * 1e3, "conv_ms_per_step": conv_sec * 1e3,
                           "igemm_ms_per_step": tsec["igemm"] * 1e3, "wgrad_ms_per_step": tsec["wgrad"] * 1e3,
                           "whole_step_frac": flops / (elapsed / args.steps) / peak,
                           "note": "data-gradient launches also compute the BatchNorm-backward sums of the tensor they "
                                   "write (88 of 112 bn_bwd_reduce launches folded into their epilogues); their time is "
                                   "charged to the convolutions here"}
        if traffic is not None:
            # the same launches against the OTHER roof: residual accumulates, fused BN sums and split-K partials make
            # several of the 1x1 layers bandwidth-bound (e.g. the 256->1024 data gradient moves ~250 MB in 90 us)
            out["roofline"]["hbm_frac_of_same_launches"] = traffic * n_launch / conv_sec / 8e12
        out["hbm_kernel"] = bench_distance_kernel(args.batch, args.size, device)
        dpath = os.path.join(ROOT, "profiles", "r01_traffic_dist_pmc.json")          # tools/run_traffic_dist.sh
        if os.path.exists(dpath) and args.batch == 16 and args.size == 768:
            with open(dpath) as fh:
                out["hbm_kernel"]["traffic"] = json.load(fh)["bytes_per_launch"]
            out["hbm_kernel"]["traffic_source"] = "profiles/r01_traffic_dist_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)"
        out["input_pipeline"] = bench_input_pipeline(args.batch, args.size, device)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = args.cpu_threads or min(os.cpu_count() or 8, 64)
        out["cpu_baseline"] = cpu_baseline(args.size, threads)
        if "input_pipeline" in out:
            out["input_pipeline"]["cpu_baseline"] = cpu_input_pipeline(args.size)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
