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

# dense MFMA peaks, MI355X_MICROARCH.md; "f32x3" = fp32 storage with the convolutions' products on the bf16 matrix cores through
# a three-term split (six bf16 MFMAs per fp32-accurate block): its ceiling in fp32-equivalent FLOPs is the bf16 peak / 6
PEAK = {"bf16": 2.5e15, "f32": 157.3e12, "f32x3": 2.5e15 / 6, "f16x2": 2.5e15 / 3}
HBM_PEAK = 8.0e12
# `dtype` of the JSON line = the arithmetic type the path's products are computed in ("f32" only for the literal fp32 MFMA mode:
# a split-product mode is named as such); config.arithmetic spells the mode out, `value_fp32_exact` carries the exact mode's rate
DTYPE_FIELD = {"f16x2": "f16x2", "f32x3": "bf16x3", "f32": "f32", "bf16": "bf16"}
ARITHMETIC = {
    "f16x2": "fp32 tensors, fp32 accumulation; convolution products on the fp16 matrix cores through a two-term split of the "
             "power-of-two-scaled operands (22 significand bits per element, three MFMAs per block: NOT the reference's literal fp32 "
             "products).  Gates it passes, same bars as the exact-fp32 mode, no mode-dependent branch in tests/: the reference-minted "
             "conditioned fixtures g5l / g8l / g12l (2x3x128x128, every ReLU input proved >= 64 eps32 sum|terms| from zero: logits, loss, "
             "running statistics and all gradient checksums at 1e-3; the trajectory at 8x the reference's own run-to-run spread), 2x768x768 vs the oracle "
             "(test_fp32_768_bs2_against_oracle[f16x2]), 1024x2048 forked eval plan + scores vs the oracle "
             "(test_config5_full_size_inference_and_scores_against_oracle[f16x2]); 16x768x768 against the exact-fp32 step of this "
             "library.  The unconditioned 64x64 fixtures g5 / g8 / g12 gate the exact-fp32 mode only; f16x2's distance from g5 / g12 is "
             "recorded and attributed to flipped knife-edge ReLU mask bits by "
             "test_f16x2_distance_from_unconditioned_fixtures_is_relu_mask_flips",
    "f32x3": "fp32 tensors, fp32 accumulation; convolution products on the bf16 matrix cores through a three-term split (six MFMAs "
             "per block); same gates as f16x2",
    "f32": "exact fp32 MFMA (v_mfma_f32_16x16x4_f32): the reference's arithmetic",
    "bf16": "bf16 storage of activations / compute weights, fp32 accumulation and statistics (statistically gated, not a 1e-3 mode)"}
FP32_PRODUCTS = {"f32": "exact", "f32x3": "bf16x3", "f16x2": "f16x2"}      # model.set_compute_dtype(torch.float32, fp32_products=...)
ROUND = "r06"


def csrc_sha():
    """Fingerprint of the kernel sources: a committed PMC traffic file is only quoted when it was collected on exactly
    these kernels (tools/run_traffic.sh stores the same value inside the file)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            with open(os.path.join(d, name), "rb") as fh:
                h.update(name.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def physical_cores():
    """(physical cores, logical CPUs) of this host from /proc/cpuinfo"""
    logical = os.cpu_count() or 1
    try:
        pairs, phys, core = set(), None, None
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.startswith("physical id"):
                    phys = ln.split(":")[1].strip()
                elif ln.startswith("core id"):
                    core = ln.split(":")[1].strip()
                elif not ln.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
        if pairs:
            return min(len(pairs), logical), logical
    except OSError:
        pass
    return logical, logical


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes of this script (one per GPU, RCCL between
    them) BEFORE anything here touches a GPU, relay their output, exit non-zero if any of them fails."""
    import socket
    import subprocess
    shared = os.environ.get("DML_BENCH_ALLOW_SHARED_GPU") == "1"        # test hook: gloo ranks on one device
    have = torch.cuda.device_count()                                      # does not initialise the GPU
    if have < n and not shared:
        sys.stderr.write("bench.py: --gpus %d but this node exposes %d GPU(s); refusing to oversubscribe\n" % (n, have))
        sys.exit(2)
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    try:
        while procs:
            for pr in list(procs):
                code = pr.poll()
                if code is None:
                    continue
                procs.remove(pr)
                if code != 0 and rc == 0:
                    rc = code
                    for other in procs:              # one rank died: the others would wait in a collective forever
                        other.terminate()
            time.sleep(0.05)
    finally:
        for pr in procs:
            pr.kill()
    sys.exit(rc)


def synth_batch(batch, size, rank, device):
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    img = torch.randn(batch, 3, size, size, generator=g)
    lab = torch.randint(0, 16, (batch, size, size), generator=g)
    lab[:, : max(1, size * 38 // 768)] = 255            # ~5 % ignored pixels
    return img.to(device), lab.to(device)


def conv_bytes_of_plan(plan):
    """Algorithmic HBM bytes of the conv launches of one train step: every operand read once, every result written
    once (accumulating data gradients read their target too), storage types of the plan; weight gradients fp32."""
    return float(sum(m["alg_bytes"] for op in conv_op_list(plan) for m in op["members"]))


def conv_flops_of_plan(plan):
    """Algorithmic FLOPs (2*MAC on the un-padded channel counts) of every conv launch of one train step."""
    lib = plan.lib
    total = 0.0
    per_op = {}
    for name, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        for i, (fn, args) in enumerate(ops):
            if fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad or fn is lib.dml_conv_wgrad_group:
                descs = args.meta if fn is lib.dml_conv_wgrad_group else [args[0]._obj]      # a grouped launch: several layers
                fl = sum(_conv_flops(d, fn is lib.dml_conv_igemm) for d in descs)
                per_op[(name, i)] = fl
                total += fl
    return total, per_op


def _conv_flops(d, igemm):
    # undo the padding of the stem (3->8) and of the decoder concat (304->320)
    if igemm and d.mode == 1:
        # data gradient: same MACs as the forward conv = dY pixels x Cout x taps x Cin
        cin = 304 if d.N == 320 else d.N
        return 2.0 * d.B * d.Hi * d.Wi * d.C * d.R * d.S * cin
    if d.C == 12 and d.R == 4:          # the stem in space-to-depth form (4x4 on 12 channels): the MACs of the 7x7 on 3 channels
        return 2.0 * d.B * d.Ho * d.Wo * d.N * 49 * 3
    cin = 3 if d.C == 8 else (304 if d.C == 320 else d.C)
    return 2.0 * d.B * d.Ho * d.Wo * d.N * d.R * d.S * cin


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
            is_conv = fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad or fn is lib.dml_conv_wgrad_group
            if is_conv:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            rc = fn(*args, stream)
            if rc:
                raise RuntimeError("kernel failed rc=%d" % rc)
            if is_conv:
                e1.record()
                records.append((id(ops), i, "igemm" if fn is lib.dml_conv_igemm else "wgrad", e0, e1))
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
            if fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad or fn is lib.dml_conv_wgrad_group:
                sec = profile_convs.per_op.get((id(ops), i), 0.0)
                descs = args.meta if fn is lib.dml_conv_wgrad_group else [args[0]._obj]
                for d in descs:                      # a grouped launch: its time is shared out by FLOPs
                    f1 = _conv_flops(d, fn is lib.dml_conv_igemm)
                    s1 = sec * f1 / fl[(name, i)]
                    kind = "wgrad" if fn is not lib.dml_conv_igemm else ("dgrad" if d.mode == 1 else "fwd")
                    rows.append(dict(kind=kind, B=d.B, Hi=d.Hi, Wi=d.Wi, C=d.C, Ho=d.Ho, Wo=d.Wo, N=d.N, R=d.R,
                                     stride=d.stride, dil=d.dil, ms=s1 * 1e3, grouped=len(descs),
                                     tflops=f1 / s1 / 1e12 if s1 > 0 else 0.0, gflop=f1 / 1e9))
    with open(path, "w") as f:
        json.dump(rows, f)


def _conv_class(d, igemm):
    """(kind, label) of one conv launch; the label is the FORWARD convolution's signature for all three kinds"""
    if igemm and d.mode == 1:          # data gradient: x = dY [Ho_fwd x Wo_fwd x Cout], result = dX
        kind, cin, cout, hw = "dgrad", (304 if d.N == 320 else d.N), d.C, (d.Hi, d.Wi)
    else:
        kind = "fwd" if igemm else "wgrad"
        cin, cout, hw = (3 if d.C == 8 else (304 if d.C == 320 else d.C)), d.N, (d.Ho, d.Wo)
        if d.C == 12 and d.R == 4:      # the stem in space-to-depth form: labelled as the convolution it computes
            return kind, "7x7 3->%d @%dx%d s2" % (cout, hw[0], hw[1])
    lab = "%dx%d %d->%d @%dx%d" % (d.R, d.S, cin, cout, hw[0], hw[1])
    if igemm and d.mode == 1 and d.sub_grid:
        lab += " s2 parity class"          # one of the stride-1 launches a stride-2 data gradient is issued as (DmlConvDesc.sub_grid)
    if d.stride != 1:
        lab += " s%d" % d.stride
    if d.dil != 1:
        lab += " d%d" % d.dil
    return kind, lab


def _conv_alg_bytes(d, igemm, es):
    """algorithmic HBM bytes of one launch: every operand read once, every result written once"""
    if igemm:
        rd = d.B * d.Hi * d.Wi * d.C * es + d.N * d.R * d.S * d.C * es
        wr = d.B * d.Ho * d.Wo * d.N * (4 if d.y_f32 else es)
        return rd + wr * (2 if (d.accum or d.res_dz) else 1) + (wr * 2 if d.acc32 else 0)
    return d.B * d.Hi * d.Wi * d.C * es + d.B * d.Ho * d.Wo * d.N * es + 2 * 4 * d.N * d.R * d.S * d.C


def conv_op_list(plan):
    """the conv launches of one step in issue order (forward list, then backward list): one entry per launch with its
    member layers -- tools/pmc_by_class.py zips this with the profiler's dispatch sequence of a serial run"""
    lib = plan.lib
    out = []
    for name, ops in (("fwd", plan.fwd), ("bwd", plan.bwd)):
        for i, (fn, args) in enumerate(ops):
            if fn is lib.dml_conv_igemm or fn is lib.dml_conv_wgrad or fn is lib.dml_conv_wgrad_group:
                ig = fn is lib.dml_conv_igemm
                descs = args.meta if fn is lib.dml_conv_wgrad_group else [args[0]._obj]
                out.append({"list": name, "op": i, "members": [
                    {"kind": _conv_class(d, ig)[0], "label": _conv_class(d, ig)[1], "flops": _conv_flops(d, ig),
                     "alg_bytes": _conv_alg_bytes(d, ig, plan.es)} for d in descs]})
    return out


def conv_class_table(plan, dtype, top=12, pmc_classes=None):
    """Per shape class of the profiled pass: launches, ms per step, TFLOP/s, and the class's OWN two-roof bound
    bound_ms = max(FLOPs / MFMA peak, algorithmic bytes / 8 TB/s) with frac = bound_ms / ms (hard part 1: the
    roofline of this network has to be read per layer class -- the 1x1 layers on the large maps are streaming GEMMs).
    Returns (rows for the JSON line: the `top` classes by time + one "other" row per kind, all rows)."""
    peak = PEAK[dtype]
    rows = {}
    for op in conv_op_list(plan):
        ops = plan.fwd if op["list"] == "fwd" else plan.bwd
        sec = profile_convs.per_op.get((id(ops), op["op"]), 0.0)
        ftot = sum(m["flops"] for m in op["members"])
        for m in op["members"]:
            r = rows.setdefault((m["kind"], m["label"]), {"kind": m["kind"], "shape": m["label"], "launches": 0, "ms": 0.0,
                                                          "gflop": 0.0, "alg_MB": 0.0})
            r["launches"] += 1
            r["ms"] += sec * m["flops"] / ftot * 1e3              # a grouped launch: its time is shared out by FLOPs
            r["gflop"] += m["flops"] / 1e9
            r["alg_MB"] += m["alg_bytes"] / 1e6

    def finish(r):
        t_mfma, t_hbm = r["gflop"] * 1e9 / peak * 1e3, r["alg_MB"] * 1e6 / HBM_PEAK * 1e3
        r["bound"] = "mfma" if t_mfma >= t_hbm else "hbm"
        r["bound_ms"] = round(max(t_mfma, t_hbm), 3)
        r["frac"] = round(r["bound_ms"] / r["ms"], 3) if r["ms"] > 0 else 0.0
        r["tflops"] = round(r["gflop"] / r["ms"], 1) if r["ms"] > 0 else 0.0
        r["ms"], r["gflop"], r["alg_MB"] = round(r["ms"], 3), round(r["gflop"], 1), round(r["alg_MB"], 1)
        return r

    full = sorted((finish(dict(r)) for r in rows.values()), key=lambda r: -r["ms"])
    # PMC bytes of the class over its algorithmic bytes (profiles/*_traffic_pmc.json "classes", same kernel sources only)
    pmc = {(c["kind"], c["shape"]): c["pmc_over_alg"] for c in (pmc_classes or [])}
    for r in full:
        r["pmc_over_alg"] = pmc.get((r["kind"], r["shape"]))
    head, rest = full[:top], {}
    for r in rows.values():
        if not any(h["kind"] == r["kind"] and h["shape"] == r["shape"] for h in head):
            o = rest.setdefault(r["kind"], {"kind": r["kind"], "shape": "other classes", "launches": 0, "ms": 0.0, "gflop": 0.0,
                                            "alg_MB": 0.0})
            for k in ("launches", "ms", "gflop", "alg_MB"):
                o[k] += r[k]
    # compact rows (the driver keeps the TAIL of stdout: the line has to stay a few KB): columns = CLASS_COLS
    line = [[r["kind"], r["shape"].replace("x48", "").replace("x192", "").replace("x96", ""), r["launches"], r["ms"], r["bound"],
             r["bound_ms"], r["frac"], r["tflops"], r.get("pmc_over_alg")]
            for r in head + sorted((finish(o) for o in rest.values()), key=lambda r: -r["ms"])]
    return line, full


CLASS_COLS = ["kind", "conv (kxk Cin->Cout @map)", "launches", "ms", "own_bound", "bound_ms=max(flops/peak,alg_bytes/8TB/s)",
              "frac=bound_ms/ms", "TFLOP/s", "PMC bytes / algorithmic bytes"]


def _time_launches(fn, n=20, warm=3):
    """average duration of `fn` (one kernel launch on torch's current stream): a HIP event pair on that stream around EVERY launch,
    mean of the n durations -- what rocprofv3's kernel trace reports as the kernel's average (one pair around the whole batch measures
    the batch's throughput: consecutive launches overlap their ramp-up / drain, 5-12 % less per launch on some boxes)"""
    for _ in range(warm):
        fn()
    pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for e0, e1 in pairs:
        e0.record()
        fn()
        e1.record()
    torch.cuda.synchronize()
    return sum(e0.elapsed_time(e1) for e0, e1 in pairs) * 1e-3 / n


def bench_distance_kernel(batch, size, device):
    """The pixel->prototype distance kernels.  `proto_dist_fwd` is the standalone head BASELINE.json names (192 B/px
    algorithmic: 64 read + 64 logits + 64 features); `step_path` lists the kernels the train / inference plans actually
    launch for network/utils.py:88-118 of the reference -- the upsample-fused forward (132 B/px: 4 read at 1/16 of the
    pixels + 128 written) and the backward chain -- timed alone at the step's shapes."""
    from dmlnet import _lib
    lib = _lib.load()
    x = torch.randn(batch, 16, size, size, device=device)
    protos = 3.0 * torch.eye(16, device=device)
    lg = torch.empty_like(x)
    ft = torch.empty(batch, size, size, 16, device=device)
    st = torch.cuda.current_stream().cuda_stream
    args = (x.data_ptr(), protos.data_ptr(), lg.data_ptr(), ft.data_ptr(), None, None, batch, 16, 16, size, size, st)
    sec = _time_launches(lambda: lib.dml_proto_dist_fwd(*args))
    px = float(batch) * size * size
    out = {"kernel": "proto_dist_fwd", "bound": "hbm", "achieved": 192.0 * px / sec / 1e9, "peak": HBM_PEAK / 1e9,
           "unit": "GB/s", "frac": 192.0 * px / sec / HBM_PEAK, "bytes_per_px": 192, "ms": sec * 1e3}
    low = size // 4
    emb = torch.randn(batch, low, low, 16, device=device)
    sec_u = _time_launches(lambda: lib.dml_upsample_dist_fwd(emb.data_ptr(), protos.data_ptr(), lg.data_ptr(), ft.data_ptr(),
                                                             None, None, batch, low, low, 16, 16, size, size, st))
    step = [{"kernel": "upsample_dist_fwd (fused x4 bilinear + distances, the forward the plans run)", "bytes_per_px": 132,
             "ms": sec_u * 1e3, "achieved": 132.0 * px / sec_u / 1e9, "frac": 132.0 * px / sec_u / HBM_PEAK}]
    lab = torch.randint(0, 16, (batch, size, size), device=device)
    for entry in head_backward_entries(lib, batch, size, low, lg, ft, lab, protos, st):
        step.append(entry)
    out["step_path"] = step
    return out


def head_backward_entries(lib, batch, size, low, lg, ft, lab, protos, st):
    """the backward of the loss + distance head + final upsample as the training plan runs it"""
    from dmlnet import _lib
    px = float(batch) * size * size
    dev = lg.device
    sums = torch.zeros(5, dtype=torch.float64, device=dev)
    part = torch.empty(_lib.LOSS_BLOCKS * 4, dtype=torch.float32, device=dev)
    lib.dml_loss_fwd(lg.data_ptr(), lab.data_ptr(), sums.data_ptr(), part.data_ptr(), batch, 16, size, size, 255, st)
    gout = torch.ones((), device=dev)
    de = torch.empty(batch, low, low, 16, dtype=torch.bfloat16, device=dev)
    if hasattr(lib, "dml_head_bwd_fused"):
        sec = _time_launches(lambda: lib.dml_head_bwd_fused(ft.data_ptr(), lab.data_ptr(), sums.data_ptr(), gout.data_ptr(),
                                                            protos.data_ptr(), de.data_ptr(), batch, low, low, 16, 16, size,
                                                            size, 255, 0.01, float(batch), 1, st))
        # reads features 64 + labels 8 per pixel, writes the low-resolution embedding gradient (2 B x 16 ch / 16 px)
        return [{"kernel": "head_bwd_fused (loss grad + distance grad + bilinear grad in one pass)", "bytes_per_px": 74,
                 "ms": sec * 1e3, "achieved": 74.0 * px / sec / 1e9, "frac": 74.0 * px / sec / HBM_PEAK}]
    gl = torch.empty_like(lg)
    df = torch.empty_like(ft)
    s1 = _time_launches(lambda: lib.dml_loss_bwd(lg.data_ptr(), lab.data_ptr(), sums.data_ptr(), gout.data_ptr(), gl.data_ptr(),
                                                 batch, 16, size, size, 255, 0.01, float(batch), st))
    s2 = _time_launches(lambda: lib.dml_proto_dist_bwd(gl.data_ptr(), None, ft.data_ptr(), protos.data_ptr(), df.data_ptr(),
                                                       batch, 16, 16, size, size, st))
    s3 = _time_launches(lambda: lib.dml_bilinear_bwd(df.data_ptr(), de.data_ptr(), batch, low, low, size, size, 16, 16, 16, 1,
                                                     1, 0, st))
    return [{"kernel": "loss_bwd", "bytes_per_px": 136, "ms": s1 * 1e3, "achieved": 136.0 * px / s1 / 1e9,
             "frac": 136.0 * px / s1 / HBM_PEAK},
            {"kernel": "proto_dist_bwd", "bytes_per_px": 192, "ms": s2 * 1e3, "achieved": 192.0 * px / s2 / 1e9,
             "frac": 192.0 * px / s2 / HBM_PEAK},
            {"kernel": "bilinear_bwd (fp32 -> low-resolution bf16)", "bytes_per_px": 66, "ms": s3 * 1e3,
             "achieved": 66.0 * px / s3 / 1e9, "frac": 66.0 * px / s3 / HBM_PEAK}]


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
    """The CPU oracle (port of the reference's PyTorch path) on the host cores: 768x768 bs=2 train step on `threads`
    threads (default: every physical core of the box) and, for comparability with the 8-vCPU authoring container
    (SURVEY 8(d)), one more step on 8 threads."""
    import helpers as H  # noqa: F401
    from oracle import dmlnet_ref as O
    phys, logical = physical_cores()
    threads = threads or phys
    bs = 2
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    o.train()
    O.set_bn_momentum(o.backbone, 0.01)
    opt = O.make_optimizer(o, lr=0.01, weight_decay=1e-4)
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(bs, 3, size, size, generator=g)
    lab = torch.randint(0, 16, (bs, size, size), generator=g)
    lab[:, :38] = 255

    def one(it):
        t0 = time.perf_counter()
        O.train_step(o, opt, img, lab, it, 100, [0.001, 0.01], lambda a, b: O.dml_loss(a, b, 0.01, 255))
        return time.perf_counter() - t0

    # more threads is not faster for this size on a 2-socket host (128 threads: 14.4 s/step, 8 threads: 4.95 s on the
    # 2 x EPYC 9575F box): ONE probe step per thread count after a warm-up picks the best count, then the protocol of
    # BASELINE.md section 4 on that count -- 1 warm-up + 3 timed steps, mean quoted
    counts = [threads] if threads != phys else sorted({c for c in (8, 32, phys) if c <= logical})
    torch.set_num_threads(min(32, logical))
    one(0)                                                   # warm-up (allocator, thread pool)
    by_threads = {}
    it = 1
    for c in counts:
        torch.set_num_threads(c)
        by_threads[c] = one(it)
        it += 1
        if by_threads[c] > 12.0 and len(by_threads) > 1:      # (bounded sample: a count that is far off is not probed further)
            break
    best = min(by_threads, key=by_threads.get)
    torch.set_num_threads(best)
    timed = [one(it + k) for k in range(3)]                  # (the probe step on this count was its warm-up)
    sec = sum(timed) / len(timed)
    out = {"value": bs / sec, "unit": "images/sec", "cores": best, "kind": "port",
           "host": "%d physical cores / %d logical CPUs" % (phys, logical),
           "probe_by_threads": {str(c): bs / t for c, t in by_threads.items()},
           "timed_steps_s": [round(t, 3) for t in timed],
           "sample": "oracle (PyTorch CPU fp32 restatement of the reference) train step, %dx%d bs=%d: 3 timed steps on %d threads "
                     "(mean %.2f s/step) after a warm-up step and one probe step per thread count, whose best it is (%s)"
                     % (size, size, bs, best, sec, ", ".join("%d threads %.2f s" % (c, t) for c, t in by_threads.items()))}
    if 8 in by_threads:
        out["value_8_threads"] = bs / by_threads[8]
    return out


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
    model.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32,
                            fp32_products=None if args.dtype == "bf16" else FP32_PRODUCTS[args.dtype])
    fwd_only = args.mode == "fwd"          # BASELINE config #2: forward-only, 768 x 768, 8 images (eval plan: BatchNorm folded into the conv epilogues)
    if fwd_only:
        batch = args.batch if args.batch != 16 else 8
        args.height = args.width = args.size
    else:
        batch = args.batch if args.batch != 16 else 1
    g = torch.Generator().manual_seed(4321 + rank)
    img = torch.randn(batch, 3, args.height, args.width, generator=g).to(device)
    proto = np.full((16,), 0.1)

    def step():
        with torch.no_grad():
            logits, centers, feats = model(img)
            if fwd_only:
                return logits, feats
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
    roof = None
    if rank == 0 and world == 1 and not args.no_profile:
        # every conv launch of the (un-forked) forward plan alone on the stream, HIP events around each: algorithmic FLOPs / summed time
        plan = next(iter(model._engine.plans.values()))
        flops, _ = conv_flops_of_plan(plan)
        tsec, counts = profile_convs(model, model._engine, step, 2)
        peak = PEAK[args.dtype]
        roof = {"bound": "mfma", "kernel": "all conv launches of the forward plan", "achieved": flops / tsec["igemm"] / 1e12,
                "peak": peak / 1e12, "unit": "TFLOP/s", "frac": flops / tsec["igemm"] / peak, "traffic": None,
                "flops_per_step": flops, "launches_per_step": counts["igemm"], "conv_ms_per_step": tsec["igemm"] * 1e3,
                "whole_step_frac": flops / (elapsed / args.steps) / peak,
                "note": "profiled pass runs the ASPP branches one after the other (Python replay); the timed region forks them where the plan does"}
    if rank == 0:
        out = {
            "metric": ("images/sec forward-only (eval), DeepLabV3+R101 OS16" if fwd_only else
                       "images/sec open-world inference (eval forward + argmax/msp + dissum + novel relabel), DeepLabV3+R101"),
            "value": batch * world * args.steps / elapsed, "unit": "images/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": DTYPE_FIELD[args.dtype], "data": "synthetic",
            "config": {"workload": ("forward-only %dx%d, %d images/GPU/step, random-init weights" if fwd_only else
                                    "open-world inference %dx%d, %d image(s)/GPU/step, 16 prototypes, random-init weights")
                                   % (args.height, args.width, batch), "parallelism": "dp%d" % world,
                       "arithmetic": ARITHMETIC[args.dtype]}}
        if roof is not None:
            out["roofline"] = roof
        print(json.dumps(out))
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
    m.set_compute_dtype(torch.bfloat16 if args.dtype == "bf16" else torch.float32,
                        fp32_products=None if args.dtype == "bf16" else FP32_PRODUCTS[args.dtype])
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
        "dtype": DTYPE_FIELD[args.dtype], "data": "synthetic",
        "config": {"workload": "one 720x1280 frame, inputs %s, segSize 720x1280, 13 prototypes, random-init weights"
                               % ", ".join("%dx%d" % (t.shape[2], t.shape[3]) for t in imgs), "parallelism": "dp1",
                   "arithmetic": ARITHMETIC[args.dtype]}}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--dtype", default="f16x2", choices=["f16x2", "bf16", "f32", "f32x3"],
                    help="f16x2 (headline): fp32 storage and accumulation, the convolutions' products on the fp16 matrix cores "
                         "through a two-term split of the scaled operands -- the fastest mode that passes the reference-minted "
                         "conditioned fixtures (g5l / g8l / g12l) at the exact mode's bars; f32 = exact fp32 MFMA (the reference's arithmetic bit for bit); f32x3 = the three-term bf16 "
                         "split; bf16 = bf16 storage (the throughput mode: statistically gated, not a 1e-3 mode)")
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=768)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--no-fp32-companion", "--no-companions", dest="no_fp32_companion", action="store_true",
                    help="skip the companion passes (bf16 storage, exact fp32, three-term split) of the default run")
    ap.add_argument("--cpu-threads", type=int, default=0, help="0 = every physical core of the host")
    ap.add_argument("--dump-conv", default=None, help="write a per-launch conv table (json) from the profiled pass")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "fwd", "ood"],
                    help="train = the headline metric (default); infer = SURVEY 8(d) config #5: open-world inference "
                         "at --height x --width, --batch images per step (default 1), scores on the device; fwd = BASELINE "
                         "config #2: forward-only (eval) at --size, --batch images per step (default 8)")
    ap.add_argument("--height", type=int, default=1024)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--print-csrc-sha", action="store_true", help="print the fingerprint of the kernel sources and exit")
    args = ap.parse_args()
    if args.print_csrc_sha:
        print(csrc_sha())
        return
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and args.mode != "ood":
        launch_ranks(args.gpus)                  # never returns
    if args.mode == "ood":
        return ood_bench(args)
    if args.mode in ("infer", "fwd"):
        return infer_bench(args)

    from dmlnet import parallel
    rank, local, world = parallel.init_from_env()
    if world != args.gpus and world > 1:
        args.gpus = world
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    dp = dist.is_available() and dist.is_initialized()      # world > 1, or a single forced rank (DML_FORCE_DIST=1)
    backend = dist.get_backend() if dp else None

    res = train_pass(args, args.dtype, device, rank, world, steps=args.steps, warmup=args.warmup,
                     profile=(world == 1 and not args.no_profile), dump_conv=args.dump_conv)
    out = None
    if rank == 0:
        out = {"metric": "images/sec train-step, DeepLabV3+R101 768x768 bs=16; % HBM & MFMA roofline",
               "value": res["value"], "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": DTYPE_FIELD[args.dtype], "data": "synthetic",
               "config": {"workload": "DMLNet train step: DeepLabV3+/ResNet-101 OS16 fwd+bwd, prototype-distance "
                                      "head, DML loss (DCE+VL), SGD; %dx%d crops, %d images/GPU, 16 prototypes, "
                                      "random-init weights" % (args.size, args.size, args.batch),
                          "arithmetic": ARITHMETIC[args.dtype],
                          "global_batch": args.batch * world, "parallelism": "dp%d" % world,
                          "rccl_ranks": dist.get_world_size() if dp else 1, "backend": backend,
                          "final_loss": res["final_loss"]},
               "host_enqueue_ms_per_step": res["host_ms"]}
        for k in ("allreduce_32mb_ms", "allreduce_32mb_busbw_GBps"):
            if k in res:
                out["config"][k] = res[k]
        if "allreduce_32mb_ms" in res:
            out["config"]["allreduce_backend"] = backend      # (gloo: host copies, says nothing about the fabric)
        if "roofline" in res:
            out["roofline"] = res["roofline"]
    if world == 1 and not args.no_profile:
        out["hbm_kernel"] = bench_distance_kernel(args.batch, args.size, device)
        dpath = os.path.join(ROOT, "profiles", ROUND + "_traffic_dist_pmc.json")          # tools/run_traffic_dist.sh
        if os.path.exists(dpath) and args.batch == 16 and args.size == 768:
            try:
                with open(dpath) as fh:
                    dj = json.load(fh)
            except ValueError:
                dj = {}
            if dj.get("csrc_sha") == csrc_sha():
                out["hbm_kernel"]["traffic"] = dj["bytes_per_launch"]
                out["hbm_kernel"]["traffic_source"] = "profiles/%s_traffic_dist_pmc.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, csrc %s)" % (ROUND, dj["csrc_sha"])
        if "traffic" not in out["hbm_kernel"]:
            out["hbm_kernel"]["traffic"] = None
        out["input_pipeline"] = bench_input_pipeline(args.batch, args.size, device)
        if args.dtype == "f16x2" and not args.no_fp32_companion:
            keep = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "algorithmic_bytes",
                    "traffic_over_algorithmic", "flops_per_step", "launches_per_step", "conv_ms_per_step", "igemm_ms_per_step",
                    "wgrad_ms_per_step", "whole_step_frac", "classes")
            # bf16 storage (activations, compute weights; fp32 accumulation): the throughput mode.  NOT a 1e-3 mode: its parity is
            # gated statistically against the bf16-storage emulation of the oracle (tests/test_gpu_bf16_parity.py)
            torch.cuda.empty_cache()
            b = train_pass(args, "bf16", device, rank, world, steps=args.steps, warmup=args.warmup, profile=True,
                           dump_conv=(args.dump_conv + ".bf16") if args.dump_conv else None)
            out["bf16_companion"] = {"dtype": "bf16", "value": b["value"], "unit": "images/sec", "ms_per_step": b["ms_per_step"],
                                     "steps": args.steps, "warmup": args.warmup, "final_loss": b["final_loss"],
                                     "roofline": {k: b["roofline"][k] for k in keep if k in b["roofline"]}}
            # the reference's arithmetic bit for bit (network/utils.py:84-118 computes in fp32): exact fp32 MFMAs
            # (v_mfma_f32_16x16x4_f32, 1/16 of the 16-bit matrix rate)
            torch.cuda.empty_cache()
            f = train_pass(args, "f32", device, rank, world, steps=args.steps, warmup=args.warmup, profile=True,
                           dump_conv=(args.dump_conv + ".fp32") if args.dump_conv else None)
            out["value_fp32_exact"] = f["value"]       # the reference's literal fp32 products, same protocol (fp32_exact_companion)
            out["fp32_exact_companion"] = {"dtype": "f32", "value": f["value"], "unit": "images/sec", "ms_per_step": f["ms_per_step"],
                                           "steps": args.steps, "warmup": args.warmup, "final_loss": f["final_loss"],
                                           "roofline": {k: f["roofline"][k] for k in keep if k in f["roofline"] and k != "classes"}}
            # the three-term bf16 split (six MFMAs per block; round 3's fp32-accurate mode)
            torch.cuda.empty_cache()
            f3 = train_pass(args, "f32x3", device, rank, world, steps=max(4, args.steps // 2), warmup=2, profile=False, dump_conv=None)
            out["fp32_exact_companion"]["three_term_split"] = {"dtype": "f32x3", "value": f3["value"], "ms_per_step": f3["ms_per_step"],
                                                               "final_loss": f3["final_loss"]}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.size, args.cpu_threads)
        if "input_pipeline" in out:
            out["input_pipeline"]["cpu_baseline"] = cpu_input_pipeline(args.size)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dp:
        dist.barrier()
        dist.destroy_process_group()


def train_pass(args, dtype, device, rank, world, steps, warmup, profile, dump_conv=None):
    """W warm-up + exactly K timed train steps in `dtype` (barrier + synchronize on both sides, max over ranks), then --
    single GPU only -- the profiled conv pass.  The model and its plans are released on return."""
    from dmlnet import parallel
    import network
    import utils
    from dmlnet.optim import FusedSGD

    dp = dist.is_available() and dist.is_initialized()
    torch.manual_seed(1)
    model = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    model.to(device)
    model.set_compute_dtype(torch.bfloat16 if dtype == "bf16" else torch.float32,
                            fp32_products={"f32x3": "bf16x3", "f16x2": "f16x2"}.get(dtype, "exact"))
    model.train()
    utils.set_bn_momentum(model.backbone, momentum=0.01)                     # main_embedding.py:379
    lr = 0.01
    opt = FusedSGD([{"params": model.backbone.parameters(), "lr": 0.1 * lr},
                    {"params": model.classifier.parameters(), "lr": lr}],
                   lr=lr, momentum=0.9, weight_decay=1e-4).bind(model)       # main_embedding.py:385-388
    sched = utils.PolyLR(opt, 30000, power=0.9)
    crit = utils.DMLLoss(alpha=0.01, ignore_index=255, sync=True if dp else None,
                         fused_backward=True)      # the loss is the only consumer of the logits (main_embedding.py:466-470)
    if dp:
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

    for _ in range(warmup):
        loss = step()
    torch.cuda.synchronize()
    ar_ms = None
    if dp:
        # self-diagnosis for the first multi-GPU run (outside the timed region): one gradient bucket's worth of all-reduce on
        # the reducer's communication stream.  bus bandwidth = 2 (n - 1) / n x bytes / time: ~150-250 GB/s says RCCL runs over
        # xGMI, a few GB/s says it fell back to host memory (DESIGN.md section 6 states what the step should then look like)
        red = model._engine.reducer
        cs = getattr(red, "comm_stream", None) or torch.cuda.current_stream(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        cs.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(cs):
            buf = torch.zeros(32 * (1 << 20) // 4, dtype=torch.float32, device=device)      # allocated and filled on the stream that uses it
            for _ in range(2):
                dist.all_reduce(buf)
            e0.record(cs)
            for _ in range(5):
                dist.all_reduce(buf)
            e1.record(cs)
        torch.cuda.synchronize()
        ar_ms = e0.elapsed_time(e1) / 5
        del buf
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    if dp:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dp:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    final_loss = float(loss.item())
    # host time to ENQUEUE one step (outside the timed region): the queue is drained first, so the launches never wait
    # for the GPU and what is measured is the host's own work (Python + the native launch-list replay + HIP launches)
    host = 0.0
    for _ in range(3):
        torch.cuda.synchronize()
        h0 = time.perf_counter()
        step()
        host += time.perf_counter() - h0
    torch.cuda.synchronize()
    res = {"value": args.batch * world * steps / elapsed, "ms_per_step": elapsed / steps * 1e3,
           "final_loss": final_loss, "host_ms": host / 3 * 1e3}
    if ar_ms is not None:
        n = dist.get_world_size()
        res["allreduce_32mb_ms"] = ar_ms
        res["allreduce_32mb_busbw_GBps"] = (2.0 * (n - 1) / n * 32 * (1 << 20) / (ar_ms * 1e-3) / 1e9) if n > 1 else None
    if os.environ.get("DML_BENCH_OPLOG") and rank == 0:
        # the conv launches of a step in issue order, for tools/pmc_by_class.py (profiler runs)
        plan = next(p for k, p in model._engine.plans.items() if k[4])
        with open(os.environ["DML_BENCH_OPLOG"], "w") as fh:
            json.dump({"dtype": dtype, "steps_in_run": warmup + steps + 3, "ops": conv_op_list(plan)}, fh)
    if profile:
        plan = next(p for k, p in model._engine.plans.items() if k[4])
        flops, _ = conv_flops_of_plan(plan)
        alg_bytes = conv_bytes_of_plan(plan)
        model._engine.overlap_wgrad = False          # profiled pass: every conv launch alone on one stream
        tsec, counts = profile_convs(model, model._engine, step, 2)
        model._engine.overlap_wgrad = True
        if dump_conv:
            dump_conv_table(plan, dump_conv)
        conv_sec = tsec["igemm"] + tsec["wgrad"]
        n_launch = counts["igemm"] + counts["wgrad"]
        peak = PEAK[dtype]
        # HBM bytes per conv launch from the committed PMC passes of this same command (tools/run_traffic.sh: separate
        # --pmc FETCH_SIZE / WRITE_SIZE runs, bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024).  Quoted only when the file was
        # collected on exactly these kernel sources (csrc_sha inside the file), else null.
        traffic, traffic_src, pmc_classes = None, None, None
        tpath = os.path.join(ROOT, "profiles", ROUND + {"bf16": "_bf16_traffic_pmc.json", "f32": "_fp32_traffic_pmc.json",
                                                        "f16x2": "_traffic_pmc.json"}.get(dtype, "_none"))
        if os.path.exists(tpath) and args.batch == 16 and args.size == 768:
            try:
                with open(tpath) as fh:
                    tj = json.load(fh)
            except ValueError:              # an empty / truncated artefact (a profiler pass that failed): no traffic figure
                tj = {}
            if tj.get("csrc_sha") == csrc_sha():
                traffic = tj["conv_GB_per_step"] * 1e9 / n_launch      # per launch OF THIS PLAN (a grouped launch is one)
                pmc_classes = tj.get("classes")
                traffic_src = "profiles/%s (rocprofv3 --pmc, %.1f GB/step over the conv launches)" \
                    % (os.path.basename(tpath), tj["conv_GB_per_step"])
        roof = {"bound": "mfma", "kernel": "all conv launches of a step (conv_igemm*: fwd / dgrad, conv_wgrad*: wgrad)",
                "achieved": flops / conv_sec / 1e12, "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": flops / conv_sec / peak, "traffic": traffic, "traffic_unit": "B/launch",
                "traffic_source": traffic_src, "algorithmic_bytes": alg_bytes / n_launch,
                "flops_per_step": flops, "launches_per_step": n_launch,
                "avg_launch_ms": conv_sec / n_launch * 1e3, "conv_ms_per_step": conv_sec * 1e3,
                "igemm_ms_per_step": tsec["igemm"] * 1e3, "wgrad_ms_per_step": tsec["wgrad"] * 1e3,
                "whole_step_frac": flops / (elapsed / steps) / peak, "csrc_sha": csrc_sha()}
        # (data-gradient launches also compute the BatchNorm-backward sums of the tensor they write -- bn_bwd_reduce folded
        # into their epilogues; that time is charged to the convolutions here)
        if traffic is not None:
            roof["traffic_over_algorithmic"] = traffic * n_launch / alg_bytes
            roof["hbm_frac_of_same_launches"] = traffic * n_launch / conv_sec / 8e12
        # per-class two-roof table: the dozen most expensive classes on the line, every class in --dump-conv's side file and
        # in profiles/ (tools/run_r03_profiles.sh); the fp32 companion carries six
        roof["classes_cols"] = CLASS_COLS
        roof["classes"], full = conv_class_table(plan, dtype, top=12 if dtype in ("f16x2", "bf16") else 3, pmc_classes=pmc_classes)
        if dump_conv:
            with open(dump_conv + ".classes.json", "w") as fh:
                json.dump(full, fh, indent=0)
        res["roofline"] = roof
    del model, opt, sched, crit, step
    torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    main()
