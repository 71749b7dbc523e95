"""GPU: the LDS-DMA conv kernel (buffer_load ... lds ring) is the default for every eligible bf16 shape; the
register-staged kernel then only sees fp32 / unaligned shapes -- force it for everything with DML_CONV_V1=1 in a
child process and run the conv parity tests through it as well.  Plus end-to-end runs of scripts in child processes."""
import os
import subprocess
import sys

import numpy as np
import pytest

import helpers as H

pytestmark = pytest.mark.gpu


def test_conv_ops_through_the_register_staged_kernel():
    env = dict(os.environ, DML_CONV_V1="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(H.ROOT, "tests", "test_gpu_ops.py"), "-m", "gpu",
                        "-q", "-x", "-k", "conv and not wave_specialised and not two_plane", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "passed" in r.stdout


def test_every_unit_locally_exact_through_the_256_row_tiles():
    """The 256-row tile of the LDS-DMA kernel (4 waves on 128 x 64 wave tiles, conv_igemm_dma_kernel<.., 256, 128>) runs by itself
    only on the long-K layers (K >= 4608, or the decoder's 3x3 at 16 x 192 x 192); DML_CONV_BM256=2 forces it on every eligible
    layer of a bf16 train step -- fused statistics, fused BN-backward sums, accumulate, the K-split tail -- and every unit must
    still be exact to one bf16 ulp (all three cases of the locally-exact gate incl. 768 x 768)."""
    env = dict(os.environ, DML_CONV_BM256="2")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(H.ROOT, "tests", "test_gpu_bf16_parity.py"), "-m", "gpu",
                        "-q", "-x", "-k", "locally_exact", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=H.ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "3 passed" in r.stdout


def test_every_unit_locally_exact_through_the_wave_specialised_kernel():
    """DML_WS_MIN_TILES=1 (DmlConvDesc.ws_min_tiles of every conv of the plan) + DML_CONV_WS=1 sends every eligible layer of
    the bf16 train step through conv_ws_kernel -- 144-row tiles, 48-row statistics groups, fused BN-backward sums and
    identity adds in its epilogue -- and every unit must still be exact to one bf16 ulp on its stored inputs."""
    env = dict(os.environ, DML_WS_MIN_TILES="1", DML_CONV_WS="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(H.ROOT, "tests", "test_gpu_bf16_parity.py"), "-m", "gpu",
                        "-q", "-x", "-k", "locally_exact", "-p", "no:cacheprovider"], env=env, capture_output=True, text=True,
                       cwd=H.ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "3 passed" in r.stdout


@pytest.mark.parametrize("products", ["exact", "f16x2"])
def test_sync_batchnorm_two_ranks_match_one_process(products):
    """set_sync_batchnorm(True): two ranks (gloo, sharing this GPU) with half a batch each reproduce the single-process
    whole-batch logits, loss, running statistics and reduced gradients (tools/check_syncbn.py) -- also in the two-plane mode,
    whose plane scales come from bounds over the GLOBAL element count (dml_h2_bound_bn with count = M x ranks)."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, DML_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", DML_F32_PRODUCTS=products)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(port), os.path.join(H.ROOT, "tools", "check_syncbn.py")], env=env,
                       capture_output=True, text=True, cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert "rank 0:" in r.stdout and "rank 1:" in r.stdout


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("products", ["exact", "f16x2"])
def test_data_parallel_reducer_two_ranks(products):
    """The product's N > 1 path (GradReducer.run_backward on the real ParamStore: reverse-order buckets launched from
    Plan.param_last_op on the comm stream while the backward plan and the weight-gradient side stream keep running),
    two gloo ranks sharing this GPU with UNEVEN shards: tools/check_ddp.py."""
    env = dict(os.environ, DML_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", DML_F32_PRODUCTS=products)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(H.ROOT, "tools", "check_ddp.py")],
                       env=env, capture_output=True, text=True, cwd=H.ROOT, timeout=1200)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-3000:]
    assert "parameters identical across ranks after 3 steps: True" in r.stdout and "uneven shards" in r.stdout


def test_data_parallel_reducer_four_ranks_small_and_large_buckets():
    """config #4 readiness without the hardware: FOUR gloo ranks sharing this GPU, 2-3 images per rank (uneven), with
    1 MB buckets (225 collectives per step, launched all along the backward) and 32 MB buckets (the bench's setting):
    reduced gradient = sum of the per-rank ones, replicas bit-identical after 3 optimizer steps (tools/check_ddp.py)."""
    env = dict(os.environ, DML_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", DDP_BUCKET_MBS="1,32", DDP_DTYPES="f32")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "4", "--master-addr",
                        "127.0.0.1", "--master-port", str(_free_port()), os.path.join(H.ROOT, "tools", "check_ddp.py")],
                       env=env, capture_output=True, text=True, cwd=H.ROOT, timeout=1500)
    assert r.returncode == 0, r.stdout[-4000:] + r.stderr[-3000:]
    assert r.stdout.count("parameters identical across ranks after 3 steps: True") == 8          # 4 ranks x 2 bucket sizes
    assert "1 MB buckets" in r.stdout and "32 MB buckets" in r.stdout


def test_reducer_launch_points_overlap_the_backward_at_768():
    """Overlap evidence for the 8-rank run this pool cannot execute: at the benchmark's shape the reducer's 32 MB buckets
    (reverse parameter order) become ready -- Plan.param_last_op: the last backward op that writes into the bucket --
    while most of the backward is still to come.  Counted in conv FLOPs of the backward plan still ahead at each launch
    point: all buckets but the one holding the stem's parameters start before the last op, and half of them with > 30 %
    of the backward left to hide a 32 MB all-reduce under."""
    import torch
    import network
    import utils
    import bench
    from dmlnet import parallel
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.cuda().train()
    m.set_compute_dtype(torch.bfloat16)
    utils.set_bn_momentum(m.backbone, 0.01)
    img = torch.randn(2, 3, 768, 768, device="cuda")
    m(img)
    plan = next(p for k, p in m._engine.plans.items() if k[4])
    red = parallel.GradReducer(m._engine.store, bucket_mb=32.0, average=False)
    sched = red._schedule(plan)
    n_b = len(red.buckets)          # 235 MB in buckets of at least 32 MB: 7
    assert n_b in (7, 8) and sum(hi - lo for lo, hi, _ in red.buckets) == m._engine.store.total
    _, per_op = bench.conv_flops_of_plan(plan)
    bwd = {i: f for (name, i), f in per_op.items() if name == "bwd"}
    total = sum(bwd.values())
    rows = []
    for op_i in sorted(sched):
        left = sum(f for i, f in bwd.items() if i > op_i) / total
        for b in sched[op_i]:
            lo, hi, _ = red.buckets[b]
            rows.append((b, op_i, (hi - lo) * 4 / 2 ** 20, left))
    print("bucket  launch op (of %d)  MB     backward conv FLOPs still ahead" % len(plan.bwd))
    for b, op_i, mb, left in rows:
        print("%4d %12d %9.1f %10.1f %%" % (b, op_i, mb, 100 * left))
    early = [r for r in rows if r[1] < len(plan.bwd) - 1]
    assert len(early) >= n_b - 1, rows          # only the bucket holding the stem's parameters waits for the last op
    assert sum(1 for r in rows if r[3] > 0.30) >= n_b // 2, rows
    launch_order = [r[0] for r in rows]
    assert launch_order == sorted(launch_order)          # bucket 0 (the head's parameters) first, the stem's last


def test_rccl_backend_runs_the_reducer_with_one_forced_rank():
    """The nccl (= RCCL) backend on real hardware: this pool has one GPU per box, so the only way to execute the RCCL code
    path -- communicator setup, the in-place all-reduce on slices of the flat gradient buffer from the comm stream, the
    loss-sum all-reduce on the device, barrier / max-over-ranks timing -- is a single rank with DML_FORCE_DIST=1.  The
    step must then equal the plain single-GPU step (sum over one rank = identity) and report backend nccl."""
    import json
    # fp32: the second step's loss depends on the first step's reduced gradients; in bf16 at this size (BatchNorm over 2 x 8 x 8
    # samples in layer3 / layer4) rounding-level differences between two runs grow to ~1 % of the loss within three steps
    # (measured: 4.757 vs 4.722), which would hide a wrong reduction behind the tolerance
    args = [sys.executable, os.path.join(H.ROOT, "bench.py"), "--size", "128", "--batch", "2", "--steps", "1", "--warmup", "1",
            "--dtype", "f32", "--no-cpu-baseline", "--no-profile", "--no-fp32-companion"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()),
               RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    out = {}
    for forced in ("0", "1"):
        e = dict(env, DML_FORCE_DIST=forced)
        e.pop("DML_DIST_BACKEND", None)
        r = subprocess.run(args, capture_output=True, text=True, cwd=H.ROOT, timeout=900, env=e)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
        assert len(lines) == 1, lines
        out[forced] = json.loads(lines[0])
    assert out["0"]["config"]["backend"] is None and out["1"]["config"]["backend"] == "nccl"
    assert out["1"]["config"]["rccl_ranks"] == 1 and out["1"]["n_gpus"] == 1
    a, b = out["0"]["config"]["final_loss"], out["1"]["config"]["final_loss"]
    assert abs(a - b) <= 2e-3 * abs(a), (a, b)         # same seeds, same batch


def test_bench_gpus_flag_launches_ranks_itself():
    """`python bench.py --gpus N` (no torchrun): N = 2 on this 1-GPU box must fail cleanly before any GPU work, and with
    DML_BENCH_ALLOW_SHARED_GPU=1 (test hook: ranks share device 0 over gloo) it must print ONE line with n_gpus 2."""
    import json
    args = [sys.executable, os.path.join(H.ROOT, "bench.py"), "--gpus", "2", "--size", "128", "--batch", "2", "--steps", "2",
            "--warmup", "1", "--no-cpu-baseline"]
    if __import__("torch").cuda.device_count() < 2:
        r = subprocess.run(args, capture_output=True, text=True, cwd=H.ROOT, timeout=600)
        assert r.returncode != 0 and "GPU" in (r.stderr + r.stdout)
    env = dict(os.environ, DML_BENCH_ALLOW_SHARED_GPU="1", DML_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(args, capture_output=True, text=True, cwd=H.ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [ln for ln in r.stdout.strip().splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["rccl_ranks"] == 2 and d["config"]["backend"] == "gloo"
    assert d["config"]["global_batch"] == 4 and d["value"] > 0
    # self-diagnosis of the first real multi-GPU run: one 32 MB all-reduce on the comm stream, time and bus bandwidth on the line
    assert d["config"]["allreduce_32mb_ms"] > 0 and d["config"]["allreduce_32mb_busbw_GBps"] > 0 and d["config"]["allreduce_backend"] == "gloo"


def test_bench_script_default_path_small():
    """bench.py end to end (timed region, profiled conv pass, distance kernel, input pipeline) at a small size: the
    JSON line must carry the contract's fields."""
    import json
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--size", "128", "--batch", "2", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "hbm_kernel", "input_pipeline"):
        assert k in d, k
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["achieved"] > 0 and d["value"] > 0
    # the headline is the fp32-accurate mode (fp32 tensors, two-term fp16 split of the conv products); bf16 storage, the exact
    # fp32 MFMA and the three-term split ride along as companions
    # (the line says so: dtype "f16x2", never "f32", and the exact-fp32 mode's rate is a first-class field next to `value`)
    assert d["dtype"] == "f16x2" and "two-term" in d["config"]["arithmetic"]
    assert d["value_fp32_exact"] == d["fp32_exact_companion"]["value"] and d["fp32_exact_companion"]["dtype"] == "f32"
    assert d["bf16_companion"]["value"] > 0 and d["bf16_companion"]["roofline"]["achieved"] > 0
    assert d["fp32_exact_companion"]["value"] > 0 and d["fp32_exact_companion"]["three_term_split"]["value"] > 0
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--mode", "infer", "--height", "128", "--width",
                        "256", "--steps", "2", "--warmup", "1"], capture_output=True, text=True, cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    di = json.loads(r.stdout.strip().splitlines()[-1])
    assert di["value"] > 0 and di["dtype"] == "f16x2" and di["roofline"]["achieved"] > 0
    r = subprocess.run([sys.executable, os.path.join(H.ROOT, "bench.py"), "--mode", "fwd", "--size", "128", "--batch", "2", "--dtype", "bf16",
                        "--steps", "2", "--warmup", "1"], capture_output=True, text=True, cwd=H.ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    df = json.loads(r.stdout.strip().splitlines()[-1])
    assert df["value"] > 0 and df["dtype"] == "bf16" and "forward-only" in df["metric"]


@pytest.mark.parametrize("dtype", ["bf16", "f16x2"])
def test_train_driver_synthetic_with_raw_label_ids(tmp_path, dtype):
    """main_embedding.py end to end at a small size: raw-id label frames -> crop / jitter / flip / encode_target in one
    kernel -> train steps -> validation (device confusion matrix) -> checkpoint; in the throughput mode and in bench.py's headline
    arithmetic (--dtype f16x2: reachable from the entry point north_star names)."""
    drv = os.path.join(H.PKG, "main_embedding.py")
    r = subprocess.run([sys.executable, drv, "--synthetic", "--crop_size", "128", "--batch_size", "4", "--total_itrs", "4",
                        "--print_interval", "2", "--val_interval", "4", "--val_images", "1", "--frame_height", "160",
                        "--frame_width", "224", "--loss_type", "dml", "--dtype", dtype, "--save_dir", str(tmp_path)],
                       capture_output=True, text=True, cwd=H.PKG, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Itrs 4/4, Loss=" in r.stdout and "Mean IoU" in r.stdout
    loss = float(r.stdout.split("Itrs 4/4, Loss=")[1].split(",")[0])
    assert np.isfinite(loss) and loss > 0
    assert any(f.endswith(".pth") for f in os.listdir(tmp_path))


def test_incremental_head_driver_synthetic(tmp_path):
    """main_self_distillation.py end to end at a small size (base checkpoint -> two-head model, pseudo-labels on)."""
    import torch
    import network
    base = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    base.load_state_dict(H.synth_state_dict(H.shapes_of(base), seed=3))      # a conditioned trunk: running statistics are used as they are
    ck = os.path.join(str(tmp_path), "base.pth")
    torch.save({"model_state": base.state_dict(), "cur_itrs": 0, "best_score": 0.0}, ck)
    drv = os.path.join(H.PKG, "main_self_distillation.py")
    r = subprocess.run([sys.executable, drv, "--synthetic", "--crop_size", "128", "--batch_size", "4", "--total_itrs", "4",
                        "--print_interval", "2", "--frame_height", "160", "--frame_width", "224", "--ckpt", ck,
                        "--pseudo_labels", "--save_interval", "4", "--save_dir", str(tmp_path)],
                       capture_output=True, text=True, cwd=H.PKG, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "Itrs 4/4, Loss=" in r.stdout
    loss = float(r.stdout.split("Itrs 4/4, Loss=")[1].split(",")[0])
    assert np.isfinite(loss) and loss > 0
    out = torch.load(os.path.join(str(tmp_path), "latest_deeplabv3plus_embedding_self_distillation_resnet101_synthetic.pth"),
                     map_location="cpu")["model_state"]
    sd = base.state_dict()
    # trunk and base head untouched by the four steps, the new head moved
    assert torch.equal(out["backbone.layer3.5.conv2.weight"], sd["backbone.layer3.5.conv2.weight"])
    assert torch.equal(out["classifier.classifier.3.weight"], sd["classifier.classifier.3.weight"])
    assert torch.equal(out["backbone.bn1.running_mean"], sd["backbone.bn1.running_mean"])


@pytest.mark.parametrize("ood,dtype", [("dissum", "bf16"), ("msp", "bf16"), ("maxlogit", "bf16"), ("dissum", "f16x2")])
def test_open_set_evaluation_driver_synthetic(ood, dtype):
    """eval_ood_traditional.py end to end at a small frame size (five concurrent scales, graphs, device scores / AUROC)."""
    drv = os.path.join(H.PKG, "eval_ood_traditional.py")
    r = subprocess.run([sys.executable, drv, "--synthetic", "--ood", ood, "--num_images", "2", "--height", "360", "--width",
                        "640", "--dtype", dtype], capture_output=True, text=True, cwd=H.PKG, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "mean auroc = " in r.stdout and "Mean IoU:" in r.stdout
    auroc = float(r.stdout.split("mean auroc = ")[1].split()[0])
    assert 0.0 <= auroc <= 1.0
