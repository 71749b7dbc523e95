"""CPU: the C-ABI library loads and exports every symbol the header declares; host-side logic of the drop-in
packages (module tree, state_dict contract, optimizer ranges, LR schedule, sharding, bucketing).  No kernels
are launched here."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import helpers as H


def header_symbols():
    txt = open(os.path.join(H.ROOT, "include", "dmlnet_hip.h")).read()
    return sorted(set(re.findall(r"^(?:int|int64_t|const char\*)\s+(dml_\w+)\s*\(", txt, flags=re.M)))


def test_library_exports_every_declared_symbol():
    from dmlnet import _lib
    assert os.path.exists(_lib.LIB_PATH), "build with __graft_entry__.build()"
    lib = ctypes.CDLL(_lib.LIB_PATH)
    syms = header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), "missing export " + s
    assert sorted(_lib.EXPORTS) == syms, "binding and header disagree"
    # the product library exports exactly the header's entry points: no tuning / debug symbols (those live in `make tuning`)
    import subprocess
    nm = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in nm.splitlines() if ln.split()[-1].startswith("dml_") and " T " in ln})
    assert exported == syms, (sorted(set(exported) - set(syms)), sorted(set(syms) - set(exported)))
    lib2 = _lib.load()
    assert lib2.dml_abi_version() == 6
    assert lib2.dml_target_arch() == b"gfx950"


def test_struct_layout_matches_header():
    from dmlnet._lib import ConvDesc, WgradDesc
    # 5 ptr + 18 int32 | bnr: 5 ptr + 2 int32 | post: 4 ptr + 2 int32 | tail: ptr, int64, ptr, 2 int32 | res: 2 ptr + 2 int32 |
    # acc32: ptr + int32, f32_split, w_tiled, ws_min_tiles | planes: 4 ptr + 2 int64 | bnr_gmax | ABI 6: sub_grid, sub_y, sub_x, pad_w_set,
    # pad_w, reserved
    assert ctypes.sizeof(ConvDesc) == 360 and ConvDesc.bnr_gmax.offset == 328
    assert ConvDesc.sub_grid.offset == 336 and ConvDesc.pad_w_set.offset == 348 and ConvDesc.pad_w.offset == 352
    assert ConvDesc.B.offset == 40 and ConvDesc.mode.offset == 40 + 17 * 4
    assert ConvDesc.bnr_y.offset == 112 and ConvDesc.bnr_ldy.offset == 152 and ConvDesc.post_scale.offset == 160
    assert ConvDesc.tail_ws.offset == 200 and ConvDesc.tail_counters_len.offset == 224
    assert ConvDesc.res_dz.offset == 232 and ConvDesc.res_ld.offset == 248
    assert ConvDesc.acc32.offset == 256 and ConvDesc.acc32_ld.offset == 264 and ConvDesc.f32_split.offset == 268 and ConvDesc.w_tiled.offset == 272
    assert ConvDesc.ws_min_tiles.offset == 276 and ConvDesc.x_planes.offset == 280 and ConvDesc.w_plane_stride.offset == 320
    from dmlnet._lib import BnEvalDesc
    assert ctypes.sizeof(BnEvalDesc) == 48
    assert ctypes.sizeof(WgradDesc) == 3 * 8 + 17 * 4 + 4 + 8 + 8 + 2 * 4 + 6 * 8 and WgradDesc.f32_split.offset == 112 and WgradDesc.x_planes.offset == 120
    from dmlnet._lib import PrepDesc
    assert ctypes.sizeof(PrepDesc) == 48 and PrepDesc.w_tiled.offset == 40 and PrepDesc.wt_tiled.offset == 44
    from dmlnet._lib import AugSample
    assert ctypes.sizeof(AugSample) == 40 and AugSample.factor.offset == 28


def test_module_tree_and_state_dict_contract():
    import network
    from oracle import dmlnet_ref as O
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    o = O.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16)
    assert list(m.state_dict().keys()) == list(o.state_dict().keys())
    assert H.shapes_of(m) == H.shapes_of(o)
    assert len(m.state_dict()) == 674
    assert [k for k, _ in m.named_parameters()] == [k for k, _ in o.named_parameters()]
    # optimizer groups partition the parameters (main_embedding.py:385-388)
    nb, nc = sum(p.numel() for p in m.backbone.parameters()), sum(p.numel() for p in m.classifier.parameters())
    assert nb == 42500160 and nc == 16252528
    # OS8 variant: dilations as resnet.py:174-191 / modeling.py:8-10
    m8 = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=8, pretrained_backbone=False)
    assert m8.backbone.layer3[0].conv2.dilation == (1, 1) and m8.backbone.layer3[1].conv2.dilation == (2, 2)
    assert m8.backbone.layer4[0].conv2.dilation == (2, 2) and m8.backbone.layer4[1].conv2.dilation == (4, 4)
    assert m8.classifier.aspp.convs[3][0].dilation == (36, 36)
    assert m.backbone.layer4[0].conv2.dilation == (1, 1) and m.backbone.layer4[2].conv2.dilation == (2, 2)
    assert m.classifier.aspp.convs[1][0].padding == (6, 6)


def test_out_of_scope_factories_raise():
    import network
    for name in ("deeplabv3_resnet50", "deeplabv3plus_resnet101", "deeplabv3plus_mobilenet"):
        with pytest.raises(NotImplementedError):
            getattr(network, name)(num_classes=16, output_stride=16)
    with pytest.raises(NotImplementedError):
        network.convert_to_separable_conv(torch.nn.Conv2d(3, 3, 3))


def test_no_cpu_fallback():
    """The product path must refuse CPU tensors instead of silently computing somewhere else."""
    import network
    import utils
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    with pytest.raises(RuntimeError, match="ROCm device"):
        m.eval()(torch.zeros(1, 3, 32, 32))
    with pytest.raises(RuntimeError, match="HIP path only"):
        utils.DMLLoss()(torch.zeros(1, 4, 2, 2), torch.zeros(1, 2, 2, dtype=torch.long))
    with pytest.raises(RuntimeError, match="HIP path only"):
        utils.dissum_score(torch.zeros(1, 4, 2, 2))
    # and nothing in the package imports the oracle
    for root, _, files in os.walk(H.PKG):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_param_store_layout():
    import network
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=1))
    st = m._engine.store
    before = {k: v.clone() for k, v in m.state_dict().items()}
    st.bind(torch.device("cpu"))                       # flattening itself is device independent
    assert st.is_bound(torch.device("cpu"))
    after = m.state_dict()
    for k in before:
        assert torch.equal(before[k], after[k]), k
    w = m.backbone.layer1[0].conv2.weight
    assert w.shape == (64, 64, 3, 3) and w.stride() == (576, 1, 192, 64)     # K-R-S-C physical order
    assert st.total % 64 == 0 and st.split == 42500160
    # loading a checkpoint writes through the views
    m.load_state_dict(H.synth_state_dict(H.shapes_of(m), seed=2))
    off = st.offsets[st._index(w)]
    assert torch.equal(st.flat_p[off:off + w.numel()].view(64, 3, 3, 64), w.detach().permute(0, 2, 3, 1))


def test_fused_sgd_ranges_and_poly_lr():
    import network
    import utils
    from dmlnet.optim import FusedSGD
    m = network.deeplabv3plus_embedding_resnet101(num_classes=16, output_stride=16, pretrained_backbone=False)
    opt = FusedSGD([{"params": m.backbone.parameters(), "lr": 0.001},
                    {"params": m.classifier.parameters(), "lr": 0.01}], lr=0.01, momentum=0.9, weight_decay=1e-4).bind(m)
    opt._build_ranges()
    st = m._engine.store
    assert [(gi, r[1], r[2]) for gi, r in opt._ranges] == [(0, 0, st.split), (1, st.split, st.total - st.split)]
    sched = utils.PolyLR(opt, 100, power=0.9)
    lrs = []
    for _ in range(100):
        sched.step()
        lrs.append([g["lr"] for g in opt.param_groups])
    assert lrs[49][1] == pytest.approx(0.01 * 0.5 ** 0.9)
    assert lrs[49][0] == pytest.approx(0.001 * 0.5 ** 0.9)
    assert lrs[-1] == [1e-6, 1e-6]


def test_shard_and_buckets():
    from dmlnet.parallel import make_buckets, shard_range
    assert [shard_range(128, r, 8) for r in (0, 7)] == [(0, 16), (112, 128)]
    spans = [shard_range(10, r, 4) for r in range(4)]
    assert spans == [(0, 3), (3, 6), (6, 8), (8, 10)]
    offs, sizes = [0, 64, 192, 1216], [10, 128, 1000, 64]
    b = make_buckets(offs, sizes, 1280, 512)
    assert b[0][0] == 192 and b[0][1] == 1280 and b[-1][0] == 0
    assert sum(hi - lo for lo, hi, _ in b) == 1280
    assert sorted(i for _, _, mem in b for i in mem) == [0, 1, 2, 3]


def test_committed_bench_line_carries_the_contract_fields():
    """The newest profiles/rNN_bench_v*.json is a default `python bench.py` line of that round; the driver's contract
    fields, the roofline object (algorithmic bytes next to the PMC traffic when that was collected on the same kernel
    sources), the fp32 companion and the CPU baseline must all be there and be self-consistent."""
    import glob
    import json
    import re
    files = [f for f in glob.glob(os.path.join(H.ROOT, "profiles", "r*_bench_v*.json")) if re.search(r"r\d+_bench_v\d+\.json$", f)]
    newest = max(files, key=lambda f: tuple(int(x) for x in re.findall(r"r(\d+)_bench_v(\d+)", f)[0]))
    d = json.load(open(newest))
    assert os.path.basename(newest).startswith("r05"), newest
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "bf16_companion", "fp32_exact_companion",
              "hbm_kernel"):
        assert k in d, k
    assert d["unit"] == "images/sec" and d["n_gpus"] == 1 and d["higher_is_better"] is True and d["vs_baseline"] is None
    assert d["config"]["workload"] and "model" not in d["config"] and d["config"]["rccl_ranks"] == 1
    # the headline is the fp32-accurate step: fp32 tensors, products of two fp16 planes (MFMA roof 2.5 PF / 3)
    assert d["dtype"] == "f32" and "fp16" in d["config"]["arithmetic"] and abs(d["roofline"]["peak"] - 2500.0 / 3) < 1e-6
    assert abs(d["value"] - d["config"]["global_batch"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["algorithmic_bytes"] > 0 and (r["traffic"] is None or r["traffic"] > 0.5 * r["algorithmic_bytes"])
    assert abs(r["achieved"] * 1e12 - r["flops_per_step"] / (r["conv_ms_per_step"] * 1e-3)) < 1e-6 * r["achieved"] * 1e12
    # per-class two-roof table: rows follow classes_cols; every class's own bound is max(FLOPs / peak, bytes / 8 TB/s)
    cols = r["classes_cols"]
    assert len(cols) == 9 and len(r["classes"]) >= 12 and all(len(row) == len(cols) for row in r["classes"])
    assert all(row[8] is None or row[8] > 0.9 for row in r["classes"][:12])          # PMC bytes never below the algorithmic ones
    assert abs(sum(row[3] for row in r["classes"]) - r["conv_ms_per_step"]) < 0.02 * r["conv_ms_per_step"]
    assert all(row[4] in ("mfma", "hbm") and 0 < row[6] <= 1.0 for row in r["classes"])
    assert len(json.dumps(d)) < 9000          # the driver keeps the tail of stdout: the line stays a few KB
    f = d["fp32_exact_companion"]
    assert f["dtype"] == "f32" and f["value"] > 0 and abs(f["roofline"]["frac"] - f["roofline"]["achieved"] / 157.3) < 1e-6
    assert f["steps"] >= 10 and f["warmup"] >= 2          # (the fp32-accurate mode is the headline now; the exact one rides along)
    assert f["roofline"]["traffic"] is None or f["roofline"]["traffic"] > 0.5 * f["roofline"]["algorithmic_bytes"]
    assert f["three_term_split"]["dtype"] == "f32x3" and f["value"] < f["three_term_split"]["value"] < d["value"]
    b = d["bf16_companion"]
    assert b["dtype"] == "bf16" and b["value"] > d["value"] and abs(b["roofline"]["frac"] - b["roofline"]["achieved"] / 2500.0) < 1e-6
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["value"] > 0 and c["sample"] and str(c["cores"]) in c["probe_by_threads"] and len(c["timed_steps_s"]) >= 3
    h = d["hbm_kernel"]
    assert h["bound"] == "hbm" and abs(h["frac"] - h["achieved"] / h["peak"]) < 1e-9
    assert h["traffic"] is None or h["traffic"] > 0.99 * 192 * 16 * 768 * 768
    assert len(h["step_path"]) >= 2 and all(e["ms"] > 0 and e["bytes_per_px"] > 0 for e in h["step_path"])
    assert d["host_enqueue_ms_per_step"] > 0


def test_native_plan_packing_and_dispatch_without_a_gpu():
    """dml_plan_run / DmlPlanOp on the host side only: entry-point ids, argument packing (pointers, negative ints,
    floats, doubles, by-reference descriptors, indirect int32), write-through of per-step argument patches, and the
    failed-op report -- using calls that the entry points reject during argument validation, i.e. before any HIP call."""
    import ctypes as C
    import struct
    from dmlnet import _lib, engine as E
    lib = _lib.load()
    assert lib.dml_plan_fn_id(b"dml_fill_f32") >= 0 and lib.dml_plan_fn_id(b"dml_no_such_entry") == -1
    for name in ("dml_conv_igemm", "dml_conv_wgrad_group", "dml_bn_bwd_apply", "dml_head_bwd_fused", "dml_sgd_step"):
        fid = lib.dml_plan_fn_id(name.encode())
        assert fid >= 0 and lib.dml_plan_fn_nargs(fid) == len(getattr(lib, name).argtypes) - 1, name
    # word packing
    assert E._pack_word(None, C.c_void_p) == (0, False)
    assert E._pack_word(-3, C.c_int) == (0xFFFFFFFFFFFFFFFD, False)
    assert E._pack_word(1.5, C.c_float) == (struct.unpack("<I", struct.pack("<f", 1.5))[0], False)
    assert E._pack_word(-2.25, C.c_double) == (struct.unpack("<Q", struct.pack("<d", -2.25))[0], False)
    boxed = C.c_int(7)
    w, ind = E._pack_word(boxed, C.c_int)
    assert ind and w == C.addressof(boxed)
    d = _lib.ConvDesc()
    assert E._pack_word(C.byref(d), C.c_void_p) == (C.addressof(d), False)
    # a launch list of three ops that all fail validation before launching anything: the first one must be reported
    ops = []
    a0 = E.BoundArgs([None, 16, 1.0])                     # dml_fill_f32(NULL, ...) -> DML_EINVAL
    ops.append((lib.dml_fill_f32, a0))
    a1 = E.BoundArgs([None, None, 0, 1, 1])               # dml_convert_dtype(NULL, ...) -> DML_EINVAL
    ops.append((lib.dml_convert_dtype, a1))
    nat = E.NativeList(lib, ops)
    assert nat.arr[0].nargs == 3 and nat.arr[0].args[1] == 16 and nat.arr[1].nargs == 5
    failed = C.c_int(-1)
    assert lib.dml_plan_run(nat.arr, 0, 2, None, None, None, 0, C.byref(failed)) == -1 and failed.value == 0
    assert lib.dml_plan_run(nat.arr, 1, 2, None, None, None, 0, C.byref(failed)) == -1 and failed.value == 1
    assert lib.dml_plan_run(nat.arr, 1, 1, None, None, None, 0, C.byref(failed)) == 0          # empty range
    a0[1] = -5                                            # per-step patch: written through to the packed copy
    assert nat.arr[0].args[1] == 0xFFFFFFFFFFFFFFFB and a0[1] == -5
    a0[2] = 0.25
    assert nat.arr[0].args[2] == struct.unpack("<I", struct.pack("<f", 0.25))[0]
    # a Python step in the list stays outside the native array
    ops.append((lambda stream: 0, E.BoundArgs()))
    nat2 = E.NativeList(lib, ops)
    assert nat2.python_ops == {2} and nat2.arr[2].fn == -1
    # malformed op -> rejected with its index
    nat.arr[1].nargs = 4
    assert lib.dml_plan_run(nat.arr, 1, 2, None, None, None, 0, C.byref(failed)) == -1 and failed.value == 1


def test_grouped_weight_gradient_eligibility_is_host_logic():
    """dml_conv_wgrad_group_eligible decides on the host which layers the plan may put into a grouped launch."""
    import ctypes as C
    from dmlnet import _lib
    lib = _lib.load()

    def desc(**kw):
        base = dict(x=0x1000, dy=0x2000, dw=0x3000, B=16, Hi=48, Wi=48, C=256, ldx=256, Ho=48, Wo=48, N=256, ldy=256, R=3, S=3,
                    stride=1, dil=1, pad=1, dtype=1, splitk=0, Cm=0, ws=None, ws_elems=0)
        base.update(kw)
        return _lib.WgradDesc(**base)

    assert lib.dml_conv_wgrad_group_eligible(C.byref(desc())) == 1                                  # layer3 3x3
    assert lib.dml_conv_wgrad_group_eligible(C.byref(desc(C=1024, ldx=1024, R=1, S=1, pad=0))) == 1   # 1x1 1024 -> 256
    assert lib.dml_conv_wgrad_group_eligible(C.byref(desc(dtype=0))) == 0                           # fp32 plans
    assert lib.dml_conv_wgrad_group_eligible(C.byref(desc(N=128, ldy=128))) == 0                    # not a whole 256-channel tile
    assert lib.dml_conv_wgrad_group_eligible(C.byref(desc(C=64, ldx=64, R=1, S=1, pad=0, N=256))) == 0   # one output tile only
    assert lib.dml_conv_wgrad_group_eligible(C.byref(desc(x=None))) == 0
    assert lib.dml_conv_wgrad_group(None, 1, None, 0, None) == -1
