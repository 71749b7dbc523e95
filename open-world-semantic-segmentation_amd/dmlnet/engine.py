"""Static-plan executor for DeepLabV3+/ResNet-101 + distance head on MI355X.

One `Plan` per (batch, H, W, dtype, training) holds every activation / gradient buffer (NHWC, sized once:
288 GB of HBM makes recomputation or buffer reuse unnecessary at 768x768 bs=16) and two flat lists of
pre-bound C-ABI calls -- forward and backward -- that are replayed each step on the caller's current HIP
stream.  PyTorch only supplies device memory, streams and the autograd hook; every FLOP runs in
libdmlnet_hip.so.

Graph structure follows the reference modules (paths relative to /root/reference/DeepLabV3Plus-Pytorch):
  backbone  network/backbone/resnet.py:75-115,139-143,171-193   (Bottleneck, stem, stride->dilation)
  head      network/utils.py:8-32,308-361                       (DeepLabHeadV3Plus, ASPP)
  distance  network/utils.py:84-118                             (upsample + prototype distances)
"""
from __future__ import annotations

import ctypes as C
import itertools
import os
import struct
import weakref
from typing import List, Optional

import numpy as np
import torch
import torch.distributed as dist
import torch.nn as nn

from . import _lib, lazy_grad
from ._lib import DML_BF16, DML_F32, STAT_ROWS, ConvDesc, WgradDesc

_PAD_CIN = 8          # stem input channels 3 -> 8 so that a 16-byte vector never straddles a filter tap


class _S2DConv:
    """The stem's k x k stride-2 convolution (resnet.py:139: 7x7, padding 3, on the 3 image channels) in space-to-depth form: a
    (k + 1) / 2 square stride-1 convolution on [B][H/2][W/2][4 C] (dml_pack_input_s2d) with the weights regrouped per step
    (dml_s2d_weights) and the weight gradient scattered back into the parameter's layout (dml_s2d_wgrad).  Same products on
    K = 192 instead of the 392 of the image padded to 8 channels.  Quacks like the nn.Conv2d the plan builder reads."""

    def __init__(self, conv: nn.Conv2d, Ho: int, Wo: int):
        k, p = conv.kernel_size[0], conv.padding[0]
        self.real, self.k = conv, k
        self.kernel_size, self.stride, self.dilation = ((k + 1) // 2,) * 2, (1, 1), (1, 1)
        self.padding = ((p + 1) // 2,) * 2
        self.in_channels, self.out_channels = 4 * conv.in_channels, conv.out_channels
        self.weight, self.bias = conv.weight, conv.bias
        self.out_hw = (Ho, Wo)

    @staticmethod
    def fits(conv: nn.Conv2d, H: int, W: int) -> bool:
        k, p = conv.kernel_size[0], conv.padding[0]
        return (conv.kernel_size == (k, k) and k % 2 == 1 and conv.stride == (2, 2) and conv.padding == (p, p) and p == (k - 1) // 2
                and p % 2 == 1 and conv.dilation == (1, 1) and conv.groups == 1 and conv.bias is None and H % 2 == 0 and W % 2 == 0)


def _dt(dtype: torch.dtype) -> int:
    if dtype == torch.float32:
        return DML_F32
    if dtype == torch.bfloat16:
        return DML_BF16
    raise ValueError("compute dtype must be torch.float32 or torch.bfloat16, got %r" % (dtype,))


def _round_up(a: int, b: int) -> int:
    return (a + b - 1) // b * b


# ---------------------------------------------------------------------------------------------------
# native launch lists (dml_plan_run)
# ---------------------------------------------------------------------------------------------------
_M64 = 0xFFFFFFFFFFFFFFFF


def _pack_word(v, ctype):
    """one argument -> (64-bit word, indirect flag) in the convention of DmlPlanOp (include/dmlnet_hip.h)"""
    if v is None:
        return 0, False
    if ctype is C.c_float:
        return struct.unpack("<I", struct.pack("<f", float(v)))[0], False
    if ctype is C.c_double:
        return struct.unpack("<Q", struct.pack("<d", float(v)))[0], False
    if isinstance(v, int):
        return v & _M64, False
    if hasattr(v, "_obj"):                         # C.byref(x): the address of x (descriptor structs, int* out-parameters)
        return C.addressof(v._obj), False
    if isinstance(v, C._SimpleCData):              # a c_int passed by value that another op fills in at issue time
        if C.sizeof(v) != 4:
            raise TypeError("only 32-bit indirect arguments are supported")
        return C.addressof(v), True
    raise TypeError("cannot pack plan argument %r for %r" % (v, ctype))


class BoundArgs(list):
    """The argument list of one plan op.  Plain list for the Python replay; once the op has a slot in a native launch
    list, item assignment (per-step pointers, seeds, momenta) writes through to the packed copy."""
    __slots__ = ("slot", "types", "meta")

    def __init__(self, it=()):
        super().__init__(it)
        self.slot, self.types, self.meta = None, None, None

    def __setitem__(self, k, v):
        super().__setitem__(k, v)
        if self.slot is not None:
            w, ind = _pack_word(v, self.types[k])
            self.slot.args[k] = w
            if ind:
                self.slot.indirect |= (1 << k)
            else:
                self.slot.indirect &= ~(1 << k) & 0xFFFFFFFF


class NativeList:
    """Packed copy of a plan's op list for dml_plan_run; ops that are Python callables (collectives) stay outside."""

    def __init__(self, lib, ops, side=None):
        self.lib = lib
        n = len(ops)
        self.arr = (_lib.PlanOp * max(n, 1))()
        self.python_ops = set()
        ids = {}
        for i, (fn, args) in enumerate(ops):
            name = getattr(fn, "__name__", None)
            types = getattr(fn, "argtypes", None)
            if name is None or types is None or not name.startswith("dml_"):
                self.python_ops.add(i)
                self.arr[i].fn = -1
                continue
            if name not in ids:
                ids[name] = lib.dml_plan_fn_id(name.encode())
            fid = ids[name]
            if fid < 0 or lib.dml_plan_fn_nargs(fid) != len(args) or len(args) > _lib.PLAN_MAX_ARGS \
                    or len(types) != len(args) + 1:
                raise RuntimeError("plan op %d (%s) does not fit the native launch list" % (i, name))
            slot = self.arr[i]
            slot.fn, slot.nargs, slot.indirect = fid, len(args), 0
            flag = side.get(i) if side else None
            slot.stream, slot.wait = (0, 0) if flag is None else (1, 1 if flag else 0)
            if not isinstance(args, BoundArgs):
                raise TypeError("plan op arguments must come from Plan.call()")
            args.slot, args.types = slot, types
            for k, v in enumerate(args):
                args[k] = v                          # packs through __setitem__
        self.n = n
        self.failed = C.c_int(-1)
        self.on_error = None

    def run(self, first, last, stream, side_stream=None, events=None, n_events=0, marks=None):
        if marks is not None:      # (marks, count, event handles): dml_plan_run_marks records an event pair after each marked op
            rc = self.lib.dml_plan_run_marks(self.arr, first, last, stream, side_stream, events, n_events, marks[0], marks[1], marks[2],
                                             C.byref(self.failed))
        else:
            rc = self.lib.dml_plan_run(self.arr, first, last, stream, side_stream, events, n_events, C.byref(self.failed))
        if rc:
            if self.on_error is not None:
                self.on_error()
            _lib.check(rc, "native plan op %d" % self.failed.value)


# ---------------------------------------------------------------------------------------------------
# flat parameter storage
# ---------------------------------------------------------------------------------------------------
class ParamStore:
    """All parameters in ONE fp32 buffer (conv weights physically K-R-S-C), gradients and SGD momentum
    in two more.  nn.Parameters become views, so `state_dict()` keeps the reference's 674 keys and OIHW
    shapes, the optimizer is one streaming kernel per LR group and the gradient all-reduce works on
    contiguous buckets."""

    ALIGN = 64

    def __init__(self, model: nn.Module):
        self.model = model
        heads = model.head_modules() if hasattr(model, "head_modules") else [model.classifier]
        self.params: List[nn.Parameter] = list(model.backbone.parameters())
        for h in heads:
            self.params += list(h.parameters())
        self.n_backbone = len(list(model.backbone.parameters()))
        self.offsets: List[int] = []
        off = 0
        for p in self.params:
            self.offsets.append(off)
            off += _round_up(p.numel(), self.ALIGN)
        self.total = off
        self.split = self.offsets[self.n_backbone] if self.n_backbone < len(self.params) else off
        self.flat_p: Optional[torch.Tensor] = None
        self.flat_g: Optional[torch.Tensor] = None
        self.flat_v: Optional[torch.Tensor] = None     # momentum, created by the optimizer
        self.bn_modules = [m for m in itertools.chain(model.backbone.modules(), *[h.modules() for h in heads])
                           if isinstance(m, nn.BatchNorm2d)]
        self.flat_rs: Optional[torch.Tensor] = None
        self.flat_nbt: Optional[torch.Tensor] = None
        self.version = 0            # bumped by FusedSGD.step(); see Plan.refresh_weights
        self.grad_views: List[torch.Tensor] = []

    @staticmethod
    def _view(flat, off, p):
        n = p.numel()
        seg = flat[off:off + n]
        if p.dim() == 4:
            o, i, kh, kw = p.shape
            return seg.view(o, kh, kw, i).permute(0, 3, 1, 2)
        return seg.view(p.shape)

    def is_bound(self, device) -> bool:
        if self.flat_p is None or self.flat_p.device != device:
            return False
        base = self.flat_p.data_ptr()
        for idx in (0, len(self.params) // 2, len(self.params) - 1):
            p = self.params[idx]
            if p.device != device or p.data_ptr() != base + 4 * self.offsets[idx] or p.dtype != torch.float32:
                return False
        return True

    @torch.no_grad()
    def bind(self, device):
        """(Re)build the flat buffers from the current parameter values and re-point the modules at them."""
        flat_p = torch.zeros(self.total, dtype=torch.float32, device=device)
        for p, off in zip(self.params, self.offsets):
            src = p.detach().to(device=device, dtype=torch.float32)
            if src.dim() == 4:
                src = src.permute(0, 2, 3, 1)
            flat_p[off:off + p.numel()].copy_(src.reshape(-1))
        for p, off in zip(self.params, self.offsets):
            p.data = self._view(flat_p, off, p)
            p.grad = None
        self.flat_p = flat_p
        self.flat_g = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad_views = [self._view(self.flat_g, off, p) for p, off in zip(self.params, self.offsets)]
        self.flat_v = None
        # running statistics
        n_rs = sum(2 * m.num_features for m in self.bn_modules)
        flat_rs = torch.zeros(n_rs, dtype=torch.float32, device=device)
        flat_nbt = torch.zeros(len(self.bn_modules), dtype=torch.int64, device=device)
        off = 0
        for i, m in enumerate(self.bn_modules):
            c = m.num_features
            flat_rs[off:off + c].copy_(m.running_mean.detach().to(device))
            flat_rs[off + c:off + 2 * c].copy_(m.running_var.detach().to(device))
            m.running_mean.data = flat_rs[off:off + c]
            m.running_var.data = flat_rs[off + c:off + 2 * c]
            flat_nbt[i] = int(m.num_batches_tracked)
            m.num_batches_tracked.data = flat_nbt[i]
            off += 2 * c
        self.flat_rs, self.flat_nbt = flat_rs, flat_nbt
        self.version += 1

    def ptr_of(self, p: nn.Parameter) -> int:
        return p.data_ptr()

    def grad_ptr_of(self, p: nn.Parameter) -> int:
        idx = self._index(p)
        return self.flat_g.data_ptr() + 4 * self.offsets[idx]

    def _index(self, p):
        if not hasattr(self, "_idmap"):
            self._idmap = {id(q): i for i, q in enumerate(self.params)}
        return self._idmap[id(p)]

    # gradient bookkeeping around a backward pass ----------------------------------------------
    def begin_backward(self) -> str:
        """Returns 'fresh' (flat_g zeroed, grads will be attached) or 'accumulate'."""
        states = set()
        for p, gv in zip(self.params, self.grad_views):
            if p.grad is None:
                states.add("none")
            elif p.grad.data_ptr() == gv.data_ptr():
                states.add("ours")
            else:
                states.add("foreign")
        if states <= {"none"}:
            self.flat_g.zero_()
            return "fresh"
        if states <= {"ours"}:
            return "accumulate"
        # mixed: fold whatever the user had into our buffer, then accumulate on top of it
        with torch.no_grad():
            for p, gv in zip(self.params, self.grad_views):
                if p.grad is None:
                    gv.zero_()
                elif p.grad.data_ptr() != gv.data_ptr():
                    gv.copy_(p.grad)
        return "accumulate"

    def end_backward(self):
        for p, gv in zip(self.params, self.grad_views):
            if p.grad is None or p.grad.data_ptr() != gv.data_ptr():
                p.grad = gv


# ---------------------------------------------------------------------------------------------------
# plan building blocks
# ---------------------------------------------------------------------------------------------------
class Act:
    """An NHWC activation (or gradient) living in a plan-owned buffer; may be a channel slice."""
    __slots__ = ("t", "ptr", "B", "H", "W", "C", "ld", "f32", "es", "grad", "grad_init", "root", "g32", "h2", "amax", "h2_used")

    def __init__(self, t, ptr, B, H, W, C, ld, f32, es):
        self.t, self.ptr, self.B, self.H, self.W, self.C, self.ld, self.f32, self.es = t, ptr, B, H, W, C, ld, f32, es
        self.grad = None            # Act holding d(loss)/d(this)
        self.grad_init = False      # has any producer written the gradient yet?
        self.root = self            # concat buffer this is a slice of
        self.g32 = None             # fp32 staging of the gradient while it still has producers to come (Plan.stage_grad32)
        self.h2 = None              # (fp16 hi / lo planes, work) of this fp32 tensor once a conv has asked for them (Plan.h2_of)
        self.amax = None            # its `work` buffer when the producers of the tensor collect max |x| into work[0] (Plan.amax_of)
        self.h2_used = False        # has a conv asked for the planes?  (planes written by the producer itself: Plan.h2_direct)

    @property
    def M(self):
        return self.B * self.H * self.W

    def slice(self, c0, c):
        a = Act(self.t, self.ptr + c0 * self.es, self.B, self.H, self.W, c, self.ld, self.f32, self.es)
        a.root = self.root
        return a


class HeadRec:
    """per-head pieces of a plan (activations, units, the argument lists patched per call)"""


class ConvUnit:
    __slots__ = ("conv", "bn", "x", "y", "z", "relu", "res", "w", "wt", "scale", "shift", "mean", "invstd",
                 "Cp", "drop", "apply_args", "gscale_slots", "dz", "dy", "mask", "frozen", "dtype", "up")


class Plan:
    def __init__(self, engine: "Engine", B: int, H: int, W: int, dtype: torch.dtype, training: bool):
        self.e = engine
        self.lib = engine.lib
        self.B, self.H, self.W = B, H, W
        self.dtype, self.dt = dtype, _dt(dtype)
        self.es = 2 if dtype == torch.bfloat16 else 4
        self.vec = 16 // self.es
        self.training = training
        self.device = engine.store.flat_p.device
        self.fwd: list = []
        self.bwd: list = []
        self.prep: list = []           # weight preparation (re-run when the masters change)
        self.keep: list = []           # keeps ctypes structs / tensors alive
        self.units: List[ConvUnit] = []
        self.momentum_slots = []       # (args_list, index, bn)
        self.seed_slots = []           # (args_list, index)
        self.prepped_version = None
        self.param_last_op = {}        # param index -> index of the last backward op that adds to its gradient
        self.prep_table = None
        self.side = {}                 # backward op index -> True if it must first wait for the main stream
        self._side_events = None
        self._nat, self._ev_handles, self._ev_objs = {}, None, None
        self.drop_units = []
        self.sync = bool(engine.sync_bn and training and dist.is_available() and dist.is_initialized()
                         and dist.get_world_size(engine.sync_group) > 1)
        self.world = dist.get_world_size(engine.sync_group) if self.sync else 1
        self.last_dgrad = {}           # gradient buffer address -> ConvDesc of the data gradient that wrote it last
        self.res_src = {}              # id(block input) -> (dz of the block output, its ReLU mask): identity-branch gradient
        #                                that conv1's data gradient adds in its epilogue (DmlConvDesc.res_*)
        # fp32 plans: products of the forward / data-gradient convolutions on the bf16 matrix cores through a three-term
        # split of both operands (DmlConvDesc.f32_split; fp32-level error, not the exact fp32 MFMA): Engine.f32_split
        # (2: two fp16 planes per operand, written by dml_h2_split before the first conv that reads a tensor: Plan.h2_of)
        self.f32_split = int(engine.f32_split) if dtype == torch.float32 else 0
        self.prep_h2 = []              # dml_h2_split argument lists of the weight copies, run after every dml_prep_weights
        self.prep_h2_table = None      # (device array of DmlH2Desc, count) built from them: dml_h2_split_table
        # `work` buffers of dml_h2_split (1025 floats per tensor), carved from one allocation so that ONE fill at the head of
        # the forward zeroes every amax word the step's producers will raise (dml_bn_apply / dml_bn_bwd_apply, `amax`)
        self.h2_slots = torch.zeros(1025 * 1024, dtype=torch.float32, device=self.device) if self.f32_split == 2 else None
        self.h2_used = 0
        # f16x2: a batch-statistics BatchNorm writes the fp16 planes of its output (forward) / of dy (backward) itself, scaled
        # from a bound known beforehand (dml_h2_bound_bn): no dml_h2_split pass over those tensors, and no fp32 copy at all where
        # only convolutions read them.  DML_H2_DIRECT=0: every tensor through dml_h2_split (A/B, tests)
        self.h2_direct_on = os.environ.get("DML_H2_DIRECT", "1") != "0"
        self.direct_planes = []        # (activation, argument list, index of its `planes` argument): cleared when nobody asks
        self.fuse_res_grad = os.environ.get("DML_FUSE_RES_GRAD", "1") != "0"
        # f16x2 training: a block output that only convolutions and the NEXT block's residual add read exists as planes only -- the
        # residual add takes (hi + lo) / s, the value those convolutions see (dml_bn_apply, res_unscale).  DML_RES_PLANES=0: off
        self.res_planes_on = os.environ.get("DML_RES_PLANES", "1") != "0"
        self.fuse_bn_reduce = os.environ.get("DML_FUSE_BN_REDUCE", "1") != "0"
        # f16x2: the plane-scale bound of a BatchNorm output / of dy computed by the finalize launch's last block instead of a
        # one-block launch of its own behind it (dml_bn_finalize_bound / dml_bn_bwd_finalize_bound).  DML_FUSE_BOUND=0: off (A/B)
        self.fuse_bound = os.environ.get("DML_FUSE_BOUND", "1") != "0"
        # f16x2: data gradients of the stride-2 convolutions as stride-1 launches per pixel-parity class (conv_dgrad_s2_classes).
        # DML_S2_CLASSES=0: off (A/B, tests)
        self.s2_classes_on = os.environ.get("DML_S2_CLASSES", "1") != "0"
        self.prep_overlap_on = os.environ.get("DML_PREP_OVERLAP", "1") != "0"      # refresh_weights (A/B)
        self.ds_grad_from_dz = os.environ.get("DML_DS_GRAD_FROM_DZ", "1") != "0"      # block_bwd (A/B)
        self.prep_gather = []          # dml_gather_taps argument lists: sub-filters of the transposed weight copies, refreshed with them
        self._bound_words, self._bound_used = None, 0
        # bf16 plans, DML_GRAD_STAGE32=1: a gradient with several producers is summed in fp32 and rounded ONCE by its last
        # producer, as autograd does in the reference (resnet.py:112-113, network/utils.py:360).  Off by default: measured
        # over 10 + 4 seeds (tests/tools/bf16_noise_seeds.py, profiles/r03_bf16_noise_seeds.txt) the per-tensor gradient
        # noise is the same to three digits with a rounding after every producer (median 1 - cos 0.0788 vs 0.0789 at
        # 4 x 256^2, 0.0995 vs 0.0995 at 2 x 768^2; emulation 0.077 / 0.097), while the fp32 staging tensors move 3 GB more
        # per step (41.2 -> 42.0 ms).  The excess noise round 2 attributed to these roundings came from the bf16 rounding
        # of the image-pooling branch's B x C tensors (_head_fwd).
        self.stage32 = training and dtype == torch.bfloat16 and os.environ.get("DML_GRAD_STAGE32", "0") == "1"
        # bf16 weight copies of the LDS-DMA layers in the tile-major layout (DmlPrepDesc.w_tiled): the weight half of every DMA
        # instruction becomes one contiguous KB (whole cache lines) instead of sixteen 64-byte row segments.  DML_W_TILED=0: off
        self.tiled_weights = os.environ.get("DML_W_TILED", "1") != "0" and os.environ.get("DML_CONV_V1") is None
        self.fork_branches = os.environ.get("DML_FORK_BRANCHES", "1") != "0"
        # DmlConvDesc.ws_min_tiles for every conv of the plan: 0 = the library's rule for its wave-specialised kernel, 1 = take it
        # whenever the shape allows (the GPU tests run the small fixtures through it), 2147483647 = never
        self.ws_min_tiles = int(os.environ.get("DML_WS_MIN_TILES", "0"))
        # shared scratch for BN partial statistics (forward: ceil(M/rows)*N*2 floats, rows = 64 or 48 (dml_conv_stat_rows):
        # <= 2/3 B*H*W for every layer of this network; backward: <= ~1100*N*2)
        self.scratch = torch.empty(max(B * H * W * 3 // 4 + 16384, 1100 * 2048 * 2 + 65536), dtype=torch.float32,
                                   device=self.device)
        self.sp = self.scratch.data_ptr()
        # split-K partials of the weight gradients (one launch at a time uses it: all of them run on one stream)
        self.wgrad_ws = torch.empty(64 * (1 << 20) if training else 1, dtype=torch.float32, device=self.device)
        # K-split of partially filled last rounds in the main-stream convolutions (DmlConvDesc.tail_*): one workspace and
        # one counter array for the whole plan (the launches that use them are serialised on the caller's stream)
        self.tail_ws = torch.empty(512 * 128 * 128 if training else 1, dtype=torch.float32, device=self.device)
        self.tail_cnt = torch.zeros(128, dtype=torch.int32, device=self.device)
        self.group_wgrad = os.environ.get("DML_GROUP_WGRAD", "1") != "0"
        # 256 x 256 output tiles per grouped launch.  17 = one layer3 bottleneck (4 + 9 + 4): its three weight gradients
        # share a launch right after its backward, while their operands are still cache-hot.  Larger groups move fewer
        # slabs but start later: whole step 17: 371.0, 34: 369.9, 48: 369.8, ungrouped 370.3 images/s (interleaved runs)
        self.group_tiles = int(os.environ.get("DML_GROUP_TILES", "17"))
        self._wg_pending = []
        self.build()
        self.bytes = sum(t.numel() * t.element_size() for t in self.keep if isinstance(t, torch.Tensor))

    # ---- a unit may keep its tensors in fp32 inside a bf16 plan (the ASPP image-pooling branch, see _head_fwd)
    class _Precision:
        def __init__(self, plan, dtype):
            self.plan, self.dtype = plan, dtype

        def __enter__(self):
            p = self.plan
            self.saved = (p.dtype, p.dt, p.es, p.vec)
            p.dtype, p.dt = self.dtype, _dt(self.dtype)
            p.es = 2 if self.dtype == torch.bfloat16 else 4
            p.vec = 16 // p.es
            return p

        def __exit__(self, *exc):
            p = self.plan
            p.dtype, p.dt, p.es, p.vec = self.saved
            return False

    def precision(self, dtype):
        return Plan._Precision(self, dtype)

    # ---- allocation helpers
    def new(self, B, H, W, C, f32=False, ld=None, zero=False) -> Act:
        ld = ld or C
        dtype = torch.float32 if f32 else self.dtype
        alloc = torch.zeros if zero else torch.empty
        t = alloc(B * H * W * ld, dtype=dtype, device=self.device)
        self.keep.append(t)
        return Act(t, t.data_ptr(), B, H, W, C, ld, f32 or self.dtype == torch.float32, t.element_size())

    def dbuf(self, n):
        t = torch.zeros(max(int(n), 1), dtype=torch.float64, device=self.device)
        self.keep.append(t)
        return t

    def py_op(self, ops, fn):
        """A host-side step of the plan (a collective): called as fn(stream), runs on the caller's current stream."""
        def op(stream):
            fn()
            return 0
        ops.append((op, BoundArgs()))

    def fbuf(self, n, zero=False):
        t = (torch.zeros if zero else torch.empty)(max(int(n), 1), dtype=torch.float32, device=self.device)
        self.keep.append(t)
        return t

    def grad_of(self, a: Act) -> Act:
        """Gradient buffer of an activation (slices share their concat buffer's gradient)."""
        root = a.root
        if root.grad is None:
            root.grad = self.new(root.B, root.H, root.W, root.C)
        if a is root:
            return root.grad
        c0 = (a.ptr - root.ptr) // a.es
        g = root.grad.slice(c0, a.C)
        return g

    def stage_grad32(self, a: Act):
        """Declare d(loss)/d(a) a gradient with several producers (bf16 plans): the producers accumulate into an fp32
        staging tensor and the last one (conv_dgrad(final=True), or a conversion) rounds the total once into the bf16
        gradient the consumers read.  Rounding after every producer -- what a bf16 accumulate does -- costs about twice
        the noise power: d(out) collects the pooled term and four ASPP data gradients, and the BatchNorm backward that
        differences it amplifies whatever rounding noise it carries (profiles/r02_bf16_noise_by_depth.txt)."""
        if not self.stage32 or a is not a.root or a.g32 is not None or a.C % 8 or a.C <= 32 or a.f32:
            return
        a.g32 = self.new(a.B, a.H, a.W, a.C, f32=True)

    def call(self, ops, fn, *args):
        lst = BoundArgs(args)
        ops.append((fn, lst))
        return lst

    # ---- fp32 plans with fp32_products = "f16x2": fp16 hi / lo planes of a conv operand (DmlConvDesc.x_planes)
    def h2_ok(self, C, N, taps):
        """shapes the planes kernel takes (conv_ws_planes_eligible); the others run the three-term split on the fp32 tensors"""
        return self.f32_split == 2 and self.dtype == torch.float32 and C % 32 == 0 and N % 64 == 0 and taps <= 32

    @staticmethod
    def planes_fit(M, ld):
        """both fp16 planes of an [M][ld] tensor within the 31-bit byte offsets of the planes kernels (launch_conv / dml_conv_wgrad
        fall back to the fp32 tensors beyond): a tensor may exist as planes ONLY when this holds"""
        return 4 * M * ld < (1 << 31) - 4096

    def bound_state(self):
        """two zeroed 32-bit words (running maximum, ticket) of one dml_bn_*finalize_bound launch: left zero by every call"""
        if self._bound_words is None:
            self._bound_words = torch.zeros(2 * 1024, dtype=torch.int32, device=self.device)
            self.keep.append(self._bound_words)
        assert self._bound_used < 1024, "out of bound-state words"
        p = self._bound_words.data_ptr() + 8 * self._bound_used
        self._bound_used += 1
        return p

    def h2_work(self):
        assert self.h2_used < 1024, "out of dml_h2_split work buffers"
        w = self.h2_slots[self.h2_used * 1025:(self.h2_used + 1) * 1025]
        self.h2_used += 1
        return w

    def amax_of(self, a: Act):
        """pointer for the `amax` argument of the kernel that writes activation `a` (None outside f16x2 plans): every
        producer of a root tensor raises the same word, and the split then needs no maximum pass of its own"""
        if self.f32_split != 2 or self.dtype != torch.float32 or not a.f32 or a is not a.root:
            return None         # (a slice of a concat buffer: its other producers -- resize, broadcast -- do not report a maximum)
        root = a.root
        if root.amax is None:
            root.amax = self.h2_work()
        return root.amax.data_ptr()

    def h2_of(self, a: Act, ops):
        """(planes pointer, plane stride in elements, pointer to 1 / scale) of activation `a`; the split of its ROOT tensor is
        appended to `ops` the first time a conv asks -- every tensor of a plan is complete before its first consumer and never
        rewritten within a step, so one split per tensor and step serves all its consumers (forward conv, weight gradient)."""
        root = a.root
        root.h2_used = True
        if root.h2 is None:
            planes = torch.empty(2 * root.M * root.ld, dtype=torch.float16, device=self.device)
            known = root.amax is not None           # its producers collected max |x| (amax_of)
            work = root.amax if known else self.h2_work()
            self.keep.append(planes)
            self.call(ops, self.lib.dml_h2_split, root.ptr, root.M, root.C, root.ld, planes.data_ptr(), root.M * root.ld, root.ld,
                      0, work.data_ptr(), 1 if known else 0)
            root.h2 = (planes, work)
        planes, work = root.h2
        return planes.data_ptr() + (a.ptr - root.ptr) // 2, root.M * root.ld, work.data_ptr() + 4096

    def h2_direct(self, B, H, W, C, fp32_too: bool, work=None) -> Act:
        """a new fp32 activation whose producer (dml_bn_apply / dml_bn_bwd_apply) writes the fp16 planes itself; without
        `fp32_too` the fp32 tensor does not exist -- the Act then points at the planes (same extent: 2 x 2 bytes per element)
        and only planes consumers may read it"""
        planes = torch.empty(2 * B * H * W * C, dtype=torch.float16, device=self.device)
        self.keep.append(planes)
        a = self.new(B, H, W, C) if fp32_too else Act(planes, planes.data_ptr(), B, H, W, C, C, True, 4)
        a.h2 = (planes, work if work is not None else self.h2_work())
        return a

    def h2_weight(self, w, rows, K):
        """planes of a prepared fp32 weight copy [rows][K] (tile-major), refreshed with the copies (refresh_weights)"""
        if getattr(w, "h2", None) is None:
            planes = torch.empty(2 * rows * K, dtype=torch.float16, device=self.device)
            work = torch.zeros(1025, dtype=torch.float32, device=self.device)
            self.keep += [planes, work]
            self.prep_h2.append((w.data_ptr(), rows, K, K, planes.data_ptr(), rows * K, K, 1, work.data_ptr(), 0))
            self.prepped_version = None
            w.h2 = (planes, work)
        planes, work = w.h2
        return planes.data_ptr(), rows * K, work.data_ptr() + 4096

    def set_planes(self, dsc, x: Act, w, rows, K, ops):
        """DmlConvDesc.f32_split / x_planes / w_planes of a conv of this plan reading activation x and weight copy w[rows][K]"""
        dsc.f32_split = min(self.f32_split, 1)
        if self.h2_ok(x.C, rows, dsc.R * dsc.S):
            dsc.f32_split = 2
            dsc.x_planes, dsc.x_plane_stride, dsc.x_unscale = self.h2_of(x, ops)
            dsc.w_planes, dsc.w_plane_stride, dsc.w_unscale = self.h2_weight(w, rows, K)

    # ---- graph pieces ------------------------------------------------------------------------
    def conv_geom(self, conv: nn.Conv2d, x: Act):
        kh, kw = conv.kernel_size
        s, d, p = conv.stride[0], conv.dilation[0], conv.padding[0]
        Ho = (x.H + 2 * p - d * (kh - 1) - 1) // s + 1
        Wo = (x.W + 2 * p - d * (kw - 1) - 1) // s + 1
        if isinstance(conv, _S2DConv):
            Ho, Wo = conv.out_hw                  # (the zero taps of the regrouped filter reach one row / column further)
        return kh, kw, s, d, p, Ho, Wo

    def prep_weight(self, conv: nn.Conv2d, Cp: int, need_wt: bool, src_ptr=None, N=None, x_bytes=0, dy_bytes=0):
        """compute copies w[N][RS][Cp] (and wt[Cp][RS][N] for the data gradient) of a master weight; `src_ptr` / `N`
        override the source and the row count (the final conv's rows are padded to a multiple of 8, see build()).
        `x_bytes` / `dy_bytes`: extents of the forward / data-gradient operand tensors of this conv."""
        N, Cm = N or conv.out_channels, conv.in_channels
        kh, kw = conv.kernel_size
        if isinstance(conv, _S2DConv):
            assert Cp == Cm and self.dtype == torch.float32 and not need_wt
            w = torch.empty(N * kh * kw * Cm, dtype=torch.float32, device=self.device)
            w.tiled = False
            self.keep.append(w)
            self.call(self.fwd, self.lib.dml_s2d_weights, conv.weight.data_ptr(), w.data_ptr(), N, conv.k, conv.real.in_channels)
            return w, None
        w = torch.empty(N * kh * kw * Cp, dtype=self.dtype, device=self.device)
        wt = torch.empty(N * kh * kw * Cp, dtype=self.dtype, device=self.device) if need_wt else None
        # tile-major copies for the LDS-DMA kernels (DmlConvDesc.w_tiled): bf16, the GEMM's K a multiple of 32 per filter tap and
        # its row count a multiple of 64 -- the shapes those kernels take; conv_fwd / conv_dgrad pass the flag on.  Those kernels
        # address their operands with 31-bit byte offsets (launch_conv's `small` test): a conv whose activation or weight tensor
        # reaches 2 GiB falls back to the register-staged kernel, which reads the plain layout only -- no tile-major copy then.
        lim = 1 << 31
        dma = (self.tiled_weights and self.dtype == torch.bfloat16 and kh * kw <= 32 and N * kh * kw * Cp * 2 < lim)
        w.tiled = bool(dma and Cp % 32 == 0 and N % 64 == 0 and x_bytes < lim)
        if wt is not None:
            wt.tiled = bool(dma and N % 32 == 0 and Cp % 64 == 0 and dy_bytes < lim)
        self.keep += [w, wt]
        self.prep.append((src_ptr or conv.weight.data_ptr(), w.data_ptr(), wt.data_ptr() if wt is not None else 0, N,
                          kh * kw, Cm, Cp, self.dt, 1 if w.tiled else 0, 1 if (wt is not None and wt.tiled) else 0))
        return w, wt

    def conv_fwd(self, x: Act, conv: nn.Conv2d, y: Act, w, stats_ptr, bias_ptr=None, post=None, N=None):
        kh, kw, s, d, p, Ho, Wo = self.conv_geom(conv, x)
        assert (Ho, Wo) == (y.H, y.W), ((Ho, Wo), (y.H, y.W))
        dsc = ConvDesc(x=x.ptr, w=w.data_ptr(), y=y.ptr, bias=bias_ptr, stats=stats_ptr, B=x.B, Hi=x.H, Wi=x.W, C=x.C, ldx=x.ld, Ho=Ho, Wo=Wo,
                       N=N or conv.out_channels, ldy=y.ld, R=kh, S=kw, stride=s, dil=d, pad=p, dtype=self.dt,
                       y_f32=1 if (y.f32 and self.dtype != torch.float32) else 0, accum=0, mode=0)
        self.set_planes(dsc, x, w, dsc.N, kh * kw * x.C, self.fwd)
        dsc.w_tiled = 1 if getattr(w, "tiled", False) else 0
        dsc.ws_min_tiles = self.ws_min_tiles
        if self.training:
            dsc.tail_ws, dsc.tail_ws_elems = self.tail_ws.data_ptr(), self.tail_ws.numel()
            dsc.tail_counters, dsc.tail_counters_len = self.tail_cnt.data_ptr(), self.tail_cnt.numel()
        if post is not None:            # inference epilogue: BN(running stats) + residual + ReLU (DmlConvDesc.post_*)
            scale, shift, mean, res, relu = post
            dsc.post_scale, dsc.post_shift, dsc.post_mean = scale.data_ptr(), shift.data_ptr(), mean.data_ptr()
            dsc.post_res, dsc.post_ldres = (res.ptr, res.ld) if res is not None else (None, 0)
            dsc.post_relu = 1 if relu else 0
        self.keep.append(dsc)
        self.call(self.fwd, self.lib.dml_conv_igemm, C.byref(dsc))
        return dsc

    def conv_dgrad(self, dy: Act, conv: nn.Conv2d, wt, x: Act, final=True):
        """d(loss)/dx (+)= conv^T(dy); x.grad is created on demand.  `final`: no producer of this gradient comes after
        this one (only looked at for staged gradients, stage_grad32)."""
        gx = self.grad_of(x)
        kh, kw, s, d, p, _, _ = self.conv_geom(conv, x)
        if self.s2_classes_ok(dy, conv, x, kh, kw, s, d, p):
            return self.conv_dgrad_s2_classes(dy, conv, wt, x, gx, kh, kw, p)
        dsc = ConvDesc(x=dy.ptr, w=wt.data_ptr(), y=gx.ptr, bias=None, stats=None, B=dy.B, Hi=dy.H, Wi=dy.W, C=dy.C, ldx=dy.ld, Ho=x.H, Wo=x.W, N=x.C,
                       ldy=gx.ld, R=kh, S=kw, stride=s, dil=d, pad=p, dtype=self.dt, y_f32=0,
                       accum=1 if x.root.grad_init else 0, mode=1)
        self.set_planes(dsc, dy, wt, x.C, kh * kw * dy.C, self.bwd)
        dsc.w_tiled = 1 if getattr(wt, "tiled", False) else 0
        dsc.ws_min_tiles = self.ws_min_tiles
        g32 = x.g32 if x is x.root else None
        convert = False
        if g32 is not None and not (final and not x.root.grad_init):
            if final and dy.C % 32 == 0 and kh * kw <= 32:
                # last producer: conv + fp32 sum of the others, rounded once (DmlConvDesc.acc32, LDS-DMA kernels)
                dsc.accum, dsc.acc32, dsc.acc32_ld = 0, g32.ptr, g32.ld
            else:
                dsc.y, dsc.ldy, dsc.y_f32 = g32.ptr, g32.ld, 1
                convert = final
        else:
            g32 = None
        res = self.res_src.pop(id(x), None)
        if res is not None:
            # the identity branch's share of this gradient (block output gradient x ReLU mask) is added in the epilogue
            assert not x.root.grad_init and x is x.root and g32 is None
            rdz, rmask = res
            dsc.res_dz, dsc.res_mask, dsc.res_ld = rdz.ptr, rmask.data_ptr(), rdz.ld
        x.root.grad_init = True
        dsc.tail_ws, dsc.tail_ws_elems = self.tail_ws.data_ptr(), self.tail_ws.numel()
        dsc.tail_counters, dsc.tail_counters_len = self.tail_cnt.data_ptr(), self.tail_cnt.numel()
        self.keep.append(dsc)
        self.call(self.bwd, self.lib.dml_conv_igemm, C.byref(dsc))
        if convert:
            self.round_staged(x)
        self.last_dgrad.pop(self.grad_of(x.root).ptr, None)
        if x is x.root and g32 is None:
            self.last_dgrad[gx.ptr] = dsc         # whole-tensor gradient: candidate for the fused BN-backward reduce

    # ---- data gradient of a stride-2 convolution as one stride-1 launch per pixel-parity class (DmlConvDesc.sub_grid)
    def s2_classes_ok(self, dy, conv, x, kh, kw, s, d, p):
        """f16x2 plans, 3x3 / pad 1 and 1x1 / pad 0 at stride 2 on even maps, shapes the two-plane kernel takes.  A pixel of dX only sees
        the taps of its own row / column parity: the 3x3 becomes four launches of 1, 2, 2 and 4 taps (nine tap products per four pixels
        instead of 36, most of them on zeros), the 1x1 one launch on the even pixels -- which requires that the gradient already holds
        its other producers' sum (it only ADDS there)."""
        if not (self.s2_classes_on and s == 2 and d == 1 and kh == kw and (kh, p) in ((3, 1), (1, 0))
                and self.f32_split == 2 and self.dtype == torch.float32 and not isinstance(conv, _S2DConv)):
            return False
        if x is not x.root or x.g32 is not None or x.H != 2 * dy.H or x.W != 2 * dy.W or dy.C % 32 or x.C % 64:
            return False
        if kh == 1 and not x.root.grad_init:
            return False
        return self.planes_fit(dy.M, dy.ld) and 16 * dy.M * x.C < (1 << 31) - 4096

    def conv_dgrad_s2_classes(self, dy, conv, wt, x, gx, kh, kw, p):
        lib = self.lib
        accum = 1 if x.root.grad_init else 0
        xpl = self.h2_of(dy, self.bwd)
        descs = []
        for cy in (0, 1):
            for cx in (0, 1):
                rs = [r for r in range(kh) if (r - cy - p) % 2 == 0]
                ss = [t for t in range(kw) if (t - cx - p) % 2 == 0]
                if not rs or not ss:
                    continue                         # (1x1: only the even pixels receive anything; the gradient is accumulated into)
                taps = [r * kw + t for r in rs for t in ss]
                # the class's operand: rows = dX channels, K = its taps x dY channels, from the transposed copy wt[C][R S][N]
                sub = self.fbuf(x.C * len(taps) * dy.C)
                self.prep_gather.append((wt.data_ptr(), sub.data_ptr(), x.C, kh * kw, dy.C, len(taps), *(taps + [0] * (4 - len(taps)))))
                self.prepped_version = None
                dsc = ConvDesc(x=dy.ptr, w=sub.data_ptr(), y=gx.ptr, bias=None, stats=None, B=dy.B, Hi=dy.H, Wi=dy.W, C=dy.C, ldx=dy.ld,
                               Ho=dy.H, Wo=dy.W, N=x.C, ldy=gx.ld, R=len(rs), S=len(ss), stride=1, dil=1, pad=(cy + p - rs[0]) // 2,
                               dtype=self.dt, y_f32=0, accum=accum, mode=1)
                dsc.pad_w_set, dsc.pad_w = 1, (cx + p - ss[0]) // 2
                dsc.sub_grid, dsc.sub_y, dsc.sub_x = 1, cy, cx
                dsc.f32_split = 2
                dsc.x_planes, dsc.x_plane_stride, dsc.x_unscale = xpl
                dsc.w_planes, dsc.w_plane_stride, dsc.w_unscale = self.h2_weight(sub, x.C, len(taps) * dy.C)
                dsc.ws_min_tiles = self.ws_min_tiles
                dsc.tail_ws, dsc.tail_ws_elems = self.tail_ws.data_ptr(), self.tail_ws.numel()
                dsc.tail_counters, dsc.tail_counters_len = self.tail_cnt.data_ptr(), self.tail_cnt.numel()
                self.keep.append(dsc)
                self.call(self.bwd, lib.dml_conv_igemm, C.byref(dsc))
                descs.append(dsc)
        x.root.grad_init = True
        prev = self.last_dgrad.pop(gx.ptr, None)
        if len(descs) == 4:
            self.last_dgrad[gx.ptr] = descs           # the four classes cover the tensor: candidates for the fused BN-backward sums
        elif (len(descs) == 1 and accum and prev is not None and not isinstance(prev, list) and not prev.sub_grid
              and os.environ.get("DML_BNR_INC", "1") != "0"
              and prev.y == gx.ptr and prev.N == x.C and prev.ldy == gx.ld and prev.f32_split == 2 and not prev.res_dz):
            # 1x1: the class launch only visits the even pixels, but the BatchNorm-backward sums are linear in the gradient -- the
            # producer before it (a whole-tensor data gradient) emits the sums of the total stored SO FAR over all pixels, this launch those
            # of its increment over its own (DmlConvDesc.bnr_inc); unit_bwd hands both their partial groups
            descs[0].bnr_inc = 1
            self.last_dgrad[gx.ptr] = [prev, descs[0]]

    def round_staged(self, x: Act):
        """fp32 staging tensor -> the bf16 gradient (a staged gradient whose last producer cannot do it itself)"""
        g = self.grad_of(x)
        assert x is x.root and g.ld == g.C and x.g32.ld == g.C
        self.call(self.bwd, self.lib.dml_convert_dtype, x.g32.ptr, g.ptr, g.M * g.C, DML_F32, DML_BF16)

    def conv_wgrad(self, x: Act, dy: Act, conv: nn.Conv2d, Cp: int, pad_rows: int = 0):
        """pad_rows: dy carries `pad_rows` >= out_channels channels (zero beyond); the padded rows of the gradient
        go to a scratch tensor and only the real ones are added to the parameter's gradient."""
        kh, kw, s, d, p, Ho, Wo = self.conv_geom(conv, x)
        Cm = conv.in_channels
        gptr = self.e.store.grad_ptr_of(conv.weight)
        h2 = None
        if self.f32_split == 2 and self.dtype == torch.float32 and x.C % 8 == 0 and (pad_rows or conv.out_channels) % 8 == 0:
            # both operands as fp16 planes (the forward already split x; dy is split once for this and the data gradient).  The
            # splits are appended HERE, before `first`: they belong to the main stream, whose data gradient reads dy's planes too
            h2 = self.h2_of(x, self.bwd) + self.h2_of(dy, self.bwd)
        first = len(self.bwd)
        Nw, tmp = conv.out_channels, None
        s2d = isinstance(conv, _S2DConv)
        if s2d:
            assert not pad_rows
            tmp = self.fbuf(Nw * kh * kw * Cm)
            self.call(self.bwd, self.lib.dml_fill_f32, tmp.data_ptr(), tmp.numel(), 0.0)
        if pad_rows and pad_rows != conv.out_channels:
            Nw = pad_rows
            tmp = self.fbuf(Nw * kh * kw * Cm)
            self.call(self.bwd, self.lib.dml_fill_f32, tmp.data_ptr(), tmp.numel(), 0.0)
        dsc = WgradDesc(x=x.ptr, dy=dy.ptr, dw=tmp.data_ptr() if tmp is not None else gptr, B=x.B, Hi=x.H, Wi=x.W, C=x.C,
                        ldx=x.ld, Ho=Ho, Wo=Wo,
                        N=Nw, ldy=dy.ld, R=kh, S=kw, stride=s, dil=d, pad=p, dtype=self.dt, splitk=0,
                        Cm=Cm, ws=self.wgrad_ws.data_ptr(), ws_elems=self.wgrad_ws.numel(),
                        f32_split=min(self.f32_split, 1) if self.dtype == torch.float32 else 0)
        if h2 is not None:
            dsc.f32_split = 2
            (dsc.x_planes, dsc.x_plane_stride, dsc.x_unscale, dsc.dy_planes, dsc.dy_plane_stride, dsc.dy_unscale) = h2
        self.keep.append(dsc)
        if tmp is None and self.group_wgrad and self.lib.dml_conv_wgrad_group_eligible(C.byref(dsc)):
            # joins the next grouped launch (dml_conv_wgrad_group): the weight gradients of a few consecutive layers
            # share one round of workgroups, so each needs a handful of split-K slabs instead of ~28
            Ktot = kh * kw * x.C
            base = (Nw // 256) * ((Ktot + 255) // 256)
            self._wg_pending.append((dsc, conv, base))
            if sum(b for _, _, b in self._wg_pending) >= self.group_tiles or len(self._wg_pending) >= 12:
                self.flush_wgrad()
            return
        self.call(self.bwd, self.lib.dml_conv_wgrad, C.byref(dsc))
        if s2d:
            self.call(self.bwd, self.lib.dml_s2d_wgrad, tmp.data_ptr(), gptr, conv.out_channels, conv.k, conv.real.in_channels)
        elif tmp is not None:
            self.call(self.bwd, self.lib.dml_unpad_wgrad, tmp.data_ptr(), gptr, conv.out_channels, kh * kw, Cm, Cm)
        # weight gradients only feed the optimizer: they run on a side stream, next to the HBM-bound BN backward
        # and the data gradient of the following layers (Plan.run_backward)
        for i in range(first, len(self.bwd)):
            self.side[i] = (i == first)
        self.mark_grad(conv.weight)

    def flush_wgrad(self):
        """emit the pending weight-gradient jobs as one grouped launch on the side stream"""
        pend, self._wg_pending = self._wg_pending, []
        if not pend:
            return
        arr = (C.c_void_p * len(pend))(*[C.addressof(d) for d, _, _ in pend])
        self.keep.append(arr)
        i = len(self.bwd)
        args = self.call(self.bwd, self.lib.dml_conv_wgrad_group, C.addressof(arr), len(pend), self.wgrad_ws.data_ptr(),
                         self.wgrad_ws.numel())
        args.meta = [d for d, _, _ in pend]                 # bench.py: FLOPs / bytes of the launch
        self.side[i] = True
        for _, conv, _ in pend:
            self.param_last_op[self.e.store._index(conv.weight)] = i

    def mark_grad(self, p):
        self.param_last_op[self.e.store._index(p)] = len(self.bwd) - 1

    def cbr(self, x: Act, conv: nn.Conv2d, bn: nn.BatchNorm2d, relu=True, res: Optional[Act] = None,
            out: Optional[Act] = None, drop: Optional[nn.Dropout] = None, need_dgrad=True, planes_only=False) -> ConvUnit:
        """conv -> BN(batch or running stats) -> (+res) -> (ReLU) -> (dropout); z may be a concat slice.
        planes_only: only planes consumers read z (the convolutions of an f16x2 plan) -- no fp32 z where the BN writes planes."""
        lib, st = self.lib, self.e.store
        u = ConvUnit()
        u.conv, u.bn, u.x, u.relu, u.res, u.drop = conv, bn, x, relu, res, drop
        u.dtype = self.dtype
        u.Cp = x.C                       # x already carries any channel padding
        kh, kw, s, d, p, Ho, Wo = self.conv_geom(conv, x)
        N = conv.out_channels
        u.w, u.wt = self.prep_weight(conv, u.Cp, self.training and need_dgrad, x_bytes=x.M * x.ld * x.es,
                                     dy_bytes=x.B * Ho * Wo * N * x.es)
        # f16x2 training: this BN writes z's fp16 planes itself (Plan.h2_direct) -- batch statistics bound the output
        direct = (self.h2_direct_on and self.training and bn.training and self.f32_split == 2 and self.dtype == torch.float32
                  and out is None and N % 8 == 0 and drop is None
                  and (res is None or (res is res.root and res.amax is not None)))
        planes_only = planes_only and direct and self.planes_fit(x.B * Ho * Wo, N)
        if direct:
            u.z = self.h2_direct(x.B, Ho, Wo, N, fp32_too=not planes_only)
            if not planes_only or res is not None:
                # max |z| is published where something may ask for it: a conv reading the fp32 tensor through dml_h2_split, or the
                # next block's residual unit (its plane scale: dml_h2_bound_bn) -- also when z itself exists as planes only
                u.z.amax = u.z.h2[1]           # (the amax words and the scale word share the tensor's work buffer)
        else:
            u.z = out if out is not None else self.new(x.B, Ho, Wo, N)
        u.y = self.new(x.B, Ho, Wo, N) if self.training else None      # inference never materialises it
        M = u.z.M
        u.scale, u.shift = self.fbuf(N), self.fbuf(N)
        g_ptr, b_ptr = bn.weight.data_ptr(), bn.bias.data_ptr()
        mean_ptr = bn.running_mean.data_ptr()
        u.frozen = self.training and not bn.training
        if u.frozen:
            # BatchNorm2d.eval() inside a training step (main_self_distillation.py:432-435 of the reference): normalise
            # with the running statistics, leave them alone; gamma / beta still train.  scale / shift / invstd come from
            # the coefficient table launch at the head of the plan (two rows: the second with gamma = beta = None
            # yields 1/sqrt(var + eps) itself), the backward is the batch-statistics one without its correction terms.
            u.mean, u.invstd = bn.running_mean, self.fbuf(N)
            dummy = self.fbuf(N)
            self.bn_eval.append(_lib.BnEvalDesc(g_ptr, b_ptr, bn.running_var.data_ptr(), u.scale.data_ptr(),
                                                u.shift.data_ptr(), N, float(bn.eps)))
            self.bn_eval.append(_lib.BnEvalDesc(None, None, bn.running_var.data_ptr(), u.invstd.data_ptr(),
                                                dummy.data_ptr(), N, float(bn.eps)))
            self.conv_fwd(x, conv, u.y, u.w, None)
        elif self.training:
            u.mean, u.invstd = self.fbuf(N), self.fbuf(N)
            mean_ptr = u.mean.data_ptr()
            bound_done = False
            dsc = self.conv_fwd(x, conv, u.y, u.w, self.sp)
            # rows per statistics partial of THIS launch (48 on the wave-specialised kernel, STAT_ROWS otherwise)
            rows = lib.dml_conv_stat_rows(C.byref(dsc))
            groups = (M + rows - 1) // rows
            assert groups * N * 2 <= self.scratch.numel(), "BN statistics scratch too small"
            if self.sync:
                # statistics over every rank's samples (equal shards): local (mean, M2) -> all_gather -> Chan merge
                mom, allmom = self.dbuf(N * 2), self.dbuf(self.world * N * 2)
                self.call(self.fwd, lib.dml_bn_moments, self.sp, M, N, rows, mom.data_ptr())
                grp = self.e.sync_group
                self.py_op(self.fwd, lambda a=allmom, b=mom, g=grp: dist.all_gather_into_tensor(a, b, group=g))
                args = self.call(self.fwd, lib.dml_bn_finalize_moments, allmom.data_ptr(), self.world, M, N, g_ptr, b_ptr,
                                 bn.running_mean.data_ptr(), bn.running_var.data_ptr(), 0.1, float(bn.eps),
                                 u.scale.data_ptr(), u.shift.data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr())
                self.momentum_slots.append((args, 8, bn))
            elif direct and self.fuse_bound and not (res is None and self.h2_bound_args is not None):
                # f16x2: the plane scale of z from its bound (dml_h2_bound_bn) comes out of the SAME launch -- the finalize block
                # that arrives last writes it (dml_bn_finalize_bound: one dependent one-block launch less per residual unit)
                args = self.call(self.fwd, lib.dml_bn_finalize_bound, self.sp, M, N, rows, g_ptr, b_ptr,
                                 bn.running_mean.data_ptr(), bn.running_var.data_ptr(), 0.1, float(bn.eps),
                                 u.scale.data_ptr(), u.shift.data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr(),
                                 M * self.world, 1.0, res.amax.data_ptr() if res is not None else None,
                                 u.z.h2[1].data_ptr(), self.bound_state())
                self.momentum_slots.append((args, 8, bn))
                bound_done = True
            else:
                args = self.call(self.fwd, lib.dml_bn_finalize, self.sp, M, N, rows, g_ptr, b_ptr,
                                 bn.running_mean.data_ptr(), bn.running_var.data_ptr(), 0.1, float(bn.eps),
                                 u.scale.data_ptr(), u.shift.data_ptr(), u.mean.data_ptr(), u.invstd.data_ptr())
                self.momentum_slots.append((args, 8, bn))
        else:
            # inference: scale / shift of every BN come from ONE table launch at the head of the plan, and BN +
            # residual + ReLU run in the conv epilogue -- one launch per unit instead of three, no y tensor
            self.bn_eval.append(_lib.BnEvalDesc(g_ptr, b_ptr, bn.running_var.data_ptr(), u.scale.data_ptr(),
                                                u.shift.data_ptr(), N, float(bn.eps)))
            self.conv_fwd(x, conv, u.z, u.w, None, post=(u.scale, u.shift, bn.running_mean, res, relu))
            u.gscale_slots, u.mask, u.apply_args = [], None, None
            self.units.append(u)
            return u
        u.gscale_slots = []
        # ReLU bitmask (bf16 training): the two BN backward passes read 1 byte per 8 elements instead of z
        u.mask = None
        if self.training and relu and N % self.vec == 0:
            # one byte per 16-byte vector of z: 8 bits in bf16 plans, 4 in fp32 plans (the BN backward reads it instead of z)
            u.mask = torch.empty(M * (N // self.vec), dtype=torch.uint8, device=self.device)
            self.keep.append(u.mask)
        mask_ptr = u.mask.data_ptr() if u.mask is not None else None
        pl = (None, 0, 0, None)
        if direct:
            planes, work = u.z.h2
            if bound_done:
                pass
            elif res is None and self.h2_bound_args is not None:
                self.h2_bound_tab.append(_lib.H2BoundDesc(g_ptr, b_ptr, work.data_ptr(), N,
                                                          float(np.float32(np.sqrt(np.float32(M * self.world))) * np.float32(1.0001)), 1.0, 0))
            else:
                self.call(self.fwd, lib.dml_h2_bound_bn, g_ptr, b_ptr, N, M * self.world, 1.0,
                          res.amax.data_ptr() if res is not None else None, work.data_ptr())
            pl = (planes.data_ptr(), M * N, N, work.data_ptr() + 4096)
        only = direct and planes_only
        if out is not None and out.root.t.dtype == torch.float16:
            # `out` is a channel slice of a tensor that exists as fp16 planes ONLY (the decoder's concat buffer of an f16x2 training
            # plan, _head_fwd): this BatchNorm writes its slice of the planes, scaled by the tensor's one scale (dml_h2_bound_bn_multi)
            assert self.training and bn.training and drop is None and self.dtype == torch.float32 and N % 8 == 0
            pp, pstride, punscale = self.h2_of(out, self.fwd)
            pl = (pp, pstride, out.ld, punscale)
            only = True
        # a residual operand that exists as planes only (the previous block's output, block_fwd): hi + lo, unscaled
        res_pl = (0, None)
        if res is not None and res.t.dtype == torch.float16:
            assert res is res.root and res.h2 is not None and self.dtype == torch.float32
            res_pl = (res.M * res.ld, res.h2[1].data_ptr() + 4096)
        u.apply_args = self.call(self.fwd, lib.dml_bn_apply, u.y.ptr, res.ptr if res is not None else None,
                                 None if only else u.z.ptr, u.scale.data_ptr(), u.shift.data_ptr(), mean_ptr, mask_ptr, M, N,
                                 u.y.ld, res.ld if res is not None else 0, u.z.ld, 1 if relu else 0, self.dt, 0.0, 0,
                                 self.amax_of(u.z) if (self.training and (not only or u.z.amax is not None)) else None, *pl,
                                 *res_pl)
        if direct and not only:
            self.direct_planes.append((u.z, u.apply_args, 17))
        if drop is not None and self.training:
            self.drop_units.append(u)
        self.units.append(u)
        return u

    def unit_bwd(self, u: ConvUnit, dz: Act, dres: Optional[Act] = None, dres_accum=False, need_dgrad=True, final=True,
                 up_mask=None):
        """Backward of `cbr`: BN (two passes) -> weight gradient -> data gradient into u.x.grad (`final`: conv_dgrad).
        up_mask: `dz` is the gradient of a LATER ReLU's output and `up_mask` that ReLU's 1-bit mask -- this unit's own output gradient
        is dz (.) mask, formed on the fly by the two BN passes (block_bwd: the downsample branch reads the block output's gradient)."""
        if u.dtype != self.dtype:
            with self.precision(u.dtype):
                return self.unit_bwd(u, dz, dres, dres_accum, need_dgrad, final, up_mask)
        lib, st = self.lib, self.e.store
        N, M = u.conv.out_channels, u.y.M
        bn = u.bn
        # f16x2: the BN backward writes dy's fp16 planes itself, scaled from a bound (dml_h2_bound_bn_bwd); dy exists in fp32
        # only if one of its two consumers (weight gradient, data gradient) cannot read planes
        kh, kw = u.conv.kernel_size
        dy_direct = self.h2_direct_on and self.f32_split == 2 and self.dtype == torch.float32 and N % 8 == 0
        gwork = None
        if dy_direct:
            only = (u.x.C % 8 == 0 and self.planes_fit(M, N) and self.planes_fit(u.x.root.M, u.x.root.ld)
                    and (not need_dgrad or self.h2_ok(N, u.x.C, kh * kw)))
            dy = self.h2_direct(u.y.B, u.y.H, u.y.W, N, fp32_too=not only)
            gwork = self.h2_work()
        else:
            only = False
            dy = self.new(u.y.B, u.y.H, u.y.W, N)
        u.dz, u.dy = dz, dy
        u.up = None                    # (block_bwd sets it: the unit whose ReLU mask gates `dz` for this unit, unit_bwd's up_mask)
        coef = self.fbuf(4 * N)
        nblk = C.c_int(0)
        self.keep.append(nblk)
        mk = u.mask.data_ptr() if u.mask is not None else None
        relu_eff = u.relu
        if up_mask is not None:
            assert not u.relu and u.drop is None and dres is None
            mk, relu_eff = up_mask.data_ptr(), True
        # The data gradient that wrote dz last can emit this BN's backward sums from its epilogue (DmlConvDesc.bnr_*):
        # one pass over dz / y / mask less.  Conditions: it wrote the whole tensor, bf16 with the 1-bit ReLU mask,
        # no dropout scale, at most 4096 row groups (on the 192 x 192 layers the finalize would fold 9216 groups in two
        # stages, which works, but the fused sums then cost the data gradients more than the stand-alone reduce:
        # 363.5 vs 364.5 images/s).
        prod = self.last_dgrad.get(dz.ptr) if (self.fuse_bn_reduce and u.z is u.z.root and up_mask is None) else None
        # (a stride-2 data gradient issued as four parity-class launches: each writes the sums of its own rows, conv_dgrad_s2_classes)
        prods = prod if isinstance(prod, list) else ([prod] if prod is not None else [])
        prod = prods[0] if prods else None
        prows = lib.dml_conv_stat_rows(C.byref(prod)) if prod is not None else STAT_ROWS      # rows per partial of that launch
        Gs = [(pd.B * pd.Ho * pd.Wo + prows - 1) // prows for pd in prods] if len(prods) > 1 else [(M + prows - 1) // prows]
        G = sum(Gs)
        fused = (prod is not None and len(prods) == 1 and self.dtype == torch.bfloat16 and (u.mask is not None or not u.relu)
                 and u.drop is None and N % 8 == 0 and N > 32 and G <= 4096 and prod.N == N and prod.ldy == N
                 and dz.ld == N and prod.y == dz.ptr)
        # fp32 tensors: the two-plane kernel's epilogue does the same (4-channel mask bytes, 48-row groups, max |g| for dy's bound).
        # Up to 16384 groups since round 5 (the 192 x 192 layers: 12288 groups, folded in two stages by dml_bn_bwd_finalize): with the
        # row epilogue's vector mask loads the fused sums win there too, 83.90 -> 83.60 ms (profiles/r05_ab_bnr_maxg.txt)
        fused = fused or (prod is not None and self.dtype == torch.float32 and prod.f32_split == 2 and prows == 48
                          and (u.mask is not None or not u.relu) and u.drop is None and N % 64 == 0
                          and G <= 16384
                          and prod.N == N and prod.ldy == N and dz.ld == N and prod.y == dz.ptr and u.y.ld % 4 == 0)
        a1 = None
        bound_done = False
        if fused:
            part = self.fbuf(G * N * 2)
            g0 = 0
            for pd, gc in zip(prods, Gs):
                pd.bnr_y, pd.bnr_mask = u.y.ptr, mk
                pd.bnr_mean, pd.bnr_invstd = u.mean.data_ptr(), u.invstd.data_ptr()
                pd.bnr_partials, pd.bnr_ldy, pd.bnr_relu = part.data_ptr() + g0 * N * 8, u.y.ld, 1 if u.relu else 0
                if gwork is not None:
                    pd.bnr_gmax = gwork.data_ptr()
                g0 += gc
            nblk = C.c_int(G)
            self.keep.append(nblk)
            sp = part.data_ptr()
        else:
            sp = self.sp
            a1 = self.call(self.bwd, lib.dml_bn_bwd_reduce, dz.ptr, u.y.ptr, u.z.ptr, mk, u.mean.data_ptr(),
                           u.invstd.data_ptr(), self.sp, M, N, dz.ld, u.y.ld, u.z.ld, 1 if relu_eff else 0, 1.0,
                           self.dt, C.byref(nblk), gwork.data_ptr() if gwork is not None else None)
        if self.sync and not u.frozen:
            sums = self.dbuf(N * 2)
            self.call(self.bwd, lib.dml_bn_bwd_sums, sp, nblk, N, sums.data_ptr(), st.grad_ptr_of(bn.weight),
                      st.grad_ptr_of(bn.bias))
            grp = self.e.sync_group
            self.py_op(self.bwd, lambda t=sums, g=grp: dist.all_reduce(t, group=g))
            self.call(self.bwd, lib.dml_bn_bwd_coef, sums.data_ptr(), M * self.world, N, bn.weight.data_ptr(),
                      u.mean.data_ptr(), u.invstd.data_ptr(), coef.data_ptr())
        elif dy_direct and self.fuse_bound:
            # (f16x2: dy's plane scale -- dml_h2_bound_bn_bwd -- from the finalize launch itself, its last block)
            self.call(self.bwd, lib.dml_bn_bwd_finalize_bound, sp, nblk, 0 if u.frozen else M, N, bn.weight.data_ptr(),
                      u.mean.data_ptr(), u.invstd.data_ptr(), st.grad_ptr_of(bn.weight),
                      st.grad_ptr_of(bn.bias), coef.data_ptr(), M * self.world, gwork.data_ptr(), dy.h2[1].data_ptr(),
                      self.bound_state())
            bound_done = True
        else:
            self.call(self.bwd, lib.dml_bn_bwd_finalize, sp, nblk, 0 if u.frozen else M, N, bn.weight.data_ptr(),
                      u.mean.data_ptr(), u.invstd.data_ptr(), st.grad_ptr_of(bn.weight),
                      st.grad_ptr_of(bn.bias), coef.data_ptr())
        self.mark_grad(bn.weight)
        self.mark_grad(bn.bias)
        pl = (None, 0, 0, None)
        if dy_direct:
            planes, work = dy.h2
            if not bound_done:
                self.call(self.bwd, lib.dml_h2_bound_bn_bwd, coef.data_ptr(), u.invstd.data_ptr(), N, M * self.world,
                          gwork.data_ptr(), work.data_ptr())
            pl = (planes.data_ptr(), M * N, N, work.data_ptr() + 4096)
        a3 = self.call(self.bwd, lib.dml_bn_bwd_apply, dz.ptr, u.y.ptr, u.z.ptr, mk, coef.data_ptr(),
                       None if only else dy.ptr, dres.ptr if dres is not None else None, M, N, dz.ld, u.y.ld, u.z.ld, dy.ld,
                       dres.ld if dres is not None else 0, 1 if relu_eff else 0, 1.0,
                       1 if dres_accum else 0, self.dt, None if dy_direct else self.amax_of(dy), *pl)
        u.gscale_slots += ([(a1, 13)] if a1 is not None else []) + [(a3, 15)]
        if dres is not None:
            self.last_dgrad.pop(dres.ptr, None)   # written by the BN kernel, not by a data gradient
        self.conv_wgrad(u.x, dy, u.conv, u.Cp)
        if need_dgrad:
            self.conv_dgrad(dy, u.conv, u.wt, u.x, final=final)

    def block_fwd(self, x: Act, blk: nn.Module, out_planes_only=False):
        """one Bottleneck (resnet.py:95-115): 1x1 -> 3x3 -> 1x1, + identity or downsample branch, ReLU.
        out_planes_only: the block output is read by planes convolutions and the next block's residual add only (build())."""
        # (f16x2: u1.z / u2.z feed one convolution each -- where that one reads planes, the fp32 tensors are never written)
        n1, n2, n3 = blk.conv1.out_channels, blk.conv2.out_channels, blk.conv3.out_channels
        kk = blk.conv2.kernel_size[0] * blk.conv2.kernel_size[1]
        u1 = self.cbr(x, blk.conv1, blk.bn1, planes_only=self.h2_ok(n1, n2, kk))
        u2 = self.cbr(u1.z, blk.conv2, blk.bn2, planes_only=self.h2_ok(n2, n3, 1))
        ud = None
        if blk.downsample is not None:
            ud = self.cbr(x, blk.downsample[0], blk.downsample[1], relu=False)
            idt = ud.z
        else:
            idt = x
        u3 = self.cbr(u2.z, blk.conv3, blk.bn3, relu=True, res=idt, planes_only=out_planes_only)
        return (x, u1, u2, u3, ud)

    def block_bwd(self, rec):
        """backward of block_fwd: consumes d(block output), produces d(block input)"""
        xb, u1, u2, u3, ud = rec
        dz = self.grad_of(u3.z)
        ud_from_dz = False
        if ud is not None:
            # The downsample branch's output gradient is the block output's gradient under the block's ReLU mask.  Instead of having
            # bn3's backward apply write that copy (4 bytes per element of the block output: 1.4 GB per step over the four blocks) the
            # branch's BN backward reads dz and the mask itself (unit_bwd, up_mask)
            ud_from_dz = (self.ds_grad_from_dz and u3.relu and u3.mask is not None and u3.drop is None and not ud.relu
                          and ud.drop is None and ud.z is ud.z.root and ud.conv.out_channels == u3.conv.out_channels)
            if ud_from_dz:
                self.unit_bwd(u3, dz, dres=None)
            else:
                dres = self.grad_of(ud.z)
                self.unit_bwd(u3, dz, dres=dres, dres_accum=False)
                ud.z.grad_init = True
        elif (self.fuse_res_grad and not xb.root.grad_init and xb is xb.root
              and u3.relu and u3.mask is not None and u3.drop is None and xb.C % 8 == 0 and xb.C > 32
              and dz.ld == xb.C and self.conv_geom(u1.conv, xb)[:3] == (1, 1, 1)
              and (self.dtype == torch.bfloat16
                   # f16x2: conv1's data gradient on the two-plane kernel takes res_* too.  (Round 4 measured this as a LOSS, 183.5 ->
                   # 179.0 images/s: the row epilogue fetched its ReLU masks with one byte load per lane and four rows, a third of
                   # its memory instructions.  With the masks of a sub-tile in one 16-byte load per row, conv_epilogue_rows, it wins:
                   # 187.6 -> 190.1 images/s, three interleaved rounds, profiles/r05_ab_maskvec_res.txt; without the fusion that
                   # change alone is +0.3 %.)  DML_FUSE_RES_GRAD=0: off
                   or (self.h2_direct_on and self.h2_ok(u1.conv.out_channels, xb.C, 1) and xb.C % 64 == 0
                       and self.planes_fit(u1.y.M, u1.conv.out_channels)))):
            # The masked output gradient is this block's contribution to d(xb) through the identity branch.  Instead
            # of having the BN-backward apply write that copy (75 MB per layer3 block) for conv1's data gradient to
            # accumulate onto, conv1's data gradient reads dz and the mask itself (DmlConvDesc.res_*).
            self.unit_bwd(u3, dz, dres=None)
            self.res_src[id(xb)] = (dz, u3.mask)
        else:
            dres = self.grad_of(xb)
            self.unit_bwd(u3, dz, dres=dres, dres_accum=xb.root.grad_init)
            xb.root.grad_init = True
        self.unit_bwd(u2, self.grad_of(u2.z))
        self.unit_bwd(u1, self.grad_of(u1.z), final=ud is None)
        assert not self.res_src, "conv1's data gradient did not take the identity-branch gradient"
        if ud is not None:
            if ud_from_dz:
                self.unit_bwd(ud, dz, up_mask=u3.mask)
                ud.up = u3
            else:
                self.unit_bwd(ud, self.grad_of(ud.z))

    # ---- the network -------------------------------------------------------------------------
    def build(self):
        lib, m, st = self.lib, self.e.model, self.e.store
        B, H, W = self.B, self.H, self.W
        bb, head_modules = m.backbone, self.e.head_modules()
        self.pre_prep = []
        n_fixed = 0
        for mod in itertools.chain(bb.modules(), *[h.modules() for h in head_modules]):
            if isinstance(mod, nn.BatchNorm2d):
                if mod.training and not self.training:
                    raise NotImplementedError("a BatchNorm2d in train() mode inside an eval() model is not supported")
                n_fixed += 0 if mod.training else 1

        self.bn_eval = []
        self.h2_bound_tab, self.h2_bound_args = [], None
        if self.h2_slots is not None and self.training:
            # every amax word of the step starts from zero (Plan.amax_of; the backward's words are raised after the forward)
            self.call(self.fwd, lib.dml_fill_f32, self.h2_slots.data_ptr(), self.h2_slots.numel(), 0.0)
            # the plane scales of every residual-free BatchNorm output (they depend on gamma / beta / count only): one launch,
            # table filled in at the end of the forward build
            self.h2_bound_args = self.call(self.fwd, lib.dml_h2_bound_bn_table, 0, 0)
        if n_fixed:
            self.bn_eval_args = self.call(self.fwd, lib.dml_bn_eval_coeffs_table, 0, 0)      # filled in below
        # input packing NCHW fp32 -> NHWC (8 ch); f16x2 plans: space-to-depth, [B][H/2][W/2][12], and the stem conv in that form
        # (_S2DConv).  The same products in another summation order: the exact-fp32 and three-term modes keep the 7x7 stride-2 form
        # on 8 channels so that their results stay what they were (DML_STEM_S2D=1: every fp32 plan, =0: none)
        stem_conv = bb.conv1
        s2d_env = os.environ.get("DML_STEM_S2D", "")
        s2d_on = s2d_env == "1" or (s2d_env != "0" and self.f32_split == 2)
        if self.dtype == torch.float32 and s2d_on and _S2DConv.fits(bb.conv1, H, W):
            x_in = self.new(B, H // 2, W // 2, 4 * bb.conv1.in_channels)
            self.images_args = self.call(self.fwd, lib.dml_pack_input_s2d, 0, x_in.ptr, B, bb.conv1.in_channels, H, W)
            stem_conv = _S2DConv(bb.conv1, H // 2, W // 2)
        else:
            x_in = self.new(B, H, W, _PAD_CIN)
            self.images_args = self.call(self.fwd, lib.dml_pack_input, 0, x_in.ptr, B, 3, H, W, _PAD_CIN, self.dt)

        # stem: 7x7 s2 conv + BN + ReLU, 3x3 s2 max pool (resnet.py:139-143,196-199)
        stem = self.cbr(x_in, stem_conv, bb.bn1, need_dgrad=False)
        z0 = stem.z
        Hp, Wp = (z0.H - 1) // 2 + 1, (z0.W - 1) // 2 + 1
        p0 = self.new(B, Hp, Wp, 64)
        amax = torch.empty(B * Hp * Wp * 64, dtype=torch.uint8, device=self.device) if self.training else None
        self.keep.append(amax)
        self.call(self.fwd, lib.dml_maxpool3x3s2_fwd, z0.ptr, p0.ptr, amax.data_ptr() if amax is not None else None,
                  B, z0.H, z0.W, 64, self.dt)

        # bottlenecks (resnet.py:95-115)
        blocks = []
        x = p0
        seq = [(li, blk) for li, layer in enumerate((bb.layer1, bb.layer2, bb.layer3, bb.layer4)) for blk in layer]
        self.prep_cut = None
        for i, (li, blk) in enumerate(seq):
            if li == 1 and self.prep_cut is None:
                # everything from layer2 on reads weight copies that refresh_weights prepares on the side stream, under the stem and
                # layer1 (Plan.refresh_weights, overlap): ops before this index only need the copies made so far
                self.prep_cut = (len(self.fwd), len(self.prep), len(self.prep_h2))
            # the output as planes only: another block follows whose conv1 / downsample conv take planes, and it is neither `low`
            # (the decoder's 48-channel projection reads the fp32 tensor) nor `out` (global average pooling does)
            nxt = seq[i + 1] if i + 1 < len(seq) else None
            c3 = blk.conv3.out_channels
            po = (self.res_planes_on and nxt is not None and not (li == 0 and nxt[0] != 0)
                  and nxt[1].conv1.kernel_size == (1, 1) and self.h2_ok(c3, nxt[1].conv1.out_channels, 1)
                  and (nxt[1].downsample is None or self.h2_ok(c3, nxt[1].downsample[0].out_channels, 1)))
            blocks.append(self.block_fwd(x, blk, out_planes_only=po))
            x = blocks[-1][3].z
            if li == 0 and (nxt is None or nxt[0] != 0):
                low = x
        out = x

        # heads: one for the DMLNet model, several (shared backbone) for the self-distillation model (utils.py:120-193)
        self.heads = [self._head_fwd(h, low, out) for h in head_modules]
        if self.h2_bound_args is not None and self.h2_bound_tab:
            arr = (_lib.H2BoundDesc * len(self.h2_bound_tab))(*self.h2_bound_tab)
            self.h2_bound_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            self.h2_bound_args[0], self.h2_bound_args[1] = self.h2_bound_table.data_ptr(), len(self.h2_bound_tab)
        self.K, self.Kp = self.heads[0].K, self.heads[0].Kp
        self.nbt_inc = None
        if self.training and n_fixed:
            self.nbt_inc = torch.tensor([1 if mod.training else 0 for mod in st.bn_modules], dtype=torch.int64,
                                        device=self.device)
        if self.bn_eval:
            arr = (_lib.BnEvalDesc * len(self.bn_eval))(*self.bn_eval)
            self.bn_eval_table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.device)
            self.bn_eval_args[0], self.bn_eval_args[1] = self.bn_eval_table.data_ptr(), len(self.bn_eval)
        if not self.training:
            return

        # ------------------------------------------------------------------ backward
        # The LAST head goes first: it initialises d(out) / d(low); the others accumulate into them, so their
        # segments of the plan can be skipped when the loss does not reach them (Engine.backward).
        self.head_bwd_range = {}
        self.to_backbone_ops = []
        # gradients with several producers: d(out) (pooled term + four ASPP data gradients per head), d(low) (decoder
        # projection + layer2.0's conv1 and downsample), and the input of every other block with a downsample branch
        self.stage_grad32(out)
        self.stage_grad32(low)
        for (xb, u1, u2, u3, ud) in blocks:
            if ud is not None:
                self.stage_grad32(xb)
        for hi in reversed(range(len(self.heads))):
            start = len(self.bwd)
            self._head_bwd(self.heads[hi], low, out)
            self.flush_wgrad()                          # a head's segment may be skipped as a whole: keep its jobs inside
            self.head_bwd_range[hi] = (start, len(self.bwd))
        # bottlenecks in reverse (this whole segment, and the head ops that only feed it, are skipped when no backbone
        # parameter requires a gradient: the reference's incremental recipe trains one new head on a fixed trunk,
        # main_self_distillation.py:352-357)
        backbone_start = len(self.bwd)
        if out.g32 is not None and len(self.heads) > 1:
            self.round_staged(out)                      # several heads: which of them runs last is only known per step
        for rec in reversed(blocks):
            self.block_bwd(rec)
        # max pool + stem
        dz0 = self.grad_of(z0)
        self.call(self.bwd, lib.dml_maxpool3x3s2_bwd, self.grad_of(p0).ptr, amax.data_ptr(), dz0.ptr, B, z0.H, z0.W,
                  64, self.dt)
        self.unit_bwd(stem, dz0, need_dgrad=False)
        self.flush_wgrad()
        self.backbone_bwd_range = (backbone_start, len(self.bwd))
        for a, args, i in self.direct_planes:       # planes nobody reads (outputs that only feed resizes / concats): not written
            if not a.h2_used:
                args[i] = None

    def _head_fwd(self, head: nn.Module, low: Act, out: Act):
        """DeepLabHeadV3Plus + final upsample + distance head (network/utils.py:8-32,84-118) on the backbone features."""
        lib, st = self.lib, self.e.store
        B, H, W = self.B, self.H, self.W
        K = head.classifier[3].out_channels
        Kp = _round_up(K, 8)           # embedding channels are carried padded to 16-byte vectors (zeros beyond K)
        if Kp > 32:
            raise NotImplementedError("num_classes=%d: the distance-head kernels hold at most 32 embedding channels and 33 "
                                      "prototypes; the reference's drivers use 16 (main_embedding.py:336)" % K)
        cat2_c = _round_up(48 + 256, 32)                   # 304 -> 320: K tiles of 32 stay inside one tap
        # f16x2 training: the concat buffer exists as fp16 planes only.  Its two producers -- the low-level projection's BatchNorm and
        # the bilinear resize of the ASPP projection -- write their channel slices of the planes themselves, with ONE scale from the
        # larger of the two BatchNorms' bounds (known from gamma / beta / count alone: dml_h2_bound_bn_multi at the head of the
        # forward); the amax + split passes of dml_h2_split over this 755 MB tensor (16 x 768 x 768) are gone.  DML_CAT_PLANES=0: off
        cat_planes = (self.training and self.f32_split == 2 and self.dtype == torch.float32 and self.h2_direct_on
                      and os.environ.get("DML_CAT_PLANES", "1") != "0" and head.project[1].training and head.aspp.project[1].training
                      and not self.sync and self.planes_fit(B * low.H * low.W, cat2_c))
        if cat_planes:
            cat2 = self.h2_direct(B, low.H, low.W, cat2_c, fp32_too=False)
            cat2.h2[0].zero_()                              # (the 16 pad channels stay zero: nobody writes them)
            pdrop = float(head.aspp.project[3].p) if isinstance(head.aspp.project[3], nn.Dropout) else 0.0
            rc = lambda n: float(np.float32(np.sqrt(np.float32(n * self.world))) * np.float32(1.0001))
            tab = (_lib.H2BoundDesc * 2)(
                _lib.H2BoundDesc(head.project[1].weight.data_ptr(), head.project[1].bias.data_ptr(), None, 48, rc(B * low.H * low.W), 1.0, 0),
                _lib.H2BoundDesc(head.aspp.project[1].weight.data_ptr(), head.aspp.project[1].bias.data_ptr(), None, 256,
                                 rc(B * out.H * out.W), 1.0 / (1.0 - pdrop) if pdrop < 1.0 else 1.0, 0))
            self.cat_bound_table = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
            self.keep.append(self.cat_bound_table)
            self.call(self.fwd, lib.dml_h2_bound_bn_multi, self.cat_bound_table.data_ptr(), 2, cat2.h2[1].data_ptr())
        else:
            cat2 = self.new(B, low.H, low.W, cat2_c, zero=True)
        up_low = self.cbr(low, head.project[0], head.project[1], out=cat2.slice(0, 48))
        aspp = head.aspp
        cat1 = self.new(B, out.H, out.W, 5 * 256)
        branches = []
        # f16x2: the planes of `out` are shared by all four conv branches.  h2_of appends a tensor's split at its FIRST consumer,
        # which would be branch 0 -- inside the first fork range of an inference plan (run_forward: every range on its own
        # stream, each waiting for the main stream only), so branches 1-3 would read planes and the unscale word that
        # branch 0's stream is still writing.  Issue the split here, on the main stream, before the fork point.
        if self.h2_ok(out.C, aspp.convs[0][0].out_channels, 1):
            self.h2_of(out, self.fwd)
        marks = [len(self.fwd)]
        for i in range(4):
            branches.append(self.cbr(out, aspp.convs[i][0], aspp.convs[i][1], out=cat1.slice(256 * i, 256)))
            marks.append(len(self.fwd))
        # image-pooling branch (network/utils.py:318-329): avg-pool -> 1x1 -> BN -> ReLU -> broadcast
        if self.training and self.dtype == torch.bfloat16:
            # This BatchNorm sees B samples per channel.  When two of them are within bf16 resolution of each other the
            # rounded pre-normalisation values collapse, 1/sigma goes to 1/sqrt(eps) and the layer's backward -- whose two
            # correction terms cancel its output gradient almost exactly -- returns a spurious gradient hundreds of times
            # too large that the broadcast spreads over every pixel of d(out) (measured at 768 x 768, 2 images: layer4's
            # bn3 gradients 20x their true norm).  The unit is B x 256 values: it runs in fp32 storage, and so do its
            # input (the pooled features) and its output gradient -- B x C numbers whose sample-to-sample DIFFERENCES
            # decide 1/sigma and the cancellation (dml_reduce_hw_f32); only its output is rounded for the bf16 concat.
            with self.precision(torch.float32):
                pooled = self.new(B, 1, 1, out.C)
                self.call(self.fwd, lib.dml_reduce_hw_f32, out.ptr, pooled.ptr, B, out.H * out.W, out.C, out.ld, DML_BF16,
                          1.0 / (out.H * out.W))
                upool = self.cbr(pooled, aspp.convs[4][1], aspp.convs[4][2])
            zq = self.new(B, 1, 1, 256)
            self.call(self.fwd, lib.dml_convert_dtype, upool.z.ptr, zq.ptr, B * 256, DML_F32, DML_BF16)
            pool_z, pool_f32 = zq, True
        else:
            pooled = self.new(B, 1, 1, out.C)
            self.call(self.fwd, lib.dml_global_avgpool_fwd, out.ptr, pooled.ptr, B, out.H * out.W, out.C, out.ld, self.dt)
            upool = self.cbr(pooled, aspp.convs[4][1], aspp.convs[4][2])
            pool_z, pool_f32 = upool.z, False
        self.call(self.fwd, lib.dml_broadcast_hw, pool_z.ptr, cat1.slice(1024, 256).ptr, B, out.H * out.W, 256,
                  cat1.ld, self.dt)
        marks.append(len(self.fwd))
        if not self.training and self.fork_branches and (out.M + 127) // 128 * 2 < 300:
            # inference on small maps (a branch's grid below ~300 workgroups; 1024 x 2048 at batch 1: 270 vs 261
            # images/s, at batch 4 the grids fill the chip alone and forking costs 1.5 %): the five branches only read
            # `out` and write disjoint channel slices of cat1 (one launch each, the pooled one three) -- independent, so
            # they may overlap (training keeps them in line: its units share the statistics scratch)
            if not hasattr(self, "fwd_forks"):
                self.fwd_forks = []
            # nothing inside a fork range may write a tensor another range reads: the only shared input is `out` (split above)
            assert not any(fn is lib.dml_h2_split and args[0] == out.root.ptr for fn, args in self.fwd[marks[0]:marks[5]]), \
                "the split of the fork's shared input must run before the fork point"
            self.fwd_forks.append([(marks[k], marks[k + 1]) for k in range(5)])
        uproj = self.cbr(cat1, aspp.project[0], aspp.project[1], drop=aspp.project[3])
        up_slice = cat2.slice(48, 256)
        if cat_planes:
            pp, pstride, punscale = self.h2_of(up_slice, self.fwd)
            self.call(self.fwd, lib.dml_bilinear_fwd_planes, uproj.z.ptr, pp, pstride, cat2.ld, punscale, B, out.H, out.W, low.H, low.W,
                      256, uproj.z.ld)
        else:
            self.call(self.fwd, lib.dml_bilinear_fwd, uproj.z.ptr, up_slice.ptr, B, out.H, out.W, low.H, low.W, 256,
                      uproj.z.ld, cat2.ld, self.dt, 0, 0)
        ucls = self.cbr(cat2, head.classifier[0], head.classifier[1])
        fin = head.classifier[3]
        fin_bias_ptr = fin.bias.data_ptr() if fin.bias is not None else None
        if Kp == K:
            w_fin, wt_fin = self.prep_weight(fin, 256, self.training)
        else:
            # num_classes that is no multiple of 8 (the factory default is 21): the final conv computes Kp channels
            # from a zero-padded copy of its weight / bias, so the embedding's pad channels are exactly 0, the
            # prototypes are padded with zeros too, and every kernel keeps its 16-byte channel vectors
            wpad = self.fbuf(Kp * 256, zero=True)
            bpad = self.fbuf(Kp, zero=True)

            def stage(wpad=wpad, bpad=bpad, fin=fin, K=K):
                wpad[:K * 256].copy_(fin.weight.detach().reshape(-1))
                if fin.bias is not None:
                    bpad[:K].copy_(fin.bias.detach())
            self.pre_prep.append(stage)
            w_fin, wt_fin = self.prep_weight(fin, 256, self.training, src_ptr=wpad.data_ptr(), N=Kp)
            fin_bias_ptr = bpad.data_ptr() if fin.bias is not None else None
        emb = self.new(B, low.H, low.W, Kp, f32=True)
        self.conv_fwd(ucls.z, fin, emb, w_fin, None, bias_ptr=fin_bias_ptr, N=Kp)
        protos = self.e.prototypes_padded(K, Kp)
        # fused final upsample + distance head (network/utils.py:88-118); outputs are per-call tensors
        head_args = self.call(self.fwd, lib.dml_upsample_dist_fwd, emb.ptr, protos.data_ptr(), 0, 0,
                                   None, None, B, emb.H, emb.W, Kp, K, H, W)
        rec = HeadRec()
        rec.K, rec.Kp, rec.emb, rec.protos, rec.head_args = K, Kp, emb, protos, head_args
        rec.cat1, rec.cat2, rec.up_low, rec.branches, rec.pooled, rec.upool = cat1, cat2, up_low, branches, pooled, upool
        rec.pool_f32 = pool_f32
        rec.uproj, rec.ucls, rec.fin, rec.wt_fin = uproj, ucls, fin, wt_fin
        rec.head_bwd_args, rec.df, rec.feats_p = None, None, None
        return rec

    def _head_bwd(self, rec, low: Act, out: Act):
        lib, st = self.lib, self.e.store
        B, H, W = self.B, self.H, self.W
        K, Kp, emb, fin, wt_fin = rec.K, rec.Kp, rec.emb, rec.fin, rec.wt_fin
        cat1, cat2, up_low, branches, pooled, upool = rec.cat1, rec.cat2, rec.up_low, rec.branches, rec.pooled, rec.upool
        uproj, ucls = rec.uproj, rec.ucls
        df = self.fbuf(B * H * W * Kp)
        rec.df = df
        de = self.new(B, emb.H, emb.W, Kp)
        rec.de = de                     # gradient of the low-resolution embedding
        # Two ways from d(loss)/d(logits) to the low-resolution embedding gradient `de`; Engine.backward skips one.
        # (1) the loss handed over a deferred gradient (dmlnet/lazy_grad.py) and the shape is the fused kernel's (16
        #     prototypes, exact x4 upsample): loss gradient + distance gradient + transposed upsample in one pass;
        # (2) anything else: distance gradient into `df`, then the transposed upsample.
        rec.fused_args, rec.fused_range = None, None
        if Kp == 16 and K == 16 and H == 4 * emb.H and W == 4 * emb.W:
            i0 = len(self.bwd)
            rec.fused_args = self.call(self.bwd, lib.dml_head_bwd_fused, 0, 0, 0, 0, rec.protos.data_ptr(), de.ptr, B,
                                       emb.H, emb.W, Kp, K, H, W, 255, 0.0, 1.0, self.dt)
            rec.fused_range = (i0, i0 + 1)
        i1 = len(self.bwd)
        rec.head_bwd_args = self.call(self.bwd, lib.dml_proto_dist_bwd, 0, None, 0, rec.protos.data_ptr(),
                                       df.data_ptr(), B, Kp, K, H, W)
        self.call(self.bwd, lib.dml_bilinear_bwd, df.data_ptr(), de.ptr, B, emb.H, emb.W, H, W, Kp, Kp, Kp, self.dt, 1, 0)
        rec.unfused_range = (i1, i1 + 2)
        if fin.bias is not None:
            bws = self.fbuf(1024 * Kp)          # per-workgroup partial sums: a fixed-order (reproducible) bias gradient
            if Kp == K:
                self.call(self.bwd, lib.dml_bias_grad_ws, de.ptr, st.grad_ptr_of(fin.bias), de.M, K, de.ld, self.dt,
                          bws.data_ptr(), bws.numel())
            else:
                gb = self.fbuf(Kp)
                self.call(self.bwd, lib.dml_fill_f32, gb.data_ptr(), Kp, 0.0)
                self.call(self.bwd, lib.dml_bias_grad_ws, de.ptr, gb.data_ptr(), de.M, Kp, de.ld, self.dt, bws.data_ptr(),
                          bws.numel())
                self.call(self.bwd, lib.dml_unpad_wgrad, gb.data_ptr(), st.grad_ptr_of(fin.bias), 1, 1, K, Kp)
            self.mark_grad(fin.bias)
        self.conv_wgrad(ucls.z, de, fin, 256, pad_rows=Kp)
        self.conv_dgrad(de, fin, wt_fin, ucls.z)
        self.unit_bwd(ucls, self.grad_of(ucls.z))                       # -> d cat2
        dcat2 = self.grad_of(cat2)
        # upsample branch -> d(aspp output)
        dproj = self.grad_of(uproj.z)
        self.call(self.bwd, lib.dml_bilinear_bwd, dcat2.slice(48, 256).ptr, dproj.ptr, B, out.H, out.W, low.H, low.W,
                  256, dcat2.ld, dproj.ld, self.dt, 0, 0)
        uproj.z.grad_init = True
        self.unit_bwd(uproj, dproj)                                      # -> d cat1
        dcat1 = self.grad_of(cat1)
        # pooling branch
        staged = out.g32 is not None
        if rec.pool_f32:                                                 # the unit lives in fp32 storage (_head_fwd)
            with self.precision(torch.float32):
                dzp = self.new(B, 1, 1, 256)
                self.call(self.bwd, lib.dml_reduce_hw_f32, dcat1.slice(1024, 256).ptr, dzp.ptr, B, out.H * out.W, 256,
                          dcat1.ld, DML_BF16, 1.0)
                self.unit_bwd(upool, dzp)                                # -> d pooled (fp32)
                gp32 = self.grad_of(pooled)
            if not staged:
                gpq = self.new(B, 1, 1, out.C)
                self.call(self.bwd, lib.dml_convert_dtype, gp32.ptr, gpq.ptr, B * out.C, DML_F32, DML_BF16)
        else:
            assert not staged
            dzp = self.new(B, 1, 1, 256)
            self.call(self.bwd, lib.dml_reduce_hw, dcat1.slice(1024, 256).ptr, dzp.ptr, B, out.H * out.W, 256, dcat1.ld,
                      self.dt)
            self.unit_bwd(upool, dzp)                                    # -> d pooled
            gpq = self.grad_of(pooled)
        feeders = self.to_backbone_ops                 # backward ops whose only product is d(out) / d(low)
        # The image-pooling branch contributes dv / HW to every pixel of d(out): far below half an ulp of the other four
        # branches' sum in bf16.  It goes in FIRST (while the buffer is still untouched), so that the data gradients
        # accumulate onto it in fp32 before the rounding and the term survives on average; a later head's segment finds
        # the buffer initialised and adds.
        first = not out.root.grad_init
        if staged:              # fp32 staging tensor (stage_grad32): the last ASPP data gradient rounds the sum once
            self.call(self.bwd, lib.dml_avgpool_bwd_set if first else lib.dml_avgpool_bwd_add, gp32.ptr, out.g32.ptr, B,
                      out.H * out.W, out.C, out.g32.ld, DML_F32)
        else:
            self.call(self.bwd, lib.dml_avgpool_bwd_set if first else lib.dml_avgpool_bwd_add, gpq.ptr,
                      self.grad_of(out).ptr, B, out.H * out.W, out.C, self.grad_of(out).ld, self.dt)
        out.root.grad_init = True
        feeders.append(len(self.bwd) - 1)
        for i in range(4):
            self.unit_bwd(branches[i], dcat1.slice(256 * i, 256), final=(i == 3 and len(self.heads) == 1))   # -> d out
            feeders.append(len(self.bwd) - 1)           # unit_bwd ends with the data gradient
        self.last_dgrad.pop(self.grad_of(out).ptr, None)      # several writers: layer4's last BN keeps its own reduce
        # low-level projection -> d low (layer1 output)
        self.unit_bwd(up_low, dcat2.slice(0, 48), final=False)          # layer2.0's data gradients follow
        feeders.append(len(self.bwd) - 1)

    # ---- execution ---------------------------------------------------------------------------
    def refresh_weights(self, stream, overlap=False):
        """compute copies / fp16 planes of the weights after the masters changed.  overlap (training forward): the copies of layer2 and
        everything behind it are made on the engine's side stream while the stem and layer1 run (their own few copies first, on the
        caller's stream); run_forward waits for the side stream's event at Plan.prep_cut.  0.5 ms of launches that only depend on the
        optimizer step leave the forward's critical path."""
        st = self.e.store
        key = (st.version, sum(p._version for p in st.params))
        if key == self.prepped_version:
            return
        for fn in self.pre_prep:
            fn()
        self.prep_event = None
        if self.prep_table is None:
            self.prep_table = []
            for dt in sorted({e[7] for e in self.prep}):          # one table launch per storage type present in the plan
                ent = [e for e in self.prep if e[7] == dt]
                arr = (_lib.PrepDesc * len(ent))()
                for i, (src, w, wt, N, RS, Cm, Cp, _, w_tiled, wt_tiled) in enumerate(ent):
                    arr[i] = _lib.PrepDesc(src, w, wt or None, N, RS, Cm, Cp, w_tiled, wt_tiled)
                raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
                self.prep_table.append((raw.to(self.device), len(ent), dt))
        if self.prep_h2 and (self.prep_h2_table is None or self.prep_h2_table[1] != len(self.prep_h2)):
            # fp16 planes of every weight copy (f16x2): one table, two launches
            arr = (_lib.H2Desc * len(self.prep_h2))()
            for i, (x, rows, Cc, ld, planes, pstride, ldp, layout, work, _) in enumerate(self.prep_h2):
                arr[i] = _lib.H2Desc(x, planes, work, rows, pstride, Cc, ld, ldp, layout)
            raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
            self.prep_h2_table = (raw.to(self.device), len(self.prep_h2))
        cut = self.prep_cut if (overlap and self.training and self.prep_overlap_on and len(self.prep_table) == 1) else None
        if cut is not None and 0 < cut[1] < self.prep_table[0][1] and self.prep_h2 and 0 < cut[2] < len(self.prep_h2):
            # early part on the caller's stream, the rest on the side stream (which first picks up the caller's stream: the masters)
            tab, n, dt = self.prep_table[0]
            nA, hA = cut[1], cut[2]
            main, side = torch.cuda.current_stream(self.device), self.e.side_stream(self.device)
            side.wait_stream(main)
            ss = side.cuda_stream
            _lib.check(self.lib.dml_prep_weights(tab.data_ptr(), nA, dt, stream), "dml_prep_weights")
            _lib.check(self.lib.dml_h2_split_table(self.prep_h2_table[0].data_ptr(), hA, stream), "dml_h2_split_table")
            _lib.check(self.lib.dml_prep_weights(tab.data_ptr() + nA * C.sizeof(_lib.PrepDesc), n - nA, dt, ss), "dml_prep_weights")
            for g in self.prep_gather:
                _lib.check(self.lib.dml_gather_taps(*g, ss), "dml_gather_taps")
            _lib.check(self.lib.dml_h2_split_table(self.prep_h2_table[0].data_ptr() + hA * C.sizeof(_lib.H2Desc),
                                                   len(self.prep_h2) - hA, ss), "dml_h2_split_table")
            if getattr(self, "_prep_ev", None) is None:
                self._prep_ev = torch.cuda.Event()
            self._prep_ev.record(side)
            self.prep_event = self._prep_ev
        else:
            for tab, n, dt in self.prep_table:
                _lib.check(self.lib.dml_prep_weights(tab.data_ptr(), n, dt, stream), "dml_prep_weights")
            for g in self.prep_gather:
                _lib.check(self.lib.dml_gather_taps(*g, stream), "dml_gather_taps")
            if self.prep_h2:
                _lib.check(self.lib.dml_h2_split_table(self.prep_h2_table[0].data_ptr(), self.prep_h2_table[1], stream),
                           "dml_h2_split_table")
        self.prepped_version = key

    # ---- replay: one C call per contiguous run of ops (dml_plan_run) unless the engine is told to stay in Python
    def _native_list(self, ops):
        key = id(ops)
        nat = self._nat.get(key)
        if nat is None:
            nat = NativeList(self.lib, ops, self.side if ops is self.bwd else None)
            # an aborted replay may leave tickets of a K-split tail (DmlConvDesc.tail_counters) behind: the counters must be
            # zero when the next launch that uses them starts
            nat.on_error = self.tail_cnt.zero_
            self._nat[key] = nat
        return nat

    def _events(self):
        if self._ev_handles is None:
            cur = torch.cuda.current_stream(self.device)
            self._ev_objs = [torch.cuda.Event() for _ in range(16)]
            for e in self._ev_objs:
                e.record(cur)                       # instantiates the hipEvent_t
            self._ev_handles = (C.c_void_p * len(self._ev_objs))(*[e.cuda_event for e in self._ev_objs])
        return self._ev_handles

    def _mark_events(self, n):
        """n (main, side) event pairs for the deferred hooks of _exec, instantiated once"""
        evs = getattr(self, "_mark_ev", [])
        cur = torch.cuda.current_stream(self.device)
        while len(evs) < n:
            pair = (torch.cuda.Event(), torch.cuda.Event())
            for e in pair:
                e.record(cur)                       # instantiates the hipEvent_t
            evs.append(pair)
        self._mark_ev = evs
        return evs[:n]

    def _exec(self, ops, stream, start=0, stop=None, hook=None, skip_ranges=(), side_stream=None):
        """Issue ops[start:stop] in order.  Native segments run through dml_plan_run; Python steps (collectives), skipped
        ranges and the hook points (hook.points: op indices after which hook(i) must run -- the gradient buckets of the
        data-parallel reducer) cut the list into segments."""
        stop = len(ops) if stop is None else stop
        if not self.e.native or Plan.run is not Plan._py_run:
            skipped = set()
            for (lo, hi) in skip_ranges:
                skipped.update(range(lo, hi))
            if side_stream is not None:
                return self._py_backward_two_streams(ops, stream, side_stream, hook, skipped)
            return Plan.run(ops, stream, start, stop, hook=hook, skipped=skipped)
        nat = self._native_list(ops)
        points = getattr(hook, "points", None) if hook is not None else ()
        if hook is not None and points is None:
            points = range(start, stop)              # a hook without declared points wants every op
        # A DEFERRED hook (hook.deferred: the data-parallel reducer) only enqueues work on another stream that has to wait for this
        # point of the list: the native replay records an event pair (main / side stream) after each of its ops (dml_plan_run_marks)
        # and the hook runs once the whole list has been enqueued, hook(op, main event, side event) -- one return to Python per
        # backward instead of one per gradient bucket.
        deferred = hook is not None and getattr(hook, "deferred", False) and points is not None
        marks, mark_ops, mark_evs = None, [], []
        if deferred:
            mark_ops = sorted(k for k in points if start <= k < stop)
            mark_evs = self._mark_events(len(mark_ops))
            if mark_ops:
                arr = (C.c_int32 * len(mark_ops))(*mark_ops)
                hnd = (C.c_void_p * (2 * len(mark_ops)))(*[e.cuda_event for pair in mark_evs for e in pair])
                marks = (arr, len(mark_ops), hnd)
                self.keep_marks = (arr, hnd)
            points = ()
        cuts = set(nat.python_ops)
        cuts.update(points)
        skips = sorted((max(lo, start), min(hi, stop)) for (lo, hi) in skip_ranges if lo < stop and hi > start)
        events = self._events() if side_stream is not None else None
        n_ev = len(events) if events is not None else 0
        i = start
        order = sorted(c for c in cuts if start <= c < stop)
        ci, si = 0, 0
        while i < stop:
            while si < len(skips) and skips[si][1] <= i:
                si += 1
            if si < len(skips) and skips[si][0] <= i:      # inside a skipped range: only the hook points fire
                hi = skips[si][1]
                for k, (em, es) in zip(mark_ops, mark_evs):      # (deferred hooks: their events, recorded here)
                    if i <= k < hi:
                        em.record(torch.cuda.current_stream(self.device))
                        if side_stream is not None:
                            es.record(self.e.side_stream(self.device))
                if hook is not None:
                    for k in order:
                        if i <= k < hi and k in points:
                            hook(k)
                i = hi
                continue
            end = skips[si][0] if si < len(skips) else stop
            while ci < len(order) and order[ci] < i:
                ci += 1
            if ci < len(order) and order[ci] < end:
                k = order[ci]
                if k in nat.python_ops:
                    if k > i:
                        nat.run(i, k, stream, side_stream, events, n_ev, marks)
                    fn, args = ops[k]
                    fn(*args, stream)
                    if k in mark_ops:                # (a Python step that is a mark itself)
                        em, es = mark_evs[mark_ops.index(k)]
                        em.record(torch.cuda.current_stream(self.device))
                        if side_stream is not None:
                            es.record(self.e.side_stream(self.device))
                else:
                    nat.run(i, k + 1, stream, side_stream, events, n_ev, marks)
                if hook is not None and k in points:
                    hook(k)
                i = k + 1
            else:
                nat.run(i, end, stream, side_stream, events, n_ev, marks)
                i = end
        for k, (em, es) in zip(mark_ops, mark_evs):
            hook(k, em, es if side_stream is not None else None)

    def run_backward(self, hook=None):
        """Replay the backward plan: weight gradients on the engine's side stream, everything else on the
        caller's current stream; joined at the end."""
        main = torch.cuda.current_stream(self.device)
        skip = tuple(getattr(self, "skip_ranges", ()))
        if not self.e.overlap_wgrad or not self.side:
            self._exec(self.bwd, main.cuda_stream, hook=hook, skip_ranges=skip)
            return
        side = self.e.side_stream(self.device)
        side.wait_stream(main)
        self._exec(self.bwd, main.cuda_stream, hook=hook, skip_ranges=skip, side_stream=side.cuda_stream)
        main.wait_stream(side)

    def _py_backward_two_streams(self, ops, ms, ss, hook, skipped):
        """the Python replay of the two-stream backward (DML_NATIVE_PLAN=0, profiling)"""
        main = torch.cuda.current_stream(self.device)
        side = self.e.side_stream(self.device)
        if self._side_events is None:
            self._side_events = {i: torch.cuda.Event() for i, f in self.side.items() if f}
        sidemap, events = self.side, self._side_events
        for i, (fn, args) in enumerate(ops):
            if i in skipped:
                if hook is not None:
                    hook(i)
                continue
            flag = sidemap.get(i)
            if flag is None:
                rc = fn(*args, ms)
            else:
                if flag:
                    ev = events[i]
                    ev.record(main)
                    side.wait_event(ev)
                rc = fn(*args, ss)
            if rc:
                _lib.check(rc, getattr(fn, "__name__", "kernel") + " (bwd op %d)" % i)
            if hook is not None:
                hook(i)

    def run_forward(self, stream):
        """Replay the forward plan.  Inference plans carry fork groups (`fwd_forks`: the independent ASPP branches): their
        op ranges run side by side on the engine's branch streams and join on the caller's stream -- at batch 1 each
        branch is a grid of 128-256 workgroups, too small to fill the chip alone."""
        forks = getattr(self, "fwd_forks", None)
        if not forks:
            ev = getattr(self, "prep_event", None)
            if ev is not None and self.prep_cut is not None:
                # (refresh_weights, overlap: the weight copies of layer2.. are being made on the side stream)
                self._exec(self.fwd, stream, 0, self.prep_cut[0])
                torch.cuda.current_stream(self.device).wait_event(ev)
                self.prep_event = None
                self._exec(self.fwd, stream, self.prep_cut[0], len(self.fwd))
                return
            self._exec(self.fwd, stream)
            return
        main = torch.cuda.current_stream(self.device)
        pos = 0
        for ranges in forks:
            self._exec(self.fwd, stream, pos, ranges[0][0])
            streams = self.e.branch_streams(self.device, len(ranges))
            for (lo, hi), s in zip(ranges, streams):
                s.wait_stream(main)
                self._exec(self.fwd, s.cuda_stream, lo, hi)
            for s in streams[:len(ranges)]:
                main.wait_stream(s)
            pos = ranges[-1][1]
        self._exec(self.fwd, stream, pos, len(self.fwd))

    @staticmethod
    def run(ops, stream, start=0, stop=None, hook=None, skipped=()):
        stop = len(ops) if stop is None else stop
        for i in range(start, stop):
            if i in skipped:
                if hook is not None:
                    hook(i)
                continue
            fn, args = ops[i]
            rc = fn(*args, stream)
            if rc:
                _lib.check(rc, getattr(fn, "__name__", "kernel") + " (op %d)" % i)
            if hook is not None:
                hook(i)

    _py_run = run          # bench.py's profiled pass replaces Plan.run: the native path then steps aside


class Engine:
    """Owns the parameter store and the plan cache of one model instance."""

    plan_cls = Plan

    def __init__(self, model: nn.Module):
        self.model = model
        self.lib = _lib.load()
        self.store = ParamStore(model)
        self.plans = {}
        self._protos = {}
        self.reducer = None             # parallel.GradReducer, attached for multi-GPU runs
        self.overlap_wgrad = os.environ.get("DML_OVERLAP_WGRAD", "1") != "0"
        self.native = os.environ.get("DML_NATIVE_PLAN", "1") != "0"      # replay launch lists through dml_plan_run
        self.sync_bn, self.sync_group = False, None     # synchronised BatchNorm statistics over the process group
        # fp32 compute dtype only: "bf16x3" products (model.set_compute_dtype(torch.float32, fp32_products="bf16x3"))
        # or "f16x2" (fp32_products="f16x2"): two fp16 planes per operand, three MFMAs per block (DmlConvDesc.x_planes)
        self.f32_split = {"exact": 0, "bf16x3": 1, "f16x2": 2}[os.environ.get("DML_F32_PRODUCTS", "exact").lower()]
        self._side = {}
        self._branch = {}
        self.step_count = 0
        self.seed = 0x5DEECE66D

    def rank_seed(self) -> int:
        """dropout seed of this rank: the reference's DataParallel replicas draw independent masks per sample
        (network/utils.py:354 under main_embedding.py:425), so the data-parallel ranks must not share one"""
        rank = dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0
        return (self.seed ^ (rank * 0xD1B54A32D192ED03)) & 0xFFFFFFFFFFFFFFFF

    def side_stream(self, device):
        if device not in self._side:
            # (stream priorities were tried: this runtime offers {normal, high} only, and neither a high-priority main
            # stream nor a high-priority side stream changed the step time -- tools/ab_bench.sh)
            self._side[device] = torch.cuda.Stream(device=device)
        return self._side[device]

    def branch_streams(self, device, n):
        pool = self._branch.setdefault(device, [])
        while len(pool) < n:
            pool.append(torch.cuda.Stream(device=device))
        return pool[:n]

    def prototypes(self, k: int) -> torch.Tensor:
        """centers = 3 * I_K (network/utils.py:103-106); built once per device instead of every forward."""
        dev = self.store.flat_p.device
        key = (k, dev)
        if key not in self._protos:
            self._protos[key] = 3.0 * torch.eye(k, dtype=torch.float32, device=dev)
        return self._protos[key]

    def head_modules(self):
        m = self.model
        return m.head_modules() if hasattr(m, "head_modules") else [m.classifier]

    def prototypes_padded(self, k: int, kp: int) -> torch.Tensor:
        """the same centers with the embedding axis zero-padded to kp (kernels' view; see Plan.build)"""
        if kp == k:
            return self.prototypes(k)
        dev = self.store.flat_p.device
        key = (k, kp, dev)
        if key not in self._protos:
            m = torch.zeros((k, kp), dtype=torch.float32, device=dev)
            m[:, :k] = self.prototypes(k)
            self._protos[key] = m
        return self._protos[key]

    def bn_modes(self) -> int:
        """Bit i set = the i-th BatchNorm2d normalises with its running statistics (module in eval()) -- part of the
        plan key, because a training plan is built around each layer's mode."""
        if getattr(self, "_bn_list", None) is None:
            self._bn_list = self.store.bn_modules
        sig = 0
        for i, m in enumerate(self._bn_list):
            if not m.training:
                sig |= 1 << i
        return sig

    def plan_for(self, x: torch.Tensor, dtype: torch.dtype, training: bool) -> Plan:
        if not self.store.is_bound(x.device):
            self.store.bind(x.device)
            self.plans.clear()
            self._protos.clear()
        B, Cin, H, W = x.shape
        key = (B, H, W, dtype, training, bool(self.sync_bn), self.bn_modes() if training else 0, int(self.f32_split))
        plan = self.plans.get(key)
        if plan is None:
            plan = self.plan_cls(self, B, H, W, dtype, training)
            self.plans[key] = plan
        return plan

    def forward(self, x: torch.Tensor, dtype: torch.dtype, training: bool):
        if not x.is_cuda:
            raise RuntimeError("DMLNet HIP engine needs a ROCm device tensor (got %s); the CPU restatement lives "
                               "in oracle/ and is test infrastructure only" % x.device)
        if x.dim() != 4 or x.shape[1] != 3:
            raise ValueError("expected input [B,3,H,W], got %s" % (tuple(x.shape),))
        synced = bool(self.sync_bn and dist.is_available() and dist.is_initialized() and dist.get_world_size(self.sync_group) > 1)
        if training and x.shape[0] < 2 and not synced and self.bn_modes() != (1 << len(self._bn_list)) - 1:
            # same failure as the reference: BatchNorm over B x 256 x 1 x 1 in ASPPPooling (network/utils.py:318-329)
            # (synchronised statistics span the ranks: one image per rank is a legal batch there)
            raise ValueError("Expected more than 1 value per channel when training, got input size "
                             "torch.Size([%d, 256, 1, 1])" % x.shape[0])
        x = x.contiguous().float()
        plan = self.plan_for(x, dtype, training)
        stream = torch.cuda.current_stream(x.device).cuda_stream
        plan.refresh_weights(stream, overlap=training)
        B, _, H, W = x.shape
        logits, feats = [], []
        for rec in plan.heads:
            # a deferred loss gradient of this head's PREVIOUS forward that never reached backward() (the loss was dropped,
            # or a second forward ran first) is void now: its features buffer is about to be replaced
            lazy_grad.drop_for(rec.logits_ref() if getattr(rec, "logits_ref", None) is not None else None)
            lg = torch.empty((B, rec.K, H, W), dtype=torch.float32, device=x.device)
            ft = torch.empty((B, H, W, rec.Kp), dtype=torch.float32, device=x.device)
            # The plan must not own the tensors it hands out: returned from the autograd node they carry its grad_fn, whose context
            # holds model and plan -- a cycle through C++ objects the garbage collector cannot see, and every dropped model would
            # keep its plans (tens of GB at the benchmark size) for the life of the process.  An alias without autograd identity
            # (same storage, same version counter) for the features, a weak reference for the identity test on the logits.
            rec.feats_p = ft.detach()        # the kernels' (channel-padded) features; the backward reads them again
            rec.feats_version = None
            rec.logits_ref = weakref.ref(lg)
            rec.head_args[2], rec.head_args[3] = lg.data_ptr(), ft.data_ptr()
            logits.append(lg)
            feats.append(ft)
        plan.images_args[0] = x.data_ptr()
        if training:
            for args, idx, bn in plan.momentum_slots:
                if bn.momentum is None:
                    raise NotImplementedError("BatchNorm2d(momentum=None) is not supported")
                args[idx] = float(bn.momentum)
            self.step_count += 1
            for u in plan.drop_units:
                p = float(u.drop.p) if u.drop.training else 0.0      # F14: dropout module in eval() => off
                u.apply_args[14] = p
                u.apply_args[15] = (self.rank_seed() + 0x9E3779B97F4A7C15 * self.step_count) & 0xFFFFFFFFFFFFFFFF
                for (a, i) in u.gscale_slots:
                    a[i] = 1.0 / (1.0 - p) if p > 0 else 1.0
            if plan.nbt_inc is None:
                self.store.flat_nbt.add_(1)
            else:
                self.store.flat_nbt.add_(plan.nbt_inc)      # layers with fixed statistics do not count the batch
        plan.run_forward(stream)
        plan.last_input = x
        for rec in plan.heads:
            rec.feats_version = rec.feats_p._version
        for hi, rec in enumerate(plan.heads):
            if rec.Kp != rec.K:              # features_out has exactly num_classes channels (utils.py:95-97)
                feats[hi] = feats[hi][..., :rec.K].contiguous()
        return plan, logits, feats

    def backward(self, plan: Plan, glogits: List[Optional[torch.Tensor]], gfeats: List[Optional[torch.Tensor]]):
        """glogits / gfeats: one entry per head (None where the loss does not reach that output)."""
        dev = plan.device
        stream = torch.cuda.current_stream(dev).cuda_stream
        keep, skip = [], []
        last = len(plan.heads) - 1
        for hi, rec in enumerate(plan.heads):
            gl, gf = glogits[hi], gfeats[hi]
            if gl is None and gf is None and hi != last:
                skip.append(plan.head_bwd_range[hi])         # accumulating segment: nothing to add
                continue
            lazy = lazy_grad.take(gl)
            if lazy is not None and (lazy.logits is not rec.logits_ref()
                                     or tuple(lazy.logits.shape) != (plan.B, rec.K, plan.H, plan.W)):
                # the marker belongs to another forward of this plan than the one whose activations it holds now (a second
                # train-mode forward ran before this backward): the stored tensors are not the ones the loss saw
                raise RuntimeError("backward through logits of an earlier forward: the plan's activations were replaced by "
                                   "a later forward of the same shape (run backward before the next forward)")
            if lazy is not None and rec.fused_args is not None and gf is None and rec.feats_p._version == rec.feats_version:
                # deferred loss gradient + a shape the fused kernel covers: one pass, no d(loss)/d(logits) tensor (the kernel
                # recomputes the logits from the features handed to the user: an in-place edit of those since the forward
                # sends the step down the materialising path below, which uses the logits the loss saw)
                a = rec.fused_args
                a[0], a[1], a[2], a[3] = rec.feats_p.data_ptr(), lazy.labels.data_ptr(), lazy.sums.data_ptr(), lazy.gout.data_ptr()
                a[13], a[14], a[15] = int(lazy.ignore_index), float(lazy.alpha), float(lazy.n_images)
                skip.append(rec.unfused_range)
                keep += [lazy]
                continue
            if lazy is not None:                             # deferred, but not fusable here: materialise it
                gl = torch.empty((plan.B, rec.K, plan.H, plan.W), dtype=torch.float32, device=dev)
                _lib.check(self.lib.dml_loss_bwd(lazy.logits.data_ptr(), lazy.labels.data_ptr(), lazy.sums.data_ptr(),
                                                 lazy.gout.data_ptr(), gl.data_ptr(), plan.B, rec.K, plan.H, plan.W,
                                                 int(lazy.ignore_index), float(lazy.alpha), float(lazy.n_images), stream),
                           "dml_loss_bwd")
            elif gl is not None and lazy_grad.outstanding_for(rec.logits_ref()):
                lazy_grad.drop_for(rec.logits_ref())
                raise RuntimeError("the loss was built with fused_backward=True but the logits have another consumer: its "
                                   "gradient was added to the deferred-gradient marker.  Construct the loss with "
                                   "fused_backward=False")
            if rec.fused_range is not None:
                skip.append(rec.fused_range)
            if gl is None:
                gl = torch.zeros((plan.B, rec.K, plan.H, plan.W), dtype=torch.float32, device=dev)
            gl = gl.contiguous()
            if gf is not None:
                gf = gf.contiguous()
                if rec.Kp != rec.K:
                    gp = torch.zeros((plan.B, plan.H, plan.W, rec.Kp), dtype=torch.float32, device=dev)
                    gp[..., :rec.K] = gf
                    gf = gp
            a = rec.head_bwd_args
            a[0] = gl.data_ptr()
            a[1] = gf.data_ptr() if gf is not None else None
            a[2] = rec.feats_p.data_ptr()
            keep += [gl, gf]
        self._keep_bwd = keep
        if not any(p.requires_grad for p in self.model.backbone.parameters()):
            skip.append(plan.backbone_bwd_range)
            skip += [(i, i + 1) for i in plan.to_backbone_ops]
        plan.skip_ranges = skip
        state = self.store.begin_backward()
        if self.reducer is not None:
            if state == "accumulate" and self.reducer.world > 1:
                # the bucketed all-reduce works in place on the flat gradient: a second backward without zero_grad()
                # would sum the already reduced gradients over the ranks again (x world), unlike the single-GPU path
                raise RuntimeError("gradient accumulation over several backward passes is not supported with the "
                                   "data-parallel reducer attached: call optimizer.zero_grad() (set_to_none=True) "
                                   "before every backward")
            self.reducer.run_backward(plan, stream)
        else:
            plan.run_backward()
        self.store.end_backward()
