"""ctypes binding of libdmlnet_hip.so (C ABI declared in include/dmlnet_hip.h).

The library is the product: there is no CPU or eager-PyTorch fallback.  If the shared object is
missing or an entry point returns non-zero, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DML_LIB_PATH: another build of the same library (A/B runs of two builds, the tuning build `make tuning`) -- selected here, not by
# copying over the in-tree file another process may have mapped
LIB_PATH = os.environ.get("DML_LIB_PATH") or os.path.join(_HERE, "libdmlnet_hip.so")

DML_F32, DML_BF16 = 0, 1
STAT_ROWS = 64
LOSS_BLOCKS = 2048

c_p = C.c_void_p
c_i = C.c_int
c_i64 = C.c_int64
c_f = C.c_float


class ConvDesc(C.Structure):
    _fields_ = [("x", c_p), ("w", c_p), ("y", c_p), ("bias", c_p), ("stats", c_p),
                ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("C", C.c_int32), ("ldx", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("N", C.c_int32), ("ldy", C.c_int32),
                ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32),
                ("pad", C.c_int32), ("dtype", C.c_int32), ("y_f32", C.c_int32), ("accum", C.c_int32),
                ("mode", C.c_int32),
                # mode 1: fused BN-backward partial sums of the tensor being written (see the header)
                ("bnr_y", c_p), ("bnr_mask", c_p), ("bnr_mean", c_p), ("bnr_invstd", c_p), ("bnr_partials", c_p),
                ("bnr_ldy", C.c_int32), ("bnr_relu", C.c_int32),
                # mode 0: inference epilogue BN(running stats) + residual + ReLU (see the header)
                ("post_scale", c_p), ("post_shift", c_p), ("post_mean", c_p), ("post_res", c_p),
                ("post_ldres", C.c_int32), ("post_relu", C.c_int32),
                # K-split of the tiles of a partially filled last round (see the header)
                ("tail_ws", c_p), ("tail_ws_elems", C.c_int64), ("tail_counters", c_p),
                ("tail_counters_len", C.c_int32), ("tail_reserved", C.c_int32),
                ("res_dz", C.c_void_p), ("res_mask", C.c_void_p), ("res_ld", C.c_int32), ("res_reserved", C.c_int32),
                # fp32 staging of a gradient with several producers, rounded once by the last one (see the header)
                ("acc32", C.c_void_p), ("acc32_ld", C.c_int32), ("f32_split", C.c_int32),
                ("w_tiled", C.c_int32), ("ws_min_tiles", C.c_int32),
                # f32_split == 2: fp16 hi / lo planes of the scaled operands (dml_h2_split; see the header)
                ("x_planes", c_p), ("w_planes", c_p), ("x_unscale", c_p), ("w_unscale", c_p),
                ("x_plane_stride", C.c_int64), ("w_plane_stride", C.c_int64), ("bnr_gmax", c_p),
                # mode 1, f32_split == 2: one parity class of a stride-2 data gradient as a stride-1 launch (see the header)
                ("sub_grid", C.c_int32), ("sub_y", C.c_int32), ("sub_x", C.c_int32),
                ("pad_w_set", C.c_int32), ("pad_w", C.c_int32), ("bnr_inc", C.c_int32)]


PLAN_MAX_ARGS = 24


class PlanOp(C.Structure):
    """DmlPlanOp: one packed launch of a native launch list (dml_plan_run)"""
    _fields_ = [("fn", C.c_int32), ("nargs", C.c_int32), ("stream", C.c_int32), ("wait", C.c_int32),
                ("indirect", C.c_uint32), ("reserved", C.c_uint32), ("args", C.c_uint64 * PLAN_MAX_ARGS)]


class BnEvalDesc(C.Structure):
    _fields_ = [("gamma", c_p), ("beta", c_p), ("running_var", c_p), ("scale", c_p), ("shift", c_p),
                ("N", C.c_int32), ("eps", C.c_float)]


class WgradDesc(C.Structure):
    _fields_ = [("x", c_p), ("dy", c_p), ("dw", c_p),
                ("B", C.c_int32), ("Hi", C.c_int32), ("Wi", C.c_int32), ("C", C.c_int32), ("ldx", C.c_int32),
                ("Ho", C.c_int32), ("Wo", C.c_int32), ("N", C.c_int32), ("ldy", C.c_int32),
                ("R", C.c_int32), ("S", C.c_int32), ("stride", C.c_int32), ("dil", C.c_int32),
                ("pad", C.c_int32), ("dtype", C.c_int32), ("splitk", C.c_int32), ("Cm", C.c_int32),
                ("ws", c_p), ("ws_elems", C.c_int64), ("f32_split", C.c_int32), ("reserved", C.c_int32),
                ("x_planes", c_p), ("dy_planes", c_p), ("x_unscale", c_p), ("dy_unscale", c_p),
                ("x_plane_stride", C.c_int64), ("dy_plane_stride", C.c_int64)]


class PrepDesc(C.Structure):
    _fields_ = [("src", c_p), ("w", c_p), ("wt", c_p), ("N", C.c_int32), ("RS", C.c_int32), ("Cm", C.c_int32),
                ("Cp", C.c_int32), ("w_tiled", C.c_int32), ("wt_tiled", C.c_int32)]


class H2Desc(C.Structure):
    """DmlH2Desc: one tensor of a dml_h2_split_table launch"""
    _fields_ = [("x", c_p), ("planes", c_p), ("work", c_p), ("rows", C.c_int64), ("plane_stride", C.c_int64),
                ("C", C.c_int32), ("ld", C.c_int32), ("ldp", C.c_int32), ("layout", C.c_int32)]


class H2BoundDesc(C.Structure):
    """DmlH2BoundDesc: one BatchNorm of a dml_h2_bound_bn_table launch"""
    _fields_ = [("gamma", c_p), ("beta", c_p), ("work", c_p), ("N", C.c_int32), ("root_count", C.c_float), ("mult", C.c_float),
                ("reserved", C.c_int32)]


class AugSample(C.Structure):
    _fields_ = [("i", C.c_int32), ("j", C.c_int32), ("flip", C.c_int32), ("n_ops", C.c_int32),
                ("op", C.c_int32 * 3), ("factor", C.c_float * 3)]


_PROTOS = {
    "dml_abi_version": (c_i, []),
    "dml_target_arch": (C.c_char_p, []),
    "dml_conv_igemm": (c_i, [C.POINTER(ConvDesc), c_p]),
    "dml_conv_stat_rows": (c_i, [C.POINTER(ConvDesc)]),
    "dml_h2_split_table": (c_i, [c_p, c_i, c_p]),
    "dml_h2_split": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_i64, c_i, c_i, c_p, c_i, c_p]),
    "dml_conv_wgrad": (c_i, [C.POINTER(WgradDesc), c_p]),
    "dml_prep_weight": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_prep_weights": (c_i, [c_p, c_i, c_i, c_p]),
    "dml_unpad_wgrad": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dml_bias_grad": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "dml_bias_grad_ws": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p, c_i64, c_p]),
    "dml_pack_input": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_pack_input_s2d": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dml_s2d_weights": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "dml_s2d_wgrad": (c_i, [c_p, c_p, c_i, c_i, c_i, c_p]),
    "dml_gather_taps": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_bn_finalize": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    "dml_bn_moments": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p]),
    "dml_bn_finalize_moments": (c_i, [c_p, c_i, c_i64, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p, c_p]),
    "dml_bn_bwd_sums": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_p]),
    "dml_bn_bwd_coef": (c_i, [c_p, c_i64, c_i, c_p, c_p, c_p, c_p, c_p]),
    "dml_bn_stats": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_i, c_p]),
    "dml_bn_eval_coeffs": (c_i, [c_p, c_p, c_p, c_p, c_f, c_p, c_p, c_i, c_p]),
    "dml_bn_eval_coeffs_table": (c_i, [c_p, c_i, c_p]),
    "dml_bn_apply": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_f, C.c_uint64, c_p,
                           c_p, c_i64, c_i, c_p, c_i64, c_p, c_p]),
    "dml_bn_bwd_reduce": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_f, c_i,
                                C.POINTER(c_i), c_p, c_p]),
    "dml_bn_bwd_finalize": (c_i, [c_p, c_i, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    "dml_bn_bwd_apply": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f,
                               c_i, c_i, c_p, c_p, c_i64, c_i, c_p, c_p]),
    "dml_h2_bound_bn": (c_i, [c_p, c_p, c_i, c_i64, c_f, c_p, c_p, c_p]),
    "dml_h2_bound_bn_table": (c_i, [c_p, c_i, c_p]),
    "dml_h2_bound_bn_multi": (c_i, [c_p, c_i, c_p, c_p]),
    "dml_bilinear_fwd_planes": (c_i, [c_p, c_p, c_i64, c_i, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_h2_bound_bn_bwd": (c_i, [c_p, c_p, c_i, c_i64, c_p, c_p, c_p]),
    "dml_bn_finalize_bound": (c_i, [c_p, c_i64, c_i, c_i, c_p, c_p, c_p, c_p, c_f, c_f, c_p, c_p, c_p, c_p,
                                    c_i64, c_f, c_p, c_p, c_p, c_p]),
    "dml_bn_bwd_finalize_bound": (c_i, [c_p, c_i, c_i64, c_i, c_p, c_p, c_p, c_p, c_p, c_p, c_i64, c_p, c_p, c_p, c_p]),
    "dml_maxpool3x3s2_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_maxpool3x3s2_bwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_global_avgpool_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_broadcast_hw": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_reduce_hw": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_reduce_hw_f32": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_p]),
    "dml_avgpool_bwd_add": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_avgpool_bwd_set": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_bilinear_fwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_bilinear_bwd": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_proto_dist_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_upsample_dist_fwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_proto_dist_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_head_bwd_fused": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_i64, c_f, c_f, c_i, c_p]),
    "dml_argmax_msp": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_p]),
    "dml_dissum_score": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "dml_novel_relabel": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_i64, c_p]),
    "dml_loss_fwd": (c_i, [c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_p]),
    "dml_loss_finalize": (c_i, [c_p, c_p, c_f, c_f, c_p]),
    "dml_loss_bwd": (c_i, [c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i64, c_f, c_f, c_p]),
    "dml_sgd_step": (c_i, [c_p, c_p, c_p, c_i64, c_f, c_f, c_f, c_f, c_p]),
    "dml_fill_f32": (c_i, [c_p, c_i64, c_f, c_p]),
    "dml_convert_dtype": (c_i, [c_p, c_p, c_i64, c_i, c_i, c_p]),
    "dml_confusion_update": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_p]),
    "dml_class_feature_sum": (c_i, [c_p, c_p, c_i64, c_i, c_i64, c_p, c_p, c_p]),
    "dml_ood_workspace_bytes": (c_i64, [c_i64]),
    "dml_ood_measures": (c_i, [c_p, c_p, c_p, c_i64, c_p, c_i, C.c_double, c_p, c_i64, c_p, c_p]),
    "dml_aug_contrast_sum": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_aug_apply": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f, c_p]),
    "dml_aug_apply_encoded": (c_i, [c_p, c_p, c_p, c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_f, c_f, c_f, c_f, c_f, c_f,
                                    c_p, c_p, c_p, c_p]),
    "dml_label_encode": (c_i, [c_p, C.c_int64, c_p, c_p, c_p, c_p, c_p]),
    "dml_adaptive_avgpool_ws_elems": (c_i64, [c_i, c_i, c_i, c_i, c_i]),
    "dml_adaptive_avgpool_fwd": (c_i, [c_p, c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_p]),
    "dml_proto_dist_nhwc": (c_i, [c_p, c_p, c_p, c_i64, c_i, c_i, c_i, c_i, c_p]),
    "dml_upsample_nhwc_to_nchw": (c_i, [c_p, c_p, c_i, c_i, c_i, c_i, c_i, c_i, c_i, c_f, c_i, c_p]),
    "dml_conv_wgrad_group_eligible": (c_i, [c_p]),
    "dml_conv_wgrad_group": (c_i, [c_p, c_i, c_p, c_i64, c_p]),
    "dml_plan_fn_id": (c_i, [C.c_char_p]),
    "dml_plan_fn_nargs": (c_i, [c_i]),
    "dml_plan_run": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_i, C.POINTER(c_i)]),
    "dml_plan_run_marks": (c_i, [c_p, c_i, c_i, c_p, c_p, c_p, c_i, c_p, c_i, c_p, C.POINTER(c_i)]),
}

EXPORTS = tuple(_PROTOS.keys())

_lib = None


class DmlError(RuntimeError):
    pass


def load():
    """Load libdmlnet_hip.so; raises (never falls back) when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DmlError(
            "libdmlnet_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _PROTOS.items():
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    if lib.dml_abi_version() != 6:
        raise DmlError("libdmlnet_hip.so ABI version mismatch")
    _lib = lib
    return lib


_ERR = {-1: "invalid argument", -2: "misaligned / non-vectorisable shape", -3: "unsupported configuration"}


def check(rc: int, what: str):
    """0 = ok; negative = a DML_E* argument check of the library; positive = the hipError_t of a failed launch, event
    record or stream wait (include/dmlnet_hip.h, conventions)."""
    if rc == 0:
        return
    if rc > 0:
        raise DmlError("%s failed: HIP runtime error, hipError_t %d" % (what, rc))
    raise DmlError("%s failed: %s (code %d)" % (what, _ERR.get(rc, "unknown DML error"), rc))
