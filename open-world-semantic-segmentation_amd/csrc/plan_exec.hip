// Native replay of a static launch list (the host side of dmlnet/engine.py's Plan).
//
// A train step is ~830 launches of this library's own entry points with arguments that are fixed when the plan is
// built.  Issuing them from Python through ctypes costs 35 ms of host time per 44 ms step (bench.py,
// host_enqueue_ms_per_step) -- the host, not the GPU, would bound the step as soon as the kernels get faster.  Here the
// list is a flat array of {entry point, packed arguments} that one C call walks: the launches are the same C-ABI calls in
// the same order on the same streams, only the interpreter is gone.  Weight gradients go to a side stream exactly as
// Plan.run_backward does it (event on the main stream, side stream waits, launch).
//
// Replaces the per-module Python dispatch of the reference's nn.Module.forward / autograd backward
// (network/utils.py:84-118, backbone/resnet.py:95-115 of the reference: one Python call per layer per step).
#include "common.h"

#include <cstring>
#include <tuple>
#include <utility>

namespace {

template <class T> struct Unpack {
    static T get(uint64_t w) {
        if constexpr (std::is_pointer<T>::value) {
            return reinterpret_cast<T>(static_cast<uintptr_t>(w));
        } else if constexpr (std::is_same<T, float>::value) {
            const uint32_t lo = (uint32_t)w;
            float f;
            std::memcpy(&f, &lo, 4);
            return f;
        } else if constexpr (std::is_same<T, double>::value) {
            double d;
            std::memcpy(&d, &w, 8);
            return d;
        } else {
            return static_cast<T>(static_cast<int64_t>(w));       // int / int64_t / uint64_t
        }
    }
};

// every entry point that can appear in a plan has the shape int f(A0, ..., An-1, void* stream)
template <class R, class... Args, size_t... I>
int invoke(R (*f)(Args...), const uint64_t* w, void* st, std::index_sequence<I...>) {
    using Tup = std::tuple<Args...>;
    return (int)f(Unpack<std::tuple_element_t<I, Tup>>::get(w[I])..., st);
}
template <class R, class... Args> int thunk_call(R (*f)(Args...), const uint64_t* w, void* st) {
    return invoke(f, w, st, std::make_index_sequence<sizeof...(Args) - 1>{});
}
template <class R, class... Args> constexpr int arity(R (*)(Args...)) { return (int)sizeof...(Args) - 1; }

struct Entry {
    const char* name;
    int (*run)(const uint64_t*, void*);
    int nargs;
};

#define DML_ENTRY(fn) {#fn, [](const uint64_t* w, void* st) -> int { return thunk_call(&fn, w, st); }, arity(&fn)}

const Entry kEntries[] = {
    DML_ENTRY(dml_conv_igemm),
    DML_ENTRY(dml_conv_wgrad),
    DML_ENTRY(dml_conv_wgrad_group),
    DML_ENTRY(dml_prep_weight),
    DML_ENTRY(dml_prep_weights),
    DML_ENTRY(dml_unpad_wgrad),
    DML_ENTRY(dml_bias_grad),
    DML_ENTRY(dml_bias_grad_ws),
    DML_ENTRY(dml_pack_input),
    DML_ENTRY(dml_pack_input_s2d),
    DML_ENTRY(dml_s2d_weights),
    DML_ENTRY(dml_s2d_wgrad),
    DML_ENTRY(dml_bn_finalize),
    DML_ENTRY(dml_bn_moments),
    DML_ENTRY(dml_bn_finalize_moments),
    DML_ENTRY(dml_bn_bwd_sums),
    DML_ENTRY(dml_bn_bwd_coef),
    DML_ENTRY(dml_bn_stats),
    DML_ENTRY(dml_bn_eval_coeffs),
    DML_ENTRY(dml_bn_eval_coeffs_table),
    DML_ENTRY(dml_bn_apply),
    DML_ENTRY(dml_bn_bwd_reduce),
    DML_ENTRY(dml_bn_bwd_finalize),
    DML_ENTRY(dml_bn_bwd_apply),
    DML_ENTRY(dml_maxpool3x3s2_fwd),
    DML_ENTRY(dml_maxpool3x3s2_bwd),
    DML_ENTRY(dml_global_avgpool_fwd),
    DML_ENTRY(dml_broadcast_hw),
    DML_ENTRY(dml_reduce_hw),
    DML_ENTRY(dml_reduce_hw_f32),
    DML_ENTRY(dml_avgpool_bwd_add),
    DML_ENTRY(dml_avgpool_bwd_set),
    DML_ENTRY(dml_bilinear_fwd),
    DML_ENTRY(dml_bilinear_bwd),
    DML_ENTRY(dml_proto_dist_fwd),
    DML_ENTRY(dml_upsample_dist_fwd),
    DML_ENTRY(dml_proto_dist_bwd),
    DML_ENTRY(dml_head_bwd_fused),
    DML_ENTRY(dml_argmax_msp),
    DML_ENTRY(dml_dissum_score),
    DML_ENTRY(dml_novel_relabel),
    DML_ENTRY(dml_loss_fwd),
    DML_ENTRY(dml_loss_finalize),
    DML_ENTRY(dml_loss_bwd),
    DML_ENTRY(dml_sgd_step),
    DML_ENTRY(dml_fill_f32),
    DML_ENTRY(dml_convert_dtype),
    DML_ENTRY(dml_adaptive_avgpool_fwd),
    DML_ENTRY(dml_proto_dist_nhwc),
    DML_ENTRY(dml_upsample_nhwc_to_nchw),
    DML_ENTRY(dml_confusion_update),
    DML_ENTRY(dml_class_feature_sum),
    DML_ENTRY(dml_label_encode),
    DML_ENTRY(dml_h2_split),
    DML_ENTRY(dml_h2_bound_bn),
    DML_ENTRY(dml_h2_bound_bn_bwd),
    DML_ENTRY(dml_h2_bound_bn_table),
    DML_ENTRY(dml_h2_bound_bn_multi),
    DML_ENTRY(dml_bilinear_fwd_planes),
    DML_ENTRY(dml_bn_finalize_bound),
    DML_ENTRY(dml_bn_bwd_finalize_bound),
};
constexpr int kNumEntries = (int)(sizeof(kEntries) / sizeof(kEntries[0]));

}  // namespace

extern "C" int dml_plan_fn_id(const char* name) {
    if (!name) return DML_EINVAL;
    for (int i = 0; i < kNumEntries; ++i)
        if (std::strcmp(kEntries[i].name, name) == 0) return i;
    return DML_EINVAL;
}

extern "C" int dml_plan_fn_nargs(int fn) { return (fn >= 0 && fn < kNumEntries) ? kEntries[fn].nargs : DML_EINVAL; }

namespace {
int plan_run(const DmlPlanOp* ops, int first, int last, void* stream, void* side_stream, void* const* events, int n_events,
             const int32_t* marks, int n_marks, void* const* mark_events, int* failed_op);
}

extern "C" int dml_plan_run(const DmlPlanOp* ops, int first, int last, void* stream, void* side_stream,
                            void* const* events, int n_events, int* failed_op) {
    return plan_run(ops, first, last, stream, side_stream, events, n_events, nullptr, 0, nullptr, failed_op);
}

// dml_plan_run + MARKS: after op marks[k] (ascending op indices; those outside [first, last) are ignored) has been issued,
// mark_events[2 k] is recorded on `stream` and -- with a side stream -- mark_events[2 k + 1] on `side_stream`.  The data-parallel
// reducer (dmlnet/parallel.py) lets its communication stream wait on them: the gradient buckets' all-reduces start at the same
// points of the backward as before, but the host walks the whole launch list in ONE call instead of returning to Python per bucket.
extern "C" int dml_plan_run_marks(const DmlPlanOp* ops, int first, int last, void* stream, void* side_stream,
                                  void* const* events, int n_events, const int32_t* marks, int n_marks,
                                  void* const* mark_events, int* failed_op) {
    if (n_marks < 0 || (n_marks > 0 && (!marks || !mark_events))) return DML_EINVAL;
    for (int k = 1; k < n_marks; ++k)
        if (marks[k] <= marks[k - 1]) return DML_EINVAL;
    return plan_run(ops, first, last, stream, side_stream, events, n_events, marks, n_marks, mark_events, failed_op);
}

namespace {
int plan_run(const DmlPlanOp* ops, int first, int last, void* stream, void* side_stream, void* const* events, int n_events,
             const int32_t* marks, int n_marks, void* const* mark_events, int* failed_op) {
    if (!ops || first < 0 || last < first) return DML_EINVAL;
    hipStream_t main_st = static_cast<hipStream_t>(stream);
    hipStream_t side_st = static_cast<hipStream_t>(side_stream);
    int ev = 0, mk = 0;
    while (mk < n_marks && marks[mk] < first) ++mk;
    for (int i = first; i < last; ++i) {
        const DmlPlanOp& op = ops[i];
        if (op.fn < 0 || op.fn >= kNumEntries || op.nargs != kEntries[op.fn].nargs || op.nargs > DML_PLAN_MAX_ARGS) {
            if (failed_op) *failed_op = i;
            return DML_EINVAL;
        }
        void* st = stream;
        if (op.stream == 1 && side_st != nullptr) {
            if (op.wait) {
                // the side stream picks up everything enqueued on the main stream so far (the producer of this op's inputs)
                if (!events || n_events <= 0) {
                    if (failed_op) *failed_op = i;
                    return DML_EINVAL;
                }
                hipEvent_t e = static_cast<hipEvent_t>(events[ev]);
                ev = (ev + 1) % n_events;
                hipError_t rc = hipEventRecord(e, main_st);
                if (rc == hipSuccess) rc = hipStreamWaitEvent(side_st, e, 0);
                if (rc != hipSuccess) {
                    if (failed_op) *failed_op = i;
                    return (int)rc;
                }
            }
            st = side_stream;
        }
        uint64_t words[DML_PLAN_MAX_ARGS];
        const uint64_t* w = op.args;
        if (op.indirect) {
            // arguments whose value another entry point writes on the host at enqueue time (dml_bn_bwd_reduce's *nblocks)
            for (int k = 0; k < op.nargs; ++k)
                words[k] = (op.indirect >> k) & 1u
                               ? (uint64_t)(int64_t)*reinterpret_cast<const int32_t*>(static_cast<uintptr_t>(op.args[k]))
                               : op.args[k];
            w = words;
        }
        const int rc = kEntries[op.fn].run(w, st);
        if (rc != 0) {
            if (failed_op) *failed_op = i;
            return rc;
        }
        if (mk < n_marks && marks[mk] == i) {
            hipError_t e = hipEventRecord(static_cast<hipEvent_t>(mark_events[2 * mk]), main_st);
            if (e == hipSuccess && side_st != nullptr) e = hipEventRecord(static_cast<hipEvent_t>(mark_events[2 * mk + 1]), side_st);
            if (e != hipSuccess) {
                if (failed_op) *failed_op = i;
                return (int)e;
            }
            ++mk;
        }
    }
    return 0;
}
}  // namespace
