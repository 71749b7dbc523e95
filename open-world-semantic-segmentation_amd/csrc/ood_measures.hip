// Pixel-level OOD measures on the device: AUROC, AUPR and FPR at a recall level of the scores -conf over the pixels
// whose label is / is not in the out-of-distribution label set (anomaly/anom_utils.py:25-78, called from
// anomaly/eval_ood_traditional.py:128-148; sklearn's roc_auc_score / average_precision_score there).
// The reference copies every score map to the host and argsorts ~2 M values per image; here:
//   1. one key per pixel: (class << 32) | order-preserving bits of the float score, class 0 = positive (OOD),
//      1 = negative, 2 = masked out;
//   2. a stable LSD radix sort of the 34 significant bits (5 passes of 8): per-wave digit histograms, one scan,
//      ballot-ranked scatter -- positives and negatives end up as two ascending runs of one array;
//   3. rank statistics by binary search: AUROC = sum_pos (#neg below + #neg not above) / (2 P N) in exact integers;
//      AP and the recall cut-off from the distinct positive values (see oracle/ood_measures_ref.py).
#include "common.h"

namespace {

constexpr int SORT_BLOCKS = 256;              // 4 waves each; every wave owns one contiguous chunk of the keys
constexpr int SORT_WAVES = SORT_BLOCKS * 4;
constexpr int RADIX = 256;
constexpr int MEAS_BLOCKS = 256;

struct OutLabels { int64_t v[8]; int n; };

__device__ __forceinline__ uint32_t order_bits(float f) {
    const uint32_t b = __float_as_uint(f);
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);       // ascending float order == ascending unsigned order
}

__global__ __launch_bounds__(256) void ood_keys_kernel(const float* __restrict__ conf, const int64_t* __restrict__ lab,
                                                       const uint8_t* __restrict__ mask, int64_t n, OutLabels ol,
                                                       uint64_t* __restrict__ keys, unsigned long long* __restrict__ pn) {
    __shared__ unsigned long long sh[2];
    if (threadIdx.x < 2) sh[threadIdx.x] = 0ull;
    __syncthreads();
    unsigned long long p = 0, q = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        uint64_t cls = 1;
        const int64_t l = lab[i];
        for (int k = 0; k < ol.n; ++k)
            if (l == ol.v[k]) cls = 0;
        if (mask != nullptr && mask[i] == 0) cls = 2;
        const float s = -conf[i] + 0.0f;                        // eval_ood_traditional.py:140-141 (+0: -0.0 and 0.0 tie)
        keys[i] = (cls << 32) | (uint64_t)order_bits(s);
        p += cls == 0;
        q += cls == 1;
    }
    atomicAdd(&sh[0], p);
    atomicAdd(&sh[1], q);
    __syncthreads();
    if (threadIdx.x < 2 && sh[threadIdx.x]) atomicAdd(pn + threadIdx.x, sh[threadIdx.x]);
}

// ---- radix sort pass: digit = (key >> shift) & 255 ------------------------------------------------------------
__global__ __launch_bounds__(256) void sort_hist_kernel(const uint64_t* __restrict__ keys, int64_t n, int64_t chunk,
                                                        int shift, uint32_t* __restrict__ counts) {
    __shared__ uint32_t h[4][RADIX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int i = lane; i < RADIX; i += 64) h[wave][i] = 0u;
    __syncthreads();
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    const int64_t beg = w * chunk, end = min(n, beg + chunk);
    for (int64_t i = beg + lane; i < end; i += 64) atomicAdd(&h[wave][(keys[i] >> shift) & 255u], 1u);
    __syncthreads();
    for (int i = lane; i < RADIX; i += 64) counts[(int64_t)i * SORT_WAVES + w] = h[wave][i];
}

// exclusive scan of counts[RADIX * SORT_WAVES] (digit-major), one workgroup
__global__ __launch_bounds__(1024) void sort_scan_kernel(uint32_t* __restrict__ counts) {
    __shared__ uint32_t part[1024];
    constexpr int PER = RADIX * SORT_WAVES / 1024;
    uint32_t s = 0;
    const int base = threadIdx.x * PER;
    for (int i = 0; i < PER; ++i) s += counts[base + i];
    part[threadIdx.x] = s;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const uint32_t v = threadIdx.x >= o ? part[threadIdx.x - o] : 0u;
        __syncthreads();
        part[threadIdx.x] += v;
        __syncthreads();
    }
    uint32_t run = part[threadIdx.x] - s;                      // exclusive prefix of this thread's segment
    for (int i = 0; i < PER; ++i) {
        const uint32_t c = counts[base + i];
        counts[base + i] = run;
        run += c;
    }
}

__global__ __launch_bounds__(256) void sort_scatter_kernel(const uint64_t* __restrict__ keys, uint64_t* __restrict__ out,
                                                           int64_t n, int64_t chunk, int shift,
                                                           const uint32_t* __restrict__ offsets) {
    __shared__ uint32_t off[4][RADIX];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t w = (int64_t)blockIdx.x * 4 + wave;
    for (int i = lane; i < RADIX; i += 64) off[wave][i] = offsets[(int64_t)i * SORT_WAVES + w];
    __syncthreads();
    const int64_t beg = w * chunk, end = min(n, beg + chunk);
    const uint64_t lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int64_t i0 = beg; i0 < end; i0 += 64) {
        const int64_t i = i0 + lane;
        const bool act = i < end;
        const uint64_t key = act ? keys[i] : 0ull;
        const uint32_t d = act ? (uint32_t)((key >> shift) & 255u) : 0u;
        uint64_t peers = __ballot(act);                        // lanes holding the same digit (stable multisplit)
#pragma unroll
        for (int b = 0; b < 8; ++b) {
            const bool bit = (d >> b) & 1u;
            const uint64_t bal = __ballot(bit);
            peers &= bit ? bal : ~bal;
        }
        if (act) {
            const uint32_t rank = (uint32_t)__popcll(peers & lt);
            const uint32_t base = off[wave][d];
            out[(int64_t)base + rank] = key;
            if (rank == 0) off[wave][d] = base + (uint32_t)__popcll(peers);   // same wave: LDS ops stay in order
        }
    }
}

// ---- rank statistics -------------------------------------------------------------------------------------------
__device__ __forceinline__ int64_t lower_bound(const uint64_t* __restrict__ a, int64_t lo, int64_t hi, uint64_t v) {
    while (lo < hi) {                                           // first index with a[i] >= v
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}
__device__ __forceinline__ int64_t upper_bound(const uint64_t* __restrict__ a, int64_t lo, int64_t hi, uint64_t v) {
    while (lo < hi) {                                           // first index with a[i] > v
        const int64_t mid = (lo + hi) >> 1;
        if (a[mid] <= v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

struct Cut { double d, r, fp; };                                // |recall - level|, recall, false positives there
__device__ __forceinline__ bool better(const Cut& a, const Cut& b) { return a.d < b.d || (a.d == b.d && a.r > b.r); }

__global__ __launch_bounds__(256) void ood_measure_kernel(const uint64_t* __restrict__ keys,
                                                          const unsigned long long* __restrict__ pn, double recall_level,
                                                          unsigned long long* __restrict__ s2_part,
                                                          double* __restrict__ ap_part, Cut* __restrict__ cut_part) {
    __shared__ unsigned long long sh_s[256];
    __shared__ double sh_a[256];
    __shared__ Cut sh_c[256];
    const int64_t P = (int64_t)pn[0], N = (int64_t)pn[1];
    unsigned long long s2 = 0;
    double ap = 0.0;
    Cut best = {1e300, -1.0, 0.0};
    const uint64_t NEG = 1ull << 32;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (int64_t)gridDim.x * 256) {
        const uint64_t k = keys[i];                             // class 0: the key is the score
        const int64_t below = lower_bound(keys, P, P + N, NEG | k) - P;     // negatives with a lower score
        const int64_t notabove = upper_bound(keys, P, P + N, NEG | k) - P;  // ... lower or equal
        s2 += (unsigned long long)(below + notabove);
        if (i == 0 || keys[i - 1] != k) {                       // first of a run of equal positives: one threshold
            const int64_t cnt = upper_bound(keys, i, P, k) - i;
            const int64_t tp = P - i, fp = N - below;           // score >= this value
            const double r = (double)tp / (double)P, rprev = (double)(tp - cnt) / (double)P;
            ap += (r - rprev) * ((double)tp / (double)(tp + fp));
            // the recall plateau's lowest threshold: everything above the next lower positive value; the plateau of
            // full recall is cut at its first threshold (anom_utils.py:58-62)
            const double fpc = i == 0 ? (double)fp : (double)(N - (upper_bound(keys, P, P + N, NEG | keys[i - 1]) - P));
            const Cut c = {fabs(r - recall_level), r, fpc};
            if (better(c, best)) best = c;
        }
    }
    sh_s[threadIdx.x] = s2;
    sh_a[threadIdx.x] = ap;
    sh_c[threadIdx.x] = best;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {                         // fixed tree: deterministic sums
        if (threadIdx.x < o) {
            sh_s[threadIdx.x] += sh_s[threadIdx.x + o];
            sh_a[threadIdx.x] += sh_a[threadIdx.x + o];
            if (better(sh_c[threadIdx.x + o], sh_c[threadIdx.x])) sh_c[threadIdx.x] = sh_c[threadIdx.x + o];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        s2_part[blockIdx.x] = sh_s[0];
        ap_part[blockIdx.x] = sh_a[0];
        cut_part[blockIdx.x] = sh_c[0];
    }
}

__global__ void ood_finalize_kernel(const unsigned long long* __restrict__ pn, const unsigned long long* __restrict__ s2_part,
                                    const double* __restrict__ ap_part, const Cut* __restrict__ cut_part, int nblk,
                                    double* __restrict__ result) {
    const double P = (double)pn[0], N = (double)pn[1];
    unsigned long long s2 = 0;
    double ap = 0.0;
    Cut best = {1e300, -1.0, 0.0};
    for (int b = 0; b < nblk; ++b) {
        s2 += s2_part[b];
        ap += ap_part[b];
        if (better(cut_part[b], best)) best = cut_part[b];
    }
    const bool ok = pn[0] > 0 && pn[1] > 0;
    result[0] = ok ? (double)s2 / (2.0 * P * N) : nan("");
    result[1] = ok ? ap : nan("");
    result[2] = ok ? best.fp / N : nan("");
    result[3] = P;
    result[4] = N;
}

struct Work {
    uint64_t *ka, *kb;
    uint32_t* counts;
    unsigned long long *pn, *s2;
    double* ap;
    Cut* cut;
};
inline int64_t align256(int64_t v) { return (v + 255) & ~255ll; }
inline int64_t carve(char* base, int64_t n, Work* w) {
    int64_t o = 0;
    auto take = [&](int64_t bytes) { char* p = base ? base + o : nullptr; o += align256(bytes); return p; };
    char* a = take(n * 8); char* b = take(n * 8); char* c = take((int64_t)RADIX * SORT_WAVES * 4);
    char* d = take(16); char* e = take(MEAS_BLOCKS * 8); char* f = take(MEAS_BLOCKS * 8);
    char* g = take(MEAS_BLOCKS * (int64_t)sizeof(Cut));
    if (w) {
        w->ka = (uint64_t*)a; w->kb = (uint64_t*)b; w->counts = (uint32_t*)c; w->pn = (unsigned long long*)d;
        w->s2 = (unsigned long long*)e; w->ap = (double*)f; w->cut = (Cut*)g;
    }
    return o;
}

}  // namespace

extern "C" int64_t dml_ood_workspace_bytes(int64_t n) { return n > 0 ? carve(nullptr, n, nullptr) : 0; }

extern "C" int dml_ood_measures(const float* conf, const int64_t* seg_label, const uint8_t* mask, int64_t n,
                                const int64_t* out_labels, int n_out, double recall_level, void* work, int64_t work_bytes,
                                double* result, void* stream) {
    if (!conf || !seg_label || !out_labels || !work || !result || n <= 0 || n_out <= 0 || n_out > 8) return DML_EINVAL;
    if (n >= (1ll << 31)) return DML_EUNSUPPORTED;             // 32-bit scatter offsets
    if ((reinterpret_cast<uintptr_t>(work) & 255) || work_bytes < dml_ood_workspace_bytes(n)) return DML_EINVAL;
    Work w;
    carve(static_cast<char*>(work), n, &w);
    OutLabels ol;
    ol.n = n_out;
    for (int k = 0; k < 8; ++k) ol.v[k] = k < n_out ? out_labels[k] : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(w.pn, 0, 16, st) != hipSuccess) return DML_EINVAL;
    hipLaunchKernelGGL(ood_keys_kernel, dim3(grid_for(n, 256, 1024)), dim3(256), 0, st, conf, seg_label, mask, n, ol, w.ka, w.pn);
    int64_t chunk = (n + SORT_WAVES - 1) / SORT_WAVES;
    chunk = (chunk + 63) & ~63ll;
    uint64_t *src = w.ka, *dst = w.kb;
    for (int pass = 0; pass < 5; ++pass) {                      // 34 significant bits: 32 of the score + 2 of the class
        hipLaunchKernelGGL(sort_hist_kernel, dim3(SORT_BLOCKS), dim3(256), 0, st, src, n, chunk, pass * 8, w.counts);
        hipLaunchKernelGGL(sort_scan_kernel, dim3(1), dim3(1024), 0, st, w.counts);
        hipLaunchKernelGGL(sort_scatter_kernel, dim3(SORT_BLOCKS), dim3(256), 0, st, src, dst, n, chunk, pass * 8, w.counts);
        uint64_t* t = src; src = dst; dst = t;
    }
    hipLaunchKernelGGL(ood_measure_kernel, dim3(MEAS_BLOCKS), dim3(256), 0, st, src, w.pn, recall_level, w.s2, w.ap, w.cut);
    hipLaunchKernelGGL(ood_finalize_kernel, dim3(1), dim3(1), 0, st, w.pn, w.s2, w.ap, w.cut, MEAS_BLOCKS, result);
    DML_LAUNCH_CHECK();
    return 0;
}
