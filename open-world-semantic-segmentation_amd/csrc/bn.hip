// BatchNorm2d pieces for NHWC tensors on gfx950: statistics merge, affine(+residual)(+ReLU)(+dropout)
// apply, and the two-pass backward.  All HBM-bound: 16-byte vector accesses, fp32 math.
//
// Replaces nn.BatchNorm2d (112 instances: network/backbone/resnet.py:84-92,140,181; network/utils.py:13,21,
// 313,323,339,352), the ReLU / residual add at resnet.py:96-113 and nn.Dropout at network/utils.py:354.
#include "common.h"
#include <cstdlib>

namespace {

struct Moments {   // Chan et al. parallel-variance state
    double n, mean, m2;
};
__device__ __forceinline__ void merge(Moments& a, double nb, double meanb, double m2b) {
    if (nb <= 0.0) return;
    const double n = a.n + nb;
    const double d = meanb - a.mean;
    a.mean += d * (nb / n);
    a.m2 += m2b + d * d * (a.n * nb / n);
    a.n = n;
}

// partials[g][n] = (sum, M2 about the group mean) over rows [64 g, 64 g + 64)
// One pass (round 2; it was two dependent passes -- total sum for the mean, then M2 about it -- with two six-step LDS
// trees: 7.7 us for a kernel that reads ~1 MB, 85 times per step): every thread folds its groups into three fp64 sums
//   S = sum s_g,  Q = sum M2_g,  P = sum s_g^2 / n_g      ->      M2 = Q + P - S^2 / M
// (the identity of the two-stage finalize of the large maps below; in fp64 the cancellation costs ~1e-16 mean^2 / var),
// the 16 slices of a wave meet through DPP-free shuffles and the four waves through one LDS exchange.
constexpr int FIN_CH = 4, FIN_SL = 64;
__device__ __forceinline__ double shfl_xor_f64(double v, int mask) {
    const long long b = __double_as_longlong(v);
    const int lo = __shfl_xor((int)(b & 0xffffffffll), mask), hi = __shfl_xor((int)(b >> 32), mask);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- plane-scale bound folded into the finalize launches (round 6) ------------------------------------------------------------
// dml_h2_bound_bn / dml_h2_bound_bn_bwd were one-block launches right behind every finalize: a dependent hop of ~7 us, 145 times
// per step.  The finalize blocks now each raise state[0] to the largest bound term of their own channels and take a ticket
// (state[1]); the block that arrives LAST turns the maximum into the scale word work[1024] and resets both words for the next
// step.  max() is order-independent, so the result is the same whatever the arrival order (and equal to the stand-alone kernel's).
__device__ __forceinline__ float h2_words_max(const float* __restrict__ words, float* sh4) {
    uint32_t wm = 0;
    if (words != nullptr)
        for (int i = threadIdx.x; i < 1024; i += 256) wm = max(wm, __float_as_uint(words[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, o, 64));
    __syncthreads();                                   // (sh4 may still be read by an earlier use)
    if ((threadIdx.x & 63) == 0) sh4[threadIdx.x >> 6] = __uint_as_float(wm);
    __syncthreads();
    return __uint_as_float(max(max(__float_as_uint(sh4[0]), __float_as_uint(sh4[1])), max(__float_as_uint(sh4[2]), __float_as_uint(sh4[3]))));
}
__device__ __forceinline__ float h2_scale_of_bound(float b) {
    b *= 1.0009765625f;                                        // rounding of the statistics and of this sum
    const uint32_t m = __float_as_uint(b);
    float s = 1.0f;
    if (m != 0) {
        int se = 14 - ((int)(m >> 23) - 127);                  // scale = 2^(14 - floor(log2 b)): b * scale in [2^14, 2^15), as dml_h2_split
        se = se > 127 ? 127 : (se < -126 ? -126 : se);
        s = __uint_as_float((uint32_t)(se + 127) << 23);
        if (m >= 0x7f800000u) s = 1.0f;                        // the bound overflowed: the tensor is not finite either
    }
    return s;
}
// v: this thread's bound term (0 where it has none).  Forward: bound = max v * mult + max |res| (res_words, may be null);
// backward: bound = max v.  Every thread of every block of the (1-D) grid must call it.
__device__ __forceinline__ void h2_bound_tail(float v, uint32_t* __restrict__ state, float* __restrict__ work,
                                              const float* __restrict__ res_words, const float mult, const bool bwd) {
    __shared__ float shb[4];
    __shared__ int sh_last;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));      // (NaN terms: fmaxf drops them; the tensor carries them anyway)
    if ((threadIdx.x & 63) == 0) shb[threadIdx.x >> 6] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float b = fmaxf(fmaxf(shb[0], shb[1]), fmaxf(shb[2], shb[3]));
        const uint32_t old = __hip_atomic_fetch_max(state, __float_as_uint(b), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::"v"(old));                           // the maximum has been applied before the ticket is taken
        const uint32_t t = __hip_atomic_fetch_add(state + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = (t == gridDim.x - 1) ? 1 : 0;
    }
    __syncthreads();
    if (!sh_last) return;
    const float wmax = bwd ? 0.f : h2_words_max(res_words, shb);
    if (threadIdx.x != 0) return;
    float b = __uint_as_float(__hip_atomic_exchange(state, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    __hip_atomic_store(state + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // both words ready for the next step
    b = bwd ? b : b * mult + wmax;
    work[1024] = 1.0f / h2_scale_of_bound(b);
}
template <int FIN_CH, int FIN_SL>      // channels x group-slices per 256-thread block
__global__ __launch_bounds__(256) void bn_finalize_kernel(
    const float* __restrict__ partials, int64_t M, int N, const int SR, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
    float* scale, float* shift, float* save_mean, float* save_invstd, double* moments_out,
    uint32_t* bound_state = nullptr, float* bound_work = nullptr, const float* bound_res = nullptr, float bound_root = 0.f,
    float bound_mult = 1.f) {
    static_assert(FIN_CH == 4 && FIN_SL == 64, "lane = 4 channels x 16 slices per wave");
    __shared__ double sh[3][4][FIN_CH];
    const int ch = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
    const int n = blockIdx.x * FIN_CH + ch;
    const int64_t G = (M + SR - 1) / SR;              // SR: rows per statistics group (dml_conv_stat_rows)
    const int last_rows = (int)(M - (G - 1) * SR);
    double S = 0.0, Q = 0.0, P = 0.0;
    if (n < N)
        #pragma unroll 8
        for (int64_t g = sl; g < G; g += FIN_SL) {
            const float2 p = *reinterpret_cast<const float2*>(partials + (g * N + n) * 2);
            const double rows = g == G - 1 ? (double)last_rows : (double)SR;
            S += (double)p.x;
            Q += (double)p.y;
            P += (double)p.x * (double)p.x / rows;
        }
#pragma unroll
    for (int m = FIN_CH; m < 64; m <<= 1) {          // the wave's 16 slices of this channel: lanes ch + 4 k
        S += shfl_xor_f64(S, m);
        Q += shfl_xor_f64(Q, m);
        P += shfl_xor_f64(P, m);
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) < FIN_CH) {
        sh[0][wave][ch] = S;
        sh[1][wave][ch] = Q;
        sh[2][wave][ch] = P;
    }
    __syncthreads();
    float bterm = 0.f;                  // this channel's term of the plane-scale bound (h2_bound_tail)
    if (threadIdx.x < FIN_CH && n < N) {
        S = sh[0][0][ch] + sh[0][1][ch] + sh[0][2][ch] + sh[0][3][ch];
        Q = sh[1][0][ch] + sh[1][1][ch] + sh[1][2][ch] + sh[1][3][ch];
        P = sh[2][0][ch] + sh[2][1][ch] + sh[2][2][ch] + sh[2][3][ch];
        const double cnt = (double)M, mean = S / cnt;
        double m2 = Q + (P - S * mean);
        if (m2 < 0.0) m2 = 0.0;
        if (moments_out != nullptr) {       // synchronised BN: this rank's (mean, M2), merged later
            moments_out[2 * n] = mean;
            moments_out[2 * n + 1] = m2;
        } else {
            const double var_b = m2 / cnt;
            const float invstd = (float)(1.0 / sqrt(var_b + (double)eps));
            const float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
            const float sc = g * invstd;
            scale[n] = sc;
            shift[n] = b;
            save_mean[n] = (float)mean;
            if (save_invstd) save_invstd[n] = invstd;
            if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)mean;
            if (running_var) {
                const double var_u = cnt > 1.0 ? m2 / (cnt - 1.0) : var_b;
                running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)var_u;
            }
            bterm = fabsf(g) * bound_root + fabsf(b);
        }
    }
    if (bound_state != nullptr) h2_bound_tail(bterm, bound_state, bound_work, bound_res, bound_mult, false);
}

template <typename T>
__global__ __launch_bounds__(64) void bn_stats_kernel(const T* __restrict__ y, float* __restrict__ partials,
                                                      int64_t M, int N, int ldy) {
    const int n = blockIdx.y * 64 + threadIdx.x;
    const int64_t g = blockIdx.x;
    if (n >= N) return;
    const int64_t r0 = g * DML_STAT_ROWS;
    const int rows = (int)min((int64_t)DML_STAT_ROWS, M - r0);
    float s = 0.f;
    for (int r = 0; r < rows; ++r) s += Elem<T>::ld(y + (r0 + r) * ldy + n);
    const float mean = s / (float)rows;
    float m2 = 0.f;
    for (int r = 0; r < rows; ++r) {
        const float d = Elem<T>::ld(y + (r0 + r) * ldy + n) - mean;
        m2 += d * d;
    }
    partials[(g * N + n) * 2] = s;
    partials[(g * N + n) * 2 + 1] = m2;
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv,
                                      float eps, float* scale, float* shift, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    const float invstd = 1.0f / sqrtf(rv[n] + eps);
    scale[n] = (gamma ? gamma[n] : 1.f) * invstd;
    shift[n] = beta ? beta[n] : 0.f;          // z = (y - running_mean) * scale + shift
}

__global__ __launch_bounds__(256) void bn_eval_coeffs_table_kernel(const DmlBnEvalDesc* __restrict__ table) {
    const DmlBnEvalDesc d = table[blockIdx.x];
    for (int n = threadIdx.x; n < d.N; n += 256) {
        const float invstd = 1.0f / sqrtf(d.running_var[n] + d.eps);
        d.scale[n] = (d.gamma ? d.gamma[n] : 1.f) * invstd;
        d.shift[n] = d.beta ? d.beta[n] : 0.f;
    }
}

__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, uint32_t thresh) {
    uint64_t h = seed + idx * 0x9E3779B97F4A7C15ull;
    h ^= h >> 30; h *= 0xBF58476D1CE4E5B9ull;
    h ^= h >> 27; h *= 0x94D049BB133111EBull;
    h ^= h >> 31;
    return (uint32_t)(h >> 32) >= thresh;
}

// Column-stationary variant: a thread keeps one 16-byte channel vector (its scale / shift / mean live in registers,
// no per-element division) and walks rows, U rows per trip with all loads issued before the first use, so that a CU
// has U x 32 KB of HBM reads in flight instead of 32 KB.
// max |x| of a tensor as a side output of the kernel that writes it (dml_h2_split, amax_known): ONE atomic maximum per
// workgroup, spread over the 1024 words of the tensor's `work` buffer by workgroup index -- tens of thousands of short-lived
// workgroups raising a single word serialise in one L2 channel (measured: the apply kernels 56 -> 329 us); the split kernel
// takes the maximum over all 1024 words.  Order-independent, so the step stays bitwise reproducible.
__device__ __forceinline__ void amax_publish(uint32_t amx, uint32_t* __restrict__ words) {
    __shared__ uint32_t sh_amax[4];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) amx = max(amx, (uint32_t)__shfl_xor((int)amx, o, 64));
    if ((threadIdx.x & 63) == 0) sh_amax[threadIdx.x >> 6] = amx;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t m = sh_amax[0];
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) m = max(m, sh_amax[w]);      // (blocks of 1 .. 4 waves)
        if (m != 0) atomicMax(words + ((blockIdx.x + blockIdx.y * gridDim.x) & 1023u), m);
    }
}

// fp32 -> (hi, lo) fp16 planes of the power-of-two-scaled value (dml_h2_split's arithmetic), four elements: the BatchNorm apply
// kernels write a conv operand's planes themselves when the scale is known beforehand (dml_h2_bound_bn / _bn_bwd)
__device__ __forceinline__ void h2_store4(_Float16* __restrict__ hi_p, _Float16* __restrict__ lo_p, const float (&v)[4], const float s) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    h4 hi, lo;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float xs = v[e] * s;
        const _Float16 h = (_Float16)xs;
        hi[e] = h;
        lo[e] = (_Float16)(xs - (float)h);
    }
    *reinterpret_cast<h4*>(hi_p) = hi;
    *reinterpret_cast<h4*>(lo_p) = lo;
}
__device__ __forceinline__ void h2_store4(_Float16*, _Float16*, const float (&)[8], const float) {}      // (bf16 plans: no planes)

template <typename T, int U>
__global__ __launch_bounds__(256) void bn_apply_cols_kernel(
    const T* __restrict__ y, const T* __restrict__ res, T* __restrict__ z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, uint8_t* __restrict__ mask, int64_t M, int N,
    int ldy, int ldres, int ldz, int relu, float drop_p, uint64_t drop_seed, int CB, int RB, int rows_per_block,
    uint32_t* __restrict__ amax, _Float16* __restrict__ planes, int64_t plane_stride, int ldp, const float* __restrict__ unscale,
    const _Float16* __restrict__ res_planes, int64_t res_plane_stride, const float* __restrict__ res_unscale) {
    constexpr int V = Vec16<T>::N;
    const float h2s = planes != nullptr ? 1.0f / unscale[0] : 1.0f;      // exact: a power of two
    // the residual operand as the two fp16 planes of a tensor that exists as planes only (pitch ldres): value = (hi + lo) / s,
    // exact in fp32 (hi + lo has at most 23 significant bits, 1 / s is a power of two)
    const float rsu = res_planes != nullptr ? res_unscale[0] : 1.0f;
    const int NV = N / V;
    const int col = threadIdx.x % CB, rl = threadIdx.x / CB;
    const int vcol = blockIdx.y * CB + col;
    const bool act = rl < RB && vcol < NV;      // (idle threads stay for the wave reduction of amax at the end)
    if (!act && amax == nullptr) return;
    const int c = act ? vcol * V : 0;
    uint32_t amx = 0;                  // largest |z| this thread stores, as its bit pattern (dml_h2_split with amax_known)
    float sc[V], sh[V], mu[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { sc[q] = scale[c + q]; sh[q] = shift[c + q]; mu[q] = mean[c + q]; }
    const uint32_t thresh = drop_p > 0.f ? (uint32_t)min(4294967295.0, (double)drop_p * 4294967296.0) : 0u;
    const float keep_scale = drop_p > 0.f ? 1.0f / (1.0f - drop_p) : 1.0f;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = act ? min(M, r0 + rows_per_block) : 0;
    for (int64_t m0 = r0 + rl; m0 < r1; m0 += (int64_t)RB * U) {
        float v[U][V], r[U][V];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m < r1) {
                Vec16<T>::load(y + m * ldy + c, v[u]);
                if (res != nullptr) Vec16<T>::load(res + m * ldres + c, r[u]);
                if constexpr (V == 4) {
                    if (res_planes != nullptr) {
                        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
                        const h4 rh = *reinterpret_cast<const h4*>(res_planes + m * ldres + c);
                        const h4 rl4 = *reinterpret_cast<const h4*>(res_planes + res_plane_stride + m * ldres + c);
#pragma unroll
                        for (int q = 0; q < 4; ++q) r[u][q] = ((float)rh[q] + (float)rl4[q]) * rsu;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m >= r1) continue;
#pragma unroll
            for (int q = 0; q < V; ++q) v[u][q] = (v[u][q] - mu[q]) * sc[q] + sh[q];
            if (res != nullptr || res_planes != nullptr) {
#pragma unroll
                for (int q = 0; q < V; ++q) v[u][q] += r[u][q];
            }
            if (relu) {
#pragma unroll
                for (int q = 0; q < V; ++q) v[u][q] = v[u][q] > 0.f ? v[u][q] : 0.f;
            }
            if (drop_p > 0.f) {
#pragma unroll
                for (int q = 0; q < V; ++q)
                    v[u][q] = drop_keep(drop_seed, (uint64_t)m * N + c + q, thresh) ? v[u][q] * keep_scale : 0.f;
            }
            if (z != nullptr) Vec16<T>::store(z + m * ldz + c, v[u]);
            if (planes != nullptr) h2_store4(planes + m * ldp + c, planes + plane_stride + m * ldp + c, v[u], h2s);
            if (mask != nullptr) {           // one byte per 16-byte vector: 8 bits (bf16) / 4 bits (fp32)
                uint32_t bits = 0;
#pragma unroll
                for (int q = 0; q < V; ++q) bits |= (v[u][q] > 0.f ? 1u : 0u) << q;
                mask[m * NV + vcol] = (uint8_t)bits;
            }
            if (amax != nullptr) {
#pragma unroll
                for (int q = 0; q < V; ++q) amx = max(amx, __float_as_uint(v[u][q]) & 0x7fffffffu);
            }
        }
    }
    if (amax != nullptr) amax_publish(amx, amax);
}

// The same for the launches of an f16x2 plan whose output exists as planes ONLY (z == NULL: every conv1 / conv2 input and, since
// round 5, the block outputs): EIGHT channels per thread, so that every access is a 16-byte vector -- y as two float4, the
// planes (and a residual operand that is itself planes) as 8 x fp16 -- where the 4-channel kernel moves the planes in 8-byte
// pieces (0.54-0.70 x the 16-byte rate per instruction, MI355X_MICROARCH.md).  The mask stays one byte per four channels: two
// bytes per thread and row, one 2-byte store.
template <int U>
__global__ __launch_bounds__(256) void bn_apply_planes8_kernel(
    const float* __restrict__ y, const float* __restrict__ res, const _Float16* __restrict__ res_planes, int64_t res_plane_stride,
    const float* __restrict__ res_unscale, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ mean, uint8_t* __restrict__ mask, int64_t M, int N, int ldy, int ldres, int relu, int CB, int RB,
    int rows_per_block, uint32_t* __restrict__ amax, _Float16* __restrict__ planes, int64_t plane_stride, int ldp,
    const float* __restrict__ unscale) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int V = 8;
    const float h2s = 1.0f / unscale[0];      // exact: a power of two
    const float rsu = res_planes != nullptr ? res_unscale[0] : 1.0f;
    const int NV = N / V;
    const int col = threadIdx.x % CB, rl = threadIdx.x / CB;
    const int vcol = blockIdx.y * CB + col;
    const bool act = rl < RB && vcol < NV;
    if (!act && amax == nullptr) return;
    const int c = act ? vcol * V : 0;
    uint32_t amx = 0;
    float sc[V], sh[V], mu[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { sc[q] = scale[c + q]; sh[q] = shift[c + q]; mu[q] = mean[c + q]; }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = act ? min(M, r0 + rows_per_block) : 0;
    for (int64_t m0 = r0 + rl; m0 < r1; m0 += (int64_t)RB * U) {
        float4 va[U], vb[U], ra[U], rb[U];
        h8 rh[U], rlo[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m < r1) {
                va[u] = *reinterpret_cast<const float4*>(y + m * ldy + c);
                vb[u] = *reinterpret_cast<const float4*>(y + m * ldy + c + 4);
                if (res != nullptr) {
                    ra[u] = *reinterpret_cast<const float4*>(res + m * ldres + c);
                    rb[u] = *reinterpret_cast<const float4*>(res + m * ldres + c + 4);
                }
                if (res_planes != nullptr) {
                    rh[u] = *reinterpret_cast<const h8*>(res_planes + m * ldres + c);
                    rlo[u] = *reinterpret_cast<const h8*>(res_planes + res_plane_stride + m * ldres + c);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m >= r1) continue;
            float v[V] = {va[u].x, va[u].y, va[u].z, va[u].w, vb[u].x, vb[u].y, vb[u].z, vb[u].w};
#pragma unroll
            for (int q = 0; q < V; ++q) v[q] = (v[q] - mu[q]) * sc[q] + sh[q];
            if (res != nullptr) {
                const float r[V] = {ra[u].x, ra[u].y, ra[u].z, ra[u].w, rb[u].x, rb[u].y, rb[u].z, rb[u].w};
#pragma unroll
                for (int q = 0; q < V; ++q) v[q] += r[q];
            }
            if (res_planes != nullptr) {
#pragma unroll
                for (int q = 0; q < V; ++q) v[q] += ((float)rh[u][q] + (float)rlo[u][q]) * rsu;
            }
            if (relu) {
#pragma unroll
                for (int q = 0; q < V; ++q) v[q] = v[q] > 0.f ? v[q] : 0.f;
            }
            h8 hi, lo;
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const float xs = v[q] * h2s;
                const _Float16 h = (_Float16)xs;
                hi[q] = h;
                lo[q] = (_Float16)(xs - (float)h);
            }
            *reinterpret_cast<h8*>(planes + m * ldp + c) = hi;
            *reinterpret_cast<h8*>(planes + plane_stride + m * ldp + c) = lo;
            if (mask != nullptr) {           // one byte per four channels (the fp32 plans' layout): two bytes per thread
                uint32_t bits = 0;
#pragma unroll
                for (int q = 0; q < V; ++q) bits |= (v[q] > 0.f ? 1u : 0u) << (q + (q >= 4 ? 4 : 0));
                *reinterpret_cast<uint16_t*>(mask + m * (N / 4) + vcol * 2) = (uint16_t)bits;
            }
            if (amax != nullptr) {
#pragma unroll
                for (int q = 0; q < V; ++q) amx = max(amx, __float_as_uint(v[q]) & 0x7fffffffu);
            }
        }
    }
    if (amax != nullptr) amax_publish(amx, amax);
}

// geometry shared by the column-stationary kernels: CB vector columns x RB row lanes per 256-thread block
struct ColGeom { int CB, RB, col_chunks, rows_per_block, row_blocks; };
static ColGeom col_geom(int64_t M, int NV, int U, int target_blocks, const int threads = 256) {
    ColGeom g;
    g.CB = NV < threads ? NV : threads;
    g.RB = threads / g.CB;
    target_blocks *= 256 / threads;
    g.col_chunks = (NV + g.CB - 1) / g.CB;
    const int64_t step = (int64_t)g.RB * U;
    int64_t rpb = (M * g.col_chunks + target_blocks - 1) / target_blocks;
    rpb = ((rpb + step - 1) / step) * step;
    if (rpb < step) rpb = step;
    g.rows_per_block = (int)rpb;
    g.row_blocks = (int)((M + rpb - 1) / rpb);
    return g;
}
// Workgroups for a streaming pass over an M x N tensor: ~16 KB of the tensor per workgroup (two trips of U = 2 rows
// per row lane).  Measured with tools/bench_bn.py on MI355X: short-lived workgroups that sweep the tensor front to
// back beat 2048 persistent grid-stride workgroups by 15-30 % (4.3 -> 5.1-5.7 TB/s on the >= 75 MB tensors);
// the small layer3 tensors (19 MB) prefer ~1024.
static int stream_blocks(int64_t M, int N, int dtype) {
    const int64_t bytes = M * N * (dtype == DML_BF16 ? 2 : 4);
    int64_t b = bytes / 16384;
    if (b < 1024) b = 1024;
    if (b > 32768) b = 32768;
    return (int)b;
}

// ---------------------------------------------------------------------------------- backward
// pass 1: partials[rb][n] = (sum g, sum g * xhat),  g = dz * [z > 0] * gscale
constexpr int RED_COLS = 64;   // vector columns per block
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(
    const T* __restrict__ dz, const T* __restrict__ y, const T* __restrict__ z, const uint8_t* __restrict__ mask,
    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, float* __restrict__ partials,
    int64_t M, int N, int lddz, int ldy, int ldz, int relu, float gscale, int rows_per_block, int chv, int rt,
    uint32_t* __restrict__ gmax) {
    constexpr int V = Vec16<T>::N;
    __shared__ float sh[256 * 2 * V];
    uint32_t gmx = 0;                  // largest |g| this thread has seen (dml_h2_bound_bn_bwd)
    const int NV = N / V;
    const int col = threadIdx.x % chv, rl = threadIdx.x / chv;
    const int v = blockIdx.y * chv + col;
    const bool active = rl < rt && v < NV;
    float sg[V], sgx[V];
#pragma unroll
    for (int q = 0; q < V; ++q) { sg[q] = 0.f; sgx[q] = 0.f; }
    if (active) {
        const int c = v * V;
        float mu[V], is[V];
#pragma unroll
        for (int q = 0; q < V; ++q) { mu[q] = save_mean[c + q]; is[q] = save_invstd[c + q]; }
        const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
        const int64_t r1 = min(M, r0 + rows_per_block);
        constexpr int U = 2;                  // rows in flight per thread
        const bool use_mask = mask != nullptr;
        for (int64_t m0 = r0 + rl; m0 < r1; m0 += (int64_t)rt * U) {
            float g[U][V], yy[U][V], zz[U][V];
            uint32_t bits[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = m0 + (int64_t)u * rt;
                if (m < r1) {
                    Vec16<T>::load(dz + m * lddz + c, g[u]);
                    Vec16<T>::load(y + m * ldy + c, yy[u]);
                    if (relu) {
                        if (use_mask) bits[u] = mask[m * NV + v];
                        else Vec16<T>::load(z + m * ldz + c, zz[u]);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t m = m0 + (int64_t)u * rt;
                if (m >= r1) continue;
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    const bool on = !relu || (use_mask ? ((bits[u] >> q) & 1u) != 0 : zz[u][q] > 0.f);
                    const float gg = on ? g[u][q] * gscale : 0.f;
                    gmx = max(gmx, __float_as_uint(gg) & 0x7fffffffu);
                    sg[q] += gg;
                    sgx[q] += gg * (yy[u][q] - mu[q]) * is[q];
                }
            }
        }
    }
    // reduce over the rt row lanes that share a column
#pragma unroll
    for (int q = 0; q < V; ++q) {
        sh[(threadIdx.x * V + q) * 2] = sg[q];
        sh[(threadIdx.x * V + q) * 2 + 1] = sgx[q];
    }
    __syncthreads();
    if (rl == 0 && v < NV) {
        for (int r = 1; r < rt; ++r) {
            const int t = r * chv + col;
#pragma unroll
            for (int q = 0; q < V; ++q) {
                sg[q] += sh[(t * V + q) * 2];
                sgx[q] += sh[(t * V + q) * 2 + 1];
            }
        }
        float* p = partials + ((int64_t)blockIdx.x * N + v * V) * 2;
#pragma unroll
        for (int q = 0; q < V; ++q) { p[2 * q] = sg[q]; p[2 * q + 1] = sgx[q]; }
    }
    if (gmax != nullptr) amax_publish(gmx, gmax);
}

// many partial rows (the 64-row groups of a fused reduce on the 192 x 192 layers: 9216): chunks of R rows are summed
// first, each chunk into ITS OWN first row (nobody else touches those columns of that row), and the finalize then
// walks the chunk heads with row stride R -- the same two coalesced stages as the forward statistics
// (32 VGPRs: see bn_bwd_finalize_kernel)
__global__ __launch_bounds__(256) void bn_bwd_fold_kernel(float* partials, int G, int N, int R) {
    __shared__ double sh[2][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int g0 = blockIdx.y * R, g1 = min(G, g0 + R);
    double s0 = 0.0, s1 = 0.0;
    if (n < N) {
        const uint32_t step = 8u * (uint32_t)N;                                        // partials hold < 2^31 floats
        uint32_t off = ((uint32_t)(g0 + rl) * (uint32_t)N + (uint32_t)n) * 2u;
        #pragma unroll 4
        for (int g = g0 + rl; g < g1; g += 4, off += step) {
            const float2 p = *reinterpret_cast<const float2*>(partials + off);
            s0 += (double)p.x; s1 += (double)p.y;
        }
    }
    sh[0][rl][c] = s0; sh[1][rl][c] = s1;
    __syncthreads();
    if (rl == 0 && n < N) {
        const double a = sh[0][0][c] + sh[0][1][c] + sh[0][2][c] + sh[0][3][c];
        const double b = sh[1][0][c] + sh[1][1][c] + sh[1][2][c] + sh[1][3][c];
        *reinterpret_cast<float2*>(partials + ((int64_t)g0 * N + n) * 2) = make_float2((float)a, (float)b);
    }
}

// At most 32 VGPRs, deliberately: this kernel sits on the backward's critical path while the weight-gradient kernel of the
// side stream holds every CU with one 8-wave workgroup of 235 registers (2 x 240 of a SIMD's 512); a 4-wave block that
// needs <= 32 registers per lane still fits beside it and starts at once, a larger one waits for a weight-gradient
// workgroup to retire (measured: 6 us -> 33 us per launch, 112 launches per step).  (The compiler drops an
// amdgpu_num_vgpr request below what eight waves per SIMD allow, so the loops are written to need few: 32-bit element
// offsets from a scalar base, four loads in flight.)
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(
    const float* __restrict__ partials, int nblocks, int64_t M, int N, const float* __restrict__ gamma,
    const float* __restrict__ save_mean, const float* __restrict__ save_invstd, float* dgamma, float* dbeta,
    float* coef, double* sums_out, int row_stride,
    uint32_t* bound_state = nullptr, float* bound_work = nullptr, const float* bound_gmax = nullptr, float bound_root = 0.f) {
    __shared__ double sh[2][4][FIN_CH];
    const int ch = threadIdx.x % FIN_CH, sl = threadIdx.x / FIN_CH;
    const int n = blockIdx.x * FIN_CH + ch;
    double s0 = 0.0, s1 = 0.0;
    if (n < N) {
        const uint32_t step = (uint32_t)(FIN_SL * row_stride) * (uint32_t)N * 2u;       // partials hold < 2^31 floats
        uint32_t off = ((uint32_t)(sl * row_stride) * (uint32_t)N + (uint32_t)n) * 2u;
        #pragma unroll 4
        for (int g = sl; g < nblocks; g += FIN_SL, off += step) {
            const float2 p = *reinterpret_cast<const float2*>(partials + off);
            s0 += p.x; s1 += p.y;
        }
    }
#pragma unroll
    for (int m = FIN_CH; m < 64; m <<= 1) {          // the wave's 16 slices of this channel (see bn_finalize_kernel)
        s0 += shfl_xor_f64(s0, m);
        s1 += shfl_xor_f64(s1, m);
    }
    if ((threadIdx.x & 63) < FIN_CH) {
        sh[0][threadIdx.x >> 6][ch] = s0;
        sh[1][threadIdx.x >> 6][ch] = s1;
    }
    __syncthreads();
    if (sl == 0) {
        sh[0][0][ch] = sh[0][0][ch] + sh[0][1][ch] + sh[0][2][ch] + sh[0][3][ch];
        sh[1][0][ch] = sh[1][0][ch] + sh[1][1][ch] + sh[1][2][ch] + sh[1][3][ch];
    }
    if (sl == 0 && n < N && sums_out != nullptr) {          // synchronised BN: local sums out, parameter gradients local
        sums_out[2 * n] = sh[0][0][ch];
        sums_out[2 * n + 1] = sh[1][0][ch];
        if (dgamma) dgamma[n] += (float)sh[1][0][ch];
        if (dbeta) dbeta[n] += (float)sh[0][0][ch];
        return;
    }
    float bterm = 0.f;                  // |A| max|g| + |Bc| sqrt(count) / invstd + |C0|: this channel's bound on |dy| (h2_bound_tail)
    float gmax = 0.f;
    if (bound_state != nullptr) {       // (uniform; the sums are in LDS, the float staging below may be reused)
        __shared__ float shg[4];
        gmax = h2_words_max(bound_gmax, shg);
    }
    if (sl == 0 && n < N) {
        const double dbeta_s = sh[0][0][ch], dgamma_s = sh[1][0][ch];
        const double g = gamma ? (double)gamma[n] : 1.0, is = save_invstd[n], mu = save_mean[n];
        const double A = g * is;
        // M == 0: the layer normalised with FIXED (running) statistics -- they do not depend on the batch, so the two
        // mean-correction terms of the batch-statistics backward vanish and dy = gamma * invstd * g
        const double Bc = M > 0 ? -A * is * dgamma_s / (double)M : 0.0;
        const double C0 = M > 0 ? -A * dbeta_s / (double)M : 0.0;
        coef[n] = (float)A; coef[N + n] = (float)Bc; coef[2 * N + n] = (float)C0; coef[3 * N + n] = (float)mu;
        if (dgamma) dgamma[n] += (float)dgamma_s;
        if (dbeta) dbeta[n] += (float)dbeta_s;
        bterm = fabsf((float)A) * gmax + fabsf((float)Bc) * bound_root / save_invstd[n] + fabsf((float)C0);
    }
    if (bound_state != nullptr) h2_bound_tail(bterm, bound_state, bound_work, nullptr, 1.f, true);
}

// column-stationary backward apply (see bn_apply_cols_kernel): coefficients in registers, U rows in flight
template <typename T, int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_cols_kernel(
    const T* __restrict__ dz, const T* __restrict__ y, const T* __restrict__ z, const uint8_t* __restrict__ mask,
    const float* __restrict__ coef, T* __restrict__ dy, T* dres, int64_t M, int N, int lddz, int ldy, int ldz, int lddy,
    int lddres, int relu, float gscale, int dres_accum, int CB, int RB, int rows_per_block, uint32_t* __restrict__ amax,
    _Float16* __restrict__ planes, int64_t plane_stride, int ldp, const float* __restrict__ unscale) {
    constexpr int V = Vec16<T>::N;
    const float h2s = planes != nullptr ? 1.0f / unscale[0] : 1.0f;
    const int NV = N / V;
    const int col = threadIdx.x % CB, rl = threadIdx.x / CB;
    const int vcol = blockIdx.y * CB + col;
    const bool act = rl < RB && vcol < NV;      // (idle threads stay for the wave reduction of amax at the end)
    if (!act && amax == nullptr) return;
    const int c = act ? vcol * V : 0;
    uint32_t amx = 0;                  // largest |dy| this thread stores (dml_h2_split with amax_known)
    float cA[V], cB[V], cC[V], cM[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        cA[q] = coef[c + q] * gscale;
        cB[q] = coef[N + c + q];
        cC[q] = coef[2 * N + c + q];
        cM[q] = coef[3 * N + c + q];
    }
    const bool use_mask = mask != nullptr;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = act ? min(M, r0 + rows_per_block) : 0;
    for (int64_t m0 = r0 + rl; m0 < r1; m0 += (int64_t)RB * U) {
        float g[U][V], yy[U][V], zz[U][V], rr[U][V];
        uint32_t bits[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m < r1) {
                Vec16<T>::load(dz + m * lddz + c, g[u]);
                Vec16<T>::load(y + m * ldy + c, yy[u]);
                if (relu) {
                    if (use_mask) bits[u] = mask[m * NV + vcol];
                    else Vec16<T>::load(z + m * ldz + c, zz[u]);
                }
                if (dres != nullptr && dres_accum) Vec16<T>::load(dres + m * lddres + c, rr[u]);
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m >= r1) continue;
            if (relu) {
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    const bool on = use_mask ? ((bits[u] >> q) & 1u) != 0 : zz[u][q] > 0.f;
                    g[u][q] = on ? g[u][q] : 0.f;
                }
            }
            float o[V];
#pragma unroll
            for (int q = 0; q < V; ++q) o[q] = cA[q] * g[u][q] + cB[q] * (yy[u][q] - cM[q]) + cC[q];
            if (dy != nullptr) Vec16<T>::store(dy + m * lddy + c, o);
            if (planes != nullptr) h2_store4(planes + m * ldp + c, planes + plane_stride + m * ldp + c, o, h2s);
            if (amax != nullptr) {
#pragma unroll
                for (int q = 0; q < V; ++q) amx = max(amx, __float_as_uint(o[q]) & 0x7fffffffu);
            }
            if (dres != nullptr) {
#pragma unroll
                for (int q = 0; q < V; ++q) g[u][q] = g[u][q] * gscale + (dres_accum ? rr[u][q] : 0.f);
                Vec16<T>::store(dres + m * lddres + c, g[u]);
            }
        }
    }
    if (amax != nullptr) amax_publish(amx, amax);
}

// The launches of an f16x2 plan whose dy exists as planes only (every unit whose weight and data gradient read planes), ReLU from the
// bit mask: eight channels per thread -- dz and y as two float4 each, the mask as ONE 2-byte load, the planes as 8 x fp16 (16-byte
// stores; bn_apply_planes8_kernel is the forward twin).  Optionally the identity branch's masked gradient dres (fp32, set or accumulate).
template <int U>
__global__ __launch_bounds__(256) void bn_bwd_apply_planes8_kernel(
    const float* __restrict__ dz, const float* __restrict__ y, const uint8_t* __restrict__ mask, const float* __restrict__ coef,
    float* dres, int64_t M, int N, int lddz, int ldy, int lddres, int relu, float gscale, int dres_accum, int CB, int RB,
    int rows_per_block, _Float16* __restrict__ planes, int64_t plane_stride, int ldp, const float* __restrict__ unscale) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    constexpr int V = 8;
    const float h2s = 1.0f / unscale[0];
    const int NV = N / V;
    const int col = threadIdx.x % CB, rl = threadIdx.x / CB;
    const int vcol = blockIdx.y * CB + col;
    if (!(rl < RB && vcol < NV)) return;
    const int c = vcol * V;
    float cA[V], cB[V], cC[V], cM[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        cA[q] = coef[c + q] * gscale;
        cB[q] = coef[N + c + q];
        cC[q] = coef[2 * N + c + q];
        cM[q] = coef[3 * N + c + q];
    }
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
    const int64_t r1 = min(M, r0 + rows_per_block);
    for (int64_t m0 = r0 + rl; m0 < r1; m0 += (int64_t)RB * U) {
        float4 ga[U], gb[U], ya[U], yb[U], ra[U], rb[U];
        uint32_t bits[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            bits[u] = 0xffffu;
            if (m < r1) {
                ga[u] = *reinterpret_cast<const float4*>(dz + m * lddz + c);
                gb[u] = *reinterpret_cast<const float4*>(dz + m * lddz + c + 4);
                ya[u] = *reinterpret_cast<const float4*>(y + m * ldy + c);
                yb[u] = *reinterpret_cast<const float4*>(y + m * ldy + c + 4);
                if (relu) bits[u] = *reinterpret_cast<const uint16_t*>(mask + m * (N / 4) + vcol * 2);
                if (dres != nullptr && dres_accum) {
                    ra[u] = *reinterpret_cast<const float4*>(dres + m * lddres + c);
                    rb[u] = *reinterpret_cast<const float4*>(dres + m * lddres + c + 4);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t m = m0 + (int64_t)u * RB;
            if (m >= r1) continue;
            float g[V] = {ga[u].x, ga[u].y, ga[u].z, ga[u].w, gb[u].x, gb[u].y, gb[u].z, gb[u].w};
            const float yy[V] = {ya[u].x, ya[u].y, ya[u].z, ya[u].w, yb[u].x, yb[u].y, yb[u].z, yb[u].w};
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const bool on = ((bits[u] >> (q + (q >= 4 ? 4 : 0))) & 1u) != 0;      // one mask byte per four channels
                g[q] = on ? g[q] : 0.f;
            }
            h8 hi, lo;
#pragma unroll
            for (int q = 0; q < V; ++q) {
                const float o = cA[q] * g[q] + cB[q] * (yy[q] - cM[q]) + cC[q];
                const float xs = o * h2s;
                const _Float16 h = (_Float16)xs;
                hi[q] = h;
                lo[q] = (_Float16)(xs - (float)h);
            }
            *reinterpret_cast<h8*>(planes + m * ldp + c) = hi;
            *reinterpret_cast<h8*>(planes + plane_stride + m * ldp + c) = lo;
            if (dres != nullptr) {
                float r[V] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                if (dres_accum) { r[0] = ra[u].x; r[1] = ra[u].y; r[2] = ra[u].z; r[3] = ra[u].w; r[4] = rb[u].x; r[5] = rb[u].y; r[6] = rb[u].z; r[7] = rb[u].w; }
#pragma unroll
                for (int q = 0; q < V; ++q) r[q] += g[q] * gscale;
                *reinterpret_cast<float4*>(dres + m * lddres + c) = make_float4(r[0], r[1], r[2], r[3]);
                *reinterpret_cast<float4*>(dres + m * lddres + c + 4) = make_float4(r[4], r[5], r[6], r[7]);
            }
        }
    }
}

// Large feature maps (G >= 2048 row groups): two coalesced stages instead of one strided walk per channel.
// Stage 1: block (channel tile of 64, chunk of R groups) reads its rows 512 B per wave and folds them to three
// fp64 sums per channel -- S = sum s_g, Q = sum M2_g, P = sum s_g^2 / n_g -- which it writes IN PLACE over the
// first three rows of its own chunk (row = N float2 = N doubles; nobody else reads those columns of those rows).
// Stage 2: per channel, sum the chunks: M2 = Q + P - S^2 / M (fp64: the cancellation costs ~1e-16 * mean^2/var).
constexpr int FIN2_MAXCHUNK = 128;
__global__ __launch_bounds__(256) void bn_fold_partials_kernel(float* partials, int64_t G, int N, int R, int NC,
                                                               int last_rows, const int SR) {
    __shared__ double sh[3][4][64];
    const int c = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + c;
    const int64_t g0 = (int64_t)blockIdx.y * R;
    const int64_t g1 = (int)blockIdx.y == NC - 1 ? G : g0 + R;
    double S = 0.0, Q = 0.0, P = 0.0;
    if (n < N)
        #pragma unroll 8
        for (int64_t g = g0 + rl; g < g1; g += 4) {
            const float2 p = *reinterpret_cast<const float2*>(partials + (g * N + n) * 2);
            const double rows = g == G - 1 ? (double)last_rows : (double)SR;
            S += (double)p.x;
            Q += (double)p.y;
            P += (double)p.x * (double)p.x / rows;
        }
    sh[0][rl][c] = S;
    sh[1][rl][c] = Q;
    sh[2][rl][c] = P;
    __syncthreads();     // every read of this block's rows is done before they are overwritten
    if (rl < 3 && n < N) {
        const double v = sh[rl][0][c] + sh[rl][1][c] + sh[rl][2][c] + sh[rl][3][c];
        reinterpret_cast<double*>(partials)[(g0 + rl) * N + n] = v;
    }
}

__global__ __launch_bounds__(256) void bn_finalize_folded_kernel(
    const float* __restrict__ partials, int64_t M, int N, int R, int NC, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps,
    float* scale, float* shift, float* save_mean, float* save_invstd,
    uint32_t* bound_state = nullptr, float* bound_work = nullptr, const float* bound_res = nullptr, float bound_root = 0.f,
    float bound_mult = 1.f) {
    // 8 channels x 32 chunk lanes per block (it was 32 x 8: two blocks for the 64-channel layers, every thread walking 16
    // chunks one after the other -- 14 us): the NC (<= 128) chunk sums are read 32 at a time, N / 8 blocks
    constexpr int FC = 8, FL = 32;
    __shared__ double sh[3][FL][FC];
    const int c = threadIdx.x % FC, cl = threadIdx.x / FC;
    const int n = blockIdx.x * FC + c;
    const double* f = reinterpret_cast<const double*>(partials);
    double S = 0.0, Q = 0.0, P = 0.0;
    if (n < N)
        #pragma unroll 4
        for (int ck = cl; ck < NC; ck += FL) {
            const int64_t g0 = (int64_t)ck * R;
            S += f[(g0 + 0) * N + n];
            Q += f[(g0 + 1) * N + n];
            P += f[(g0 + 2) * N + n];
        }
    sh[0][cl][c] = S;
    sh[1][cl][c] = Q;
    sh[2][cl][c] = P;
    __syncthreads();
    float bterm = 0.f;
    if (cl == 0 && n < N) {
        S = Q = P = 0.0;
#pragma unroll
        for (int k = 0; k < FL; ++k) {
            S += sh[0][k][c];
            Q += sh[1][k][c];
            P += sh[2][k][c];
        }
        const double cnt = (double)M, mean = S / cnt;
        double m2 = Q + (P - S * mean);
        if (m2 < 0.0) m2 = 0.0;
        const double var_b = m2 / cnt;
        const float invstd = (float)(1.0 / sqrt(var_b + (double)eps));
        const float g = gamma ? gamma[n] : 1.f, b = beta ? beta[n] : 0.f;
        scale[n] = g * invstd;
        shift[n] = b;
        save_mean[n] = (float)mean;
        if (save_invstd) save_invstd[n] = invstd;
        if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)mean;
        if (running_var) {
            const double var_u = cnt > 1.0 ? m2 / (cnt - 1.0) : var_b;
            running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)var_u;
        }
        bterm = fabsf(g) * bound_root + fabsf(b);
    }
    if (bound_state != nullptr) h2_bound_tail(bterm, bound_state, bound_work, bound_res, bound_mult, false);
}


inline bool vec_ok(int dtype, int a) { return a % (dtype == DML_BF16 ? 8 : 4) == 0; }

}  // namespace

static int bn_finalize_impl(float* partials, int64_t M, int N, int stat_rows, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps,
                            float* scale, float* shift, float* save_mean, float* save_invstd, void* stream,
                            uint32_t* bstate, float* bwork, const float* bres, float broot, float bmult);

extern "C" int dml_bn_finalize(float* partials, int64_t M, int N, int stat_rows, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float momentum, float eps,
                               float* scale, float* shift, float* save_mean, float* save_invstd, void* stream) {
    return bn_finalize_impl(partials, M, N, stat_rows, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                            save_invstd, stream, nullptr, nullptr, nullptr, 0.f, 1.f);
}

// dml_bn_finalize + dml_h2_bound_bn(gamma, beta, N, count, mult, res_amax, work) in ONE launch: the finalize block that arrives
// last writes the plane scale (h2_bound_tail).  `state`: two zero-initialised words owned by this BatchNorm, left zero again.
extern "C" int dml_bn_finalize_bound(float* partials, int64_t M, int N, int stat_rows, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps,
                                     float* scale, float* shift, float* save_mean, float* save_invstd,
                                     int64_t count, float mult, const float* res_amax, float* work, uint32_t* state, void* stream) {
    if (!work || !state || count <= 0 || !(mult > 0.f)) return DML_EINVAL;
    return bn_finalize_impl(partials, M, N, stat_rows, gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean,
                            save_invstd, stream, state, work, res_amax, sqrtf((float)count) * 1.0001f, mult);
}

static int bn_finalize_impl(float* partials, int64_t M, int N, int stat_rows, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps,
                            float* scale, float* shift, float* save_mean, float* save_invstd, void* stream,
                            uint32_t* bstate, float* bwork, const float* bres, float broot, float bmult) {
    if (!partials || !scale || !shift || !save_mean || M <= 0 || N <= 0 || stat_rows <= 0) return DML_EINVAL;
    const int64_t G = (M + stat_rows - 1) / stat_rows;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (G >= 2048) {
        // large feature maps (the 192x192 layers have G = 9216 and as few as 64 channels): fold, then finish
        if ((reinterpret_cast<uintptr_t>(partials) & 7) != 0) return DML_EALIGN;
        const int R = (int)((G + FIN2_MAXCHUNK - 1) / FIN2_MAXCHUNK);      // >= 16 rows per chunk
        const int NC = (int)(G / R);                                         // the last chunk takes the remainder
        const int last_rows = (int)(M - (G - 1) * stat_rows);
        hipLaunchKernelGGL(bn_fold_partials_kernel, dim3((N + 63) / 64, NC), dim3(256), 0, st, partials, G, N, R, NC,
                           last_rows, stat_rows);
        hipLaunchKernelGGL(bn_finalize_folded_kernel, dim3((N + 7) / 8), dim3(256), 0, st, partials, M, N, R, NC,
                           gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd,
                           bstate, bwork, bres, broot, bmult);
    } else {
        hipLaunchKernelGGL((bn_finalize_kernel<4, 64>), dim3((N + 3) / 4), dim3(256), 0, st, partials, M, N, stat_rows,
                           gamma, beta, running_mean, running_var, momentum, eps, scale, shift, save_mean, save_invstd,
                           (double*)nullptr, bstate, bwork, bres, broot, bmult);
    }
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_stats(const void* y, float* partials, int64_t M, int N, int ldy, int dtype, void* stream) {
    if (!y || !partials || M <= 0 || N <= 0) return DML_EINVAL;
    dim3 grid((unsigned)((M + DML_STAT_ROWS - 1) / DML_STAT_ROWS), (N + 63) / 64);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == DML_BF16)
        hipLaunchKernelGGL(bn_stats_kernel<bf16_t>, grid, dim3(64), 0, st, (const bf16_t*)y, partials, M, N, ldy);
    else
        hipLaunchKernelGGL(bn_stats_kernel<float>, grid, dim3(64), 0, st, (const float*)y, partials, M, N, ldy);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                  const float* running_var, float eps, float* scale, float* shift, int N,
                                  void* stream) {
    if (!running_mean || !running_var || !scale || !shift || N <= 0) return DML_EINVAL;
    hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((N + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), gamma, beta, running_mean, running_var, eps, scale,
                       shift, N);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_eval_coeffs_table(const DmlBnEvalDesc* table, int count, void* stream) {
    if (!table || count <= 0) return DML_EINVAL;
    hipLaunchKernelGGL(bn_eval_coeffs_table_kernel, dim3(count), dim3(256), 0, static_cast<hipStream_t>(stream), table);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_apply(const void* y, const void* res, void* z, const float* scale, const float* shift,
                            const float* mean, uint8_t* mask, int64_t M, int N, int ldy, int ldres, int ldz, int relu,
                            int dtype, float drop_p, uint64_t drop_seed, float* amax, void* planes, int64_t plane_stride,
                            int32_t ldp, const float* unscale, int64_t res_plane_stride, const float* res_unscale, void* stream) {
    if (!y || (!z && !planes) || !scale || !shift || !mean || M <= 0 || N <= 0) return DML_EINVAL;
    // res_unscale != NULL: `res` points at the fp16 planes (hi, then lo `res_plane_stride` elements further, pitch ldres) of a
    // residual tensor that exists as planes only
    const void* res_pl = nullptr;
    if (res_unscale) {
        if (dtype != DML_F32 || !res || res_plane_stride <= 0 || (res_plane_stride & 3) || (ldres & 3) ||
            (reinterpret_cast<uintptr_t>(res) & 7))
            return DML_EINVAL;
        res_pl = res;
        res = nullptr;
    }
    if (!vec_ok(dtype, N) || !vec_ok(dtype, ldy) || (z && !vec_ok(dtype, ldz)) || (res && !vec_ok(dtype, ldres)))
        return DML_EALIGN;
    if (M >= (1ll << 31)) return DML_EINVAL;
    if (planes) {           // fp32 only: the output also (or only: z == NULL) as two fp16 planes scaled by 1 / unscale[0]
        if (dtype != DML_F32 || !unscale || plane_stride <= 0 || ldp < N) return DML_EINVAL;
        if ((ldp & 3) || (plane_stride & 3) || (reinterpret_cast<uintptr_t>(planes) & 7)) return DML_EALIGN;
    }
    const int V = dtype == DML_BF16 ? 8 : 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == DML_F32 && planes && !z && drop_p == 0.f && (N & 7) == 0 && (ldy & 3) == 0 && (ldp & 7) == 0 && (plane_stride & 7) == 0 &&
        (reinterpret_cast<uintptr_t>(planes) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0 &&
        (!res || ((ldres & 3) == 0 && (reinterpret_cast<uintptr_t>(res) & 15) == 0)) &&
        (!res_pl || ((ldres & 7) == 0 && (res_plane_stride & 7) == 0 && (reinterpret_cast<uintptr_t>(res_pl) & 15) == 0))) {
        // planes-only output: eight channels per thread, 16-byte accesses throughout
        const ColGeom g8 = col_geom(M, N / 8, 2, stream_blocks(M, N, dtype));
        hipLaunchKernelGGL((bn_apply_planes8_kernel<2>), dim3(g8.row_blocks, g8.col_chunks), dim3(256), 0, st, (const float*)y,
                           (const float*)res, static_cast<const _Float16*>(res_pl), res_plane_stride, res_unscale, scale, shift, mean,
                           mask, M, N, ldy, ldres, relu, g8.CB, g8.RB, g8.rows_per_block, reinterpret_cast<uint32_t*>(amax),
                           static_cast<_Float16*>(planes), plane_stride, (int)ldp, unscale);
        DML_LAUNCH_CHECK();
        return 0;
    }
    const ColGeom g = col_geom(M, N / V, 2, stream_blocks(M, N, dtype));
    dim3 grid(g.row_blocks, g.col_chunks);
    if (dtype == DML_BF16)
        hipLaunchKernelGGL((bn_apply_cols_kernel<bf16_t, 2>), grid, dim3(256), 0, st, (const bf16_t*)y, (const bf16_t*)res,
                           (bf16_t*)z, scale, shift, mean, mask, M, N, ldy, ldres, ldz, relu, drop_p, drop_seed, g.CB, g.RB,
                           g.rows_per_block, reinterpret_cast<uint32_t*>(amax), (_Float16*)nullptr, (int64_t)0, 0,
                           (const float*)nullptr, (const _Float16*)nullptr, (int64_t)0, (const float*)nullptr);
    else
        hipLaunchKernelGGL((bn_apply_cols_kernel<float, 2>), grid, dim3(256), 0, st, (const float*)y, (const float*)res,
                           (float*)z, scale, shift, mean, mask, M, N, ldy, ldres, ldz, relu, drop_p, drop_seed, g.CB, g.RB,
                           g.rows_per_block, reinterpret_cast<uint32_t*>(amax), static_cast<_Float16*>(planes), plane_stride,
                           (int)ldp, unscale, static_cast<const _Float16*>(res_pl), res_plane_stride, res_unscale);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_bwd_reduce(const void* dz, const void* y, const void* z, const uint8_t* mask, const float* save_mean,
                                 const float* save_invstd, float* partials, int64_t M, int N, int lddz, int ldy,
                                 int ldz, int relu, float gscale, int dtype, int* nblocks, float* gmax, void* stream) {
    if (!dz || !y || !save_mean || !save_invstd || !partials || !nblocks || M <= 0 || N <= 0) return DML_EINVAL;
    if (relu && !z && !mask) return DML_EINVAL;
    if (!vec_ok(dtype, N) || !vec_ok(dtype, lddz) || !vec_ok(dtype, ldy) || (relu && !mask && !vec_ok(dtype, ldz)))
        return DML_EALIGN;
    const int V = dtype == DML_BF16 ? 8 : 4;
    const int NV = N / V;
    const int chv = NV < RED_COLS ? NV : RED_COLS;
    const int rt = 256 / chv;
    const int col_chunks = (NV + chv - 1) / chv;
    int64_t rpb = (M * col_chunks + 1023) / 1024;
    if (rpb < 4 * rt) rpb = 4 * rt;
    rpb = ((rpb + rt - 1) / rt) * rt;
    const int rb = (int)((M + rpb - 1) / rpb);
    *nblocks = rb;
    dim3 grid(rb, col_chunks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == DML_BF16)
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dz,
                           (const bf16_t*)y, (const bf16_t*)z, mask, save_mean, save_invstd, partials, M, N, lddz, ldy,
                           ldz, relu, gscale, (int)rpb, chv, rt, reinterpret_cast<uint32_t*>(gmax));
    else
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, grid, dim3(256), 0, st, (const float*)dz,
                           (const float*)y, (const float*)z, mask, save_mean, save_invstd, partials, M, N, lddz, ldy,
                           ldz, relu, gscale, (int)rpb, chv, rt, reinterpret_cast<uint32_t*>(gmax));
    DML_LAUNCH_CHECK();
    return 0;
}

// folds `nblocks` partial rows down to <= 128 chunk heads when there are many; returns the row stride of the heads
static int fold_bwd_partials(float* partials, int& nblocks, int N, hipStream_t st) {
    if (nblocks <= 2048) return 1;
    const int R = (nblocks + 127) / 128;
    const int NC = (nblocks + R - 1) / R;
    hipLaunchKernelGGL(bn_bwd_fold_kernel, dim3((N + 63) / 64, NC), dim3(256), 0, st, partials, nblocks, N, R);
    nblocks = NC;
    return R;
}

extern "C" int dml_bn_bwd_finalize(float* partials, int nblocks, int64_t M, int N, const float* gamma,
                                   const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                   float* coef, void* stream) {
    if (!partials || !save_mean || !save_invstd || !coef || nblocks <= 0 || N <= 0 || M < 0) return DML_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int stride = fold_bwd_partials(partials, nblocks, N, st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((N + FIN_CH - 1) / FIN_CH), dim3(256), 0, st, partials, nblocks, M, N,
                       gamma, save_mean, save_invstd, dgamma, dbeta, coef, (double*)nullptr, stride);
    DML_LAUNCH_CHECK();
    return 0;
}

// dml_bn_bwd_finalize + dml_h2_bound_bn_bwd(coef, save_invstd, N, count, g_amax, work) in ONE launch (h2_bound_tail)
extern "C" int dml_bn_bwd_finalize_bound(float* partials, int nblocks, int64_t M, int N, const float* gamma,
                                         const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                                         float* coef, int64_t count, const float* g_amax, float* work, uint32_t* state,
                                         void* stream) {
    if (!partials || !save_mean || !save_invstd || !coef || nblocks <= 0 || N <= 0 || M < 0) return DML_EINVAL;
    if (!g_amax || !work || !state || count <= 0) return DML_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int stride = fold_bwd_partials(partials, nblocks, N, st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((N + FIN_CH - 1) / FIN_CH), dim3(256), 0, st, partials, nblocks, M, N,
                       gamma, save_mean, save_invstd, dgamma, dbeta, coef, (double*)nullptr, stride, state, work, g_amax,
                       sqrtf((float)count) * 1.0001f);
    DML_LAUNCH_CHECK();
    return 0;
}

// ---- synchronised BatchNorm (statistics over all ranks; anomaly/lib/nn/modules/batchnorm.py:56-139 of the reference,
// SURVEY 8(e)/(f) rank 2).  The collectives themselves are the caller's (RCCL through torch.distributed):
//   forward : dml_bn_moments -> all_gather of the [N][2] doubles -> dml_bn_finalize_moments
//   backward: dml_bn_bwd_sums -> all_reduce(sum) of the [N][2] doubles -> dml_bn_bwd_coef
namespace {
__global__ __launch_bounds__(256) void bn_finalize_moments_kernel(
    const double* __restrict__ moments, int R, int64_t M_each, int N, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, float momentum, float eps, float* scale,
    float* shift, float* save_mean, float* save_invstd) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    Moments t = {0.0, 0.0, 0.0};
    for (int r = 0; r < R; ++r) merge(t, (double)M_each, moments[((int64_t)r * N + n) * 2], moments[((int64_t)r * N + n) * 2 + 1]);
    const double var_b = t.m2 / t.n;
    const float invstd = (float)(1.0 / sqrt(var_b + (double)eps));
    scale[n] = (gamma ? gamma[n] : 1.f) * invstd;
    shift[n] = beta ? beta[n] : 0.f;
    save_mean[n] = (float)t.mean;
    if (save_invstd) save_invstd[n] = invstd;
    if (running_mean) running_mean[n] = (1.f - momentum) * running_mean[n] + momentum * (float)t.mean;
    if (running_var) {
        const double var_u = t.n > 1.0 ? t.m2 / (t.n - 1.0) : var_b;      // unbiased over ALL ranks' samples (batchnorm.py:133-136)
        running_var[n] = (1.f - momentum) * running_var[n] + momentum * (float)var_u;
    }
}
__global__ __launch_bounds__(256) void bn_bwd_coef_kernel(const double* __restrict__ sums, int64_t M_total, int N,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ save_mean,
                                                          const float* __restrict__ save_invstd, float* __restrict__ coef) {
    const int n = blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const double dbeta_s = sums[2 * n], dgamma_s = sums[2 * n + 1];
    const double g = gamma ? (double)gamma[n] : 1.0, is = save_invstd[n];
    const double A = g * is;
    coef[n] = (float)A;
    coef[N + n] = (float)(-A * is * dgamma_s / (double)M_total);
    coef[2 * N + n] = (float)(-A * dbeta_s / (double)M_total);
    coef[3 * N + n] = save_mean[n];
}
}  // namespace

extern "C" int dml_bn_moments(float* partials, int64_t M, int N, int stat_rows, double* moments, void* stream) {
    if (!partials || !moments || M <= 0 || N <= 0 || stat_rows <= 0) return DML_EINVAL;
    hipLaunchKernelGGL((bn_finalize_kernel<4, 64>), dim3((N + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream),
                       partials, M, N, stat_rows, (const float*)nullptr, (const float*)nullptr, (float*)nullptr, (float*)nullptr, 0.f,
                       0.f, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, moments);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_finalize_moments(const double* moments, int ranks, int64_t M_each, int N, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var, float momentum,
                                       float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                                       void* stream) {
    if (!moments || !scale || !shift || !save_mean || ranks <= 0 || M_each <= 0 || N <= 0) return DML_EINVAL;
    hipLaunchKernelGGL(bn_finalize_moments_kernel, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       moments, ranks, M_each, N, gamma, beta, running_mean, running_var, momentum, eps, scale, shift,
                       save_mean, save_invstd);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_bwd_sums(float* partials, int nblocks, int N, double* sums, float* dgamma, float* dbeta,
                               void* stream) {
    if (!partials || !sums || nblocks <= 0 || N <= 0) return DML_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int stride = fold_bwd_partials(partials, nblocks, N, st);
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((N + FIN_CH - 1) / FIN_CH), dim3(256), 0, st, partials, nblocks,
                       (int64_t)1, N, (const float*)nullptr, (const float*)nullptr, (const float*)nullptr, dgamma, dbeta,
                       (float*)nullptr, sums, stride);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_bwd_coef(const double* sums, int64_t M_total, int N, const float* gamma, const float* save_mean,
                               const float* save_invstd, float* coef, void* stream) {
    if (!sums || !save_mean || !save_invstd || !coef || M_total <= 0 || N <= 0) return DML_EINVAL;
    hipLaunchKernelGGL(bn_bwd_coef_kernel, dim3((N + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), sums,
                       M_total, N, gamma, save_mean, save_invstd, coef);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bn_bwd_apply(const void* dz, const void* y, const void* z, const uint8_t* mask, const float* coef, void* dy,
                                void* dres, int64_t M, int N, int lddz, int ldy, int ldz, int lddy, int lddres,
                                int relu, float gscale, int dres_accum, int dtype, float* amax, void* planes,
                                int64_t plane_stride, int32_t ldp, const float* unscale, void* stream) {
    if (!dz || !y || !coef || (!dy && !planes) || M <= 0 || N <= 0) return DML_EINVAL;
    if (relu && !z && !mask) return DML_EINVAL;
    if (!vec_ok(dtype, N) || !vec_ok(dtype, lddz) || !vec_ok(dtype, ldy) || (dy && !vec_ok(dtype, lddy)) ||
        (relu && !mask && !vec_ok(dtype, ldz)) || (dres && !vec_ok(dtype, lddres)))
        return DML_EALIGN;
    if (M >= (1ll << 31)) return DML_EINVAL;
    if (planes) {           // fp32 only: dy also (or only: dy == NULL) as two fp16 planes scaled by 1 / unscale[0]
        if (dtype != DML_F32 || !unscale || plane_stride <= 0 || ldp < N) return DML_EINVAL;
        if ((ldp & 3) || (plane_stride & 3) || (reinterpret_cast<uintptr_t>(planes) & 7)) return DML_EALIGN;
    }
    const int V = dtype == DML_BF16 ? 8 : 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // ONE-WAVE workgroups.  In the f16x2 backward this kernel runs on the main stream while a weight-gradient kernel
    // of the side stream holds every CU with one persistent 7-wave workgroup of 256 registers per lane -- 3 SIMDs full, 256 registers
    // free on the fourth.  A 4-wave block of ~128 registers per lane does not fit beside it and waited for a weight-gradient
    // workgroup to retire (in the overlapped trace this kernel lasted 175 us per launch against 66 alone); one-wave blocks start at
    // once, two per CU: 80.35 -> 79.15 ms per step, alone unchanged (profiles/r05_ab_bn_bwd_one_wave_blocks.txt).  The bf16 step, whose
    // convolution workgroups are not persistent, gains 0.5 % from the same geometry (40.06 -> 39.87 ms).
    // DML_BN_BWD_THREADS=256: the old geometry (A/B)
    static const int bt_env = getenv("DML_BN_BWD_THREADS") ? atoi(getenv("DML_BN_BWD_THREADS")) : 64;
    const int bt = (bt_env == 64 || bt_env == 128 || bt_env == 256) ? bt_env : 256;
    if (dtype == DML_F32 && planes && !dy && !amax && (mask || !relu) && (N & 7) == 0 && (lddz & 3) == 0 && (ldy & 3) == 0 &&
        (ldp & 7) == 0 && (plane_stride & 7) == 0 && (!dres || (lddres & 3) == 0) &&
        ((reinterpret_cast<uintptr_t>(planes) | reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(y) |
          reinterpret_cast<uintptr_t>(dres)) & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 1) == 0) {
        // dy as planes only: eight channels per thread, 16-byte accesses throughout
        const ColGeom g8 = col_geom(M, N / 8, 2, stream_blocks(M, N, dtype), bt);
        hipLaunchKernelGGL((bn_bwd_apply_planes8_kernel<2>), dim3(g8.row_blocks, g8.col_chunks), dim3(bt), 0, st, (const float*)dz,
                           (const float*)y, mask, coef, (float*)dres, M, N, lddz, ldy, lddres, relu, gscale, dres_accum, g8.CB, g8.RB,
                           g8.rows_per_block, static_cast<_Float16*>(planes), plane_stride, (int)ldp, unscale);
        DML_LAUNCH_CHECK();
        return 0;
    }
    const ColGeom g = col_geom(M, N / V, 2, stream_blocks(M, N, dtype), bt);
    dim3 grid(g.row_blocks, g.col_chunks);
    if (dtype == DML_BF16)
        hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<bf16_t, 2>), grid, dim3(bt), 0, st, (const bf16_t*)dz,
                           (const bf16_t*)y, (const bf16_t*)z, mask, coef, (bf16_t*)dy, (bf16_t*)dres, M, N, lddz, ldy,
                           ldz, lddy, lddres, relu, gscale, dres_accum, g.CB, g.RB, g.rows_per_block,
                           reinterpret_cast<uint32_t*>(amax), (_Float16*)nullptr, (int64_t)0, 0, (const float*)nullptr);
    else
        hipLaunchKernelGGL((bn_bwd_apply_cols_kernel<float, 2>), grid, dim3(bt), 0, st, (const float*)dz,
                           (const float*)y, (const float*)z, mask, coef, (float*)dy, (float*)dres, M, N, lddz, ldy, ldz,
                           lddy, lddres, relu, gscale, dres_accum, g.CB, g.RB, g.rows_per_block, reinterpret_cast<uint32_t*>(amax),
                           static_cast<_Float16*>(planes), plane_stride, (int)ldp, unscale);
    DML_LAUNCH_CHECK();
    return 0;
}

// ---- scale of a BatchNorm output's fp16 planes from a BOUND on its magnitude, known before the tensor exists -----------
// Batch statistics bound the normalised value: sum_m (y_m - mean)^2 = count * var, so |y_m - mean| * invstd <= sqrt(count)
// for every element, whatever the data.  Forward:  |z| <= max_c (|gamma_c| sqrt(count) + |beta_c|) * mult + max |res|.
// Backward (dy = A g + Bc (y - mean) + C0, bn_bwd_finalize_kernel): |dy| <= max_c (|A_c| max|g| + |Bc_c| sqrt(count) / invstd_c
// + |C0_c|).  The bounds sit 2^4 .. 2^8 above the true maxima of this network's tensors; the planes then resolve an element x
// to 2^-22 |x| down to |x| ~ 2^-11 of the bound and to 2^-33 of the bound below -- invisible beside the 2^-22 of the large
// elements of the same dot product.  (Running-statistics BatchNorm has no such bound: those outputs go through dml_h2_split.)
namespace {
__global__ __launch_bounds__(256) void h2_bound_kernel(const float* __restrict__ p0, const float* __restrict__ p1,
                                                       const float* __restrict__ p2, const float* __restrict__ invstd, int N,
                                                       float root_count, float mult, const float* __restrict__ words,
                                                       float* __restrict__ work, int bwd) {
    __shared__ float sh[4];
    __shared__ uint32_t shw[4];
    uint32_t wm = 0;
    if (words != nullptr)
        for (int i = threadIdx.x; i < 1024; i += 256) wm = max(wm, __float_as_uint(words[i]));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) wm = max(wm, (uint32_t)__shfl_xor((int)wm, o, 64));
    if ((threadIdx.x & 63) == 0) shw[threadIdx.x >> 6] = wm;
    __syncthreads();
    const float wmax = __uint_as_float(max(max(shw[0], shw[1]), max(shw[2], shw[3])));      // max |res| (forward) / max |g| (backward)
    float b = 0.f;
    for (int n = threadIdx.x; n < N; n += 256) {
        float v;
        if (!bwd) v = fabsf(p0 ? p0[n] : 1.f) * root_count + fabsf(p1 ? p1[n] : 0.f);
        else v = fabsf(p0[n]) * wmax + fabsf(p1[n]) * root_count / invstd[n] + fabsf(p2[n]);
        b = fmaxf(b, v);                                       // (NaN coefficients: fmaxf drops them; the tensor carries them anyway)
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x != 0) return;
    b = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    b = bwd ? b : b * mult + wmax;
    b *= 1.0009765625f;                                        // rounding of the statistics and of this sum
    // scale = 2^(14 - floor(log2 b)): b * scale in [2^14, 2^15), as dml_h2_split
    const uint32_t m = __float_as_uint(b);
    float s = 1.0f;
    if (m != 0) {
        int se = 14 - ((int)(m >> 23) - 127);
        se = se > 127 ? 127 : (se < -126 ? -126 : se);
        s = __uint_as_float((uint32_t)(se + 127) << 23);
        if (m >= 0x7f800000u) s = 1.0f;                        // the bound overflowed: the tensor is not finite either
    }
    work[1024] = 1.0f / s;
}
}  // namespace

// every residual-free BatchNorm of a plan in one launch (their bounds depend on gamma / beta / count only): one workgroup per entry
namespace {
__global__ __launch_bounds__(256) void h2_bound_table_kernel(const DmlH2BoundDesc* __restrict__ table) {
    const DmlH2BoundDesc d = table[blockIdx.x];
    __shared__ float sh[4];
    float b = 0.f;
    for (int n = threadIdx.x; n < d.N; n += 256)
        b = fmaxf(b, fabsf(d.gamma ? d.gamma[n] : 1.f) * d.root_count + fabsf(d.beta ? d.beta[n] : 0.f));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x != 0) return;
    b = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3])) * d.mult * 1.0009765625f;
    const uint32_t m = __float_as_uint(b);
    float s = 1.0f;
    if (m != 0) {
        int se = 14 - ((int)(m >> 23) - 127);
        se = se > 127 ? 127 : (se < -126 ? -126 : se);
        s = __uint_as_float((uint32_t)(se + 127) << 23);
        if (m >= 0x7f800000u) s = 1.0f;
    }
    d.work[1024] = 1.0f / s;
}
}  // namespace

// ONE scale for a tensor several BatchNorms write channel slices of (the decoder's concat buffer: low-level projection + upsampled
// ASPP projection): the largest of their bounds.  (A bilinear resize of a bounded tensor is bounded by the same value: its weights are
// non-negative and sum to one.)
namespace {
__global__ __launch_bounds__(256) void h2_bound_multi_kernel(const DmlH2BoundDesc* __restrict__ table, int count, float* __restrict__ work) {
    __shared__ float sh[4];
    float b = 0.f;
    for (int e = 0; e < count; ++e) {
        const DmlH2BoundDesc d = table[e];
        float be = 0.f;
        for (int n = threadIdx.x; n < d.N; n += 256)
            be = fmaxf(be, fabsf(d.gamma ? d.gamma[n] : 1.f) * d.root_count + fabsf(d.beta ? d.beta[n] : 0.f));
        b = fmaxf(b, be * d.mult);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) b = fmaxf(b, __shfl_xor(b, o, 64));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = b;
    __syncthreads();
    if (threadIdx.x != 0) return;
    b = fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
    work[1024] = 1.0f / h2_scale_of_bound(b);
}
}  // namespace

extern "C" int dml_h2_bound_bn_multi(const DmlH2BoundDesc* table_device, int count, float* work, void* stream) {
    if (!table_device || count <= 0 || !work) return DML_EINVAL;
    hipLaunchKernelGGL(h2_bound_multi_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), table_device, count, work);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_h2_bound_bn_table(const DmlH2BoundDesc* table_device, int count, void* stream) {
    if (count == 0) return 0;
    if (!table_device || count < 0) return DML_EINVAL;
    hipLaunchKernelGGL(h2_bound_table_kernel, dim3(count), dim3(256), 0, static_cast<hipStream_t>(stream), table_device);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_h2_bound_bn(const float* gamma, const float* beta, int N, int64_t count, float mult, const float* res_amax,
                               float* work, void* stream) {
    if (!work || N <= 0 || count <= 0 || !(mult > 0.f)) return DML_EINVAL;
    hipLaunchKernelGGL(h2_bound_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), gamma, beta, (const float*)nullptr,
                       (const float*)nullptr, N, sqrtf((float)count) * 1.0001f, mult, res_amax, work, 0);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_h2_bound_bn_bwd(const float* coef, const float* save_invstd, int N, int64_t count, const float* g_amax,
                                   float* work, void* stream) {
    if (!coef || !save_invstd || !g_amax || !work || N <= 0 || count <= 0) return DML_EINVAL;
    hipLaunchKernelGGL(h2_bound_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), coef, coef + N, coef + 2 * N,
                       save_invstd, N, sqrtf((float)count) * 1.0001f, 1.0f, g_amax, work, 1);
    DML_LAUNCH_CHECK();
    return 0;
}
