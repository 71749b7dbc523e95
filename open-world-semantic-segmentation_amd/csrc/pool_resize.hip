// NHWC pooling and bilinear resize kernels for gfx950 (HBM-bound; 16-byte vectors per lane).
//
// Replaces nn.MaxPool2d(3,2,1) (network/backbone/resnet.py:143), nn.AdaptiveAvgPool2d(1)
// (network/utils.py:320) and F.interpolate(mode='bilinear', align_corners=False)
// (network/utils.py:30,329; the 1x1 -> HxW case of :329 is a broadcast).
#include "common.h"

namespace {

// ------------------------------------------------------------------------------- max pool 3x3 s2 p1
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          uint8_t* __restrict__ argmax, int B, int H, int W,
                                                          int C, int Ho, int Wo) {
    constexpr int V = Vec16<T>::N;
    const int CV = C / V;
    const int64_t total = (int64_t)B * Ho * Wo * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t p = i / CV;
        const int xo = (int)(p % Wo); p /= Wo;
        const int yo = (int)(p % Ho);
        const int b = (int)(p / Ho);
        float best[V];
        int idx[V];
#pragma unroll
        for (int q = 0; q < V; ++q) { best[q] = -INFINITY; idx[q] = -1; }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yi = yo * 2 - 1 + r;
            if ((unsigned)yi >= (unsigned)H) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int xi = xo * 2 - 1 + s;
                if ((unsigned)xi >= (unsigned)W) continue;
                float v[V];
                Vec16<T>::load(x + (((int64_t)b * H + yi) * W + xi) * C + cv * V, v);
#pragma unroll
                for (int q = 0; q < V; ++q) {
                    if (idx[q] < 0 || v[q] > best[q] || v[q] != v[q]) { best[q] = v[q]; idx[q] = r * 3 + s; }
                }
            }
        }
        const int64_t o = (((int64_t)b * Ho + yo) * Wo + xo) * C + cv * V;
        Vec16<T>::store(y + o, best);
        if (argmax != nullptr) {
#pragma unroll
            for (int q = 0; q < V; ++q) argmax[o + q] = (uint8_t)idx[q];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy,
                                                          const uint8_t* __restrict__ argmax, T* __restrict__ dx,
                                                          int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int V = Vec16<T>::N;
    const int CV = C / V;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t p = i / CV;
        const int xi = (int)(p % W); p /= W;
        const int yi = (int)(p % H);
        const int b = (int)(p / H);
        float acc[V];
#pragma unroll
        for (int q = 0; q < V; ++q) acc[q] = 0.f;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int ty = yi + 1 - r;
            if (ty < 0 || (ty & 1)) continue;
            const int yo = ty >> 1;
            if (yo >= Ho) continue;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int tx = xi + 1 - s;
                if (tx < 0 || (tx & 1)) continue;
                const int xo = tx >> 1;
                if (xo >= Wo) continue;
                const int64_t o = (((int64_t)b * Ho + yo) * Wo + xo) * C + cv * V;
                float g[V];
                Vec16<T>::load(dy + o, g);
                uint8_t am[V];
                if (V == 8) {
                    const uint2 t = *reinterpret_cast<const uint2*>(argmax + o);
                    const uint32_t w[2] = {t.x, t.y};
#pragma unroll
                    for (int q = 0; q < 8; ++q) am[q] = (uint8_t)(w[q >> 2] >> ((q & 3) * 8));
                } else {
                    const uint32_t t = *reinterpret_cast<const uint32_t*>(argmax + o);
#pragma unroll
                    for (int q = 0; q < 4; ++q) am[q] = (uint8_t)(t >> (q * 8));
                }
#pragma unroll
                for (int q = 0; q < V; ++q) acc[q] += (am[q] == r * 3 + s) ? g[q] : 0.f;
            }
        }
        Vec16<T>::store(dx + (((int64_t)b * H + yi) * W + xi) * C + cv * V, acc);
    }
}

// ------------------------------------------------------------------- reductions over HW / broadcasts
// out[b][c] = scale * sum_{p < HW} x[b][p][c];  block = 8 vector columns (128 B of a row) x 32 row lanes, four rows in
// flight per lane: with 32 x 8 the launch had B * C / 256 workgroups (16 for the 256-channel gradient of the pooled
// branch) and one load in flight per lane -- 145 us for 19 MB.
template <typename T, typename TO = T>
__global__ __launch_bounds__(256) void reduce_hw_kernel(const T* __restrict__ x, TO* __restrict__ out, int HW,
                                                        int C, int ldx, float scale) {
    constexpr int V = Vec16<T>::N;
    constexpr int COLS = 8, LANES = 32, U = 4;
    __shared__ float sh[LANES][COLS * V + 1];
    const int col = threadIdx.x & (COLS - 1), rl = threadIdx.x / COLS;
    const int b = blockIdx.x;
    const int c = (blockIdx.y * COLS + col) * V;
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    if (c < C) {
        const T* base = x + (int64_t)b * HW * ldx + c;
        int p = rl;
        for (; p + (U - 1) * LANES < HW; p += U * LANES) {
            float v[U][V];
#pragma unroll
            for (int u = 0; u < U; ++u) Vec16<T>::load(base + (int64_t)(p + u * LANES) * ldx, v[u]);
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int q = 0; q < V; ++q) acc[q] += v[u][q];
        }
        for (; p < HW; p += LANES) {
            float v[V];
            Vec16<T>::load(base + (int64_t)p * ldx, v);
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += v[q];
        }
    }
#pragma unroll
    for (int q = 0; q < V; ++q) sh[rl][col * V + q] = acc[q];
    __syncthreads();
    // fold the 32 row lanes: thread t < COLS * V owns one channel
    if (threadIdx.x < COLS * V) {
        const int cc = blockIdx.y * COLS * V + threadIdx.x;
        if (cc < C) {
            float s = 0.f;
#pragma unroll
            for (int r = 0; r < LANES; ++r) s += sh[r][threadIdx.x];
            Elem<TO>::st(out + (int64_t)b * C + cc, s * scale);
        }
    }
}

// z[b][p][c] (op)= alpha * v[b][c]
template <typename T, bool ACCUM>
__global__ __launch_bounds__(256) void broadcast_hw_kernel(const T* __restrict__ v, T* __restrict__ z, int B,
                                                           int HW, int C, int ldz, float alpha) {
    constexpr int V = Vec16<T>::N;
    const int CV = C / V;
    const int64_t total = (int64_t)B * HW * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        const int64_t p = i / CV;
        const int b = (int)(p / HW);
        float s[V];
        Vec16<T>::load(v + (int64_t)b * C + cv * V, s);
#pragma unroll
        for (int q = 0; q < V; ++q) s[q] *= alpha;
        T* zp = z + p * ldz + cv * V;
        if (ACCUM) {
            float o[V];
            Vec16<T>::load(zp, o);
#pragma unroll
            for (int q = 0; q < V; ++q) s[q] += o[q];
        }
        Vec16<T>::store(zp, s);
    }
}

// ------------------------------------------------------------------------------- bilinear resize
// PyTorch area_pixel_compute_source_index, align_corners = False:
//   src = max(scale * (dst + 0.5) - 0.5, 0), i0 = (int)src, i1 = i0 + (i0 < in - 1), lambda = src - i0
struct Lerp {
    int i0, i1;
    float l0, l1;
};
__device__ __forceinline__ Lerp src_index(int dst, float scale, int in) {
    float s = scale * ((float)dst + 0.5f) - 0.5f;
    s = s < 0.f ? 0.f : s;
    Lerp r;
    r.i0 = min((int)s, in - 1);
    r.i1 = r.i0 + (r.i0 < in - 1 ? 1 : 0);
    r.l1 = s - (float)r.i0;
    r.l0 = 1.f - r.l1;
    return r;
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bilinear_fwd_kernel(const TI* __restrict__ x, TO* __restrict__ y, int B,
                                                           int h, int w, int H, int W, int C, int ldx, int ldy,
                                                           float sy, float sx) {
    constexpr int V = 4;   // 4 channels per thread (16 B fp32 / 8 B bf16)
    const int CV = C / V;
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t p = i / CV;
        const int X = (int)(p % W); p /= W;
        const int Y = (int)(p % H);
        const int b = (int)(p / H);
        const Lerp ly = src_index(Y, sy, h), lx = src_index(X, sx, w);
        const TI* base = x + (int64_t)b * h * w * ldx + cv * V;
        float o[V];
#pragma unroll
        for (int q = 0; q < V; ++q) {
            const float v00 = Elem<TI>::ld(base + ((int64_t)ly.i0 * w + lx.i0) * ldx + q);
            const float v01 = Elem<TI>::ld(base + ((int64_t)ly.i0 * w + lx.i1) * ldx + q);
            const float v10 = Elem<TI>::ld(base + ((int64_t)ly.i1 * w + lx.i0) * ldx + q);
            const float v11 = Elem<TI>::ld(base + ((int64_t)ly.i1 * w + lx.i1) * ldx + q);
            o[q] = ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
        }
        TO* yp = y + (((int64_t)b * H + Y) * W + X) * ldy + cv * V;
#pragma unroll
        for (int q = 0; q < V; ++q) Elem<TO>::st(yp + q, o[q]);
    }
}

// The same resize with the result written as the two fp16 planes of the power-of-two-scaled value (dml_h2_split's arithmetic, as the
// BatchNorm apply kernels write them): the decoder's concat buffer exists as planes only in an f16x2 training plan
__global__ __launch_bounds__(256) void bilinear_fwd_planes_kernel(const float* __restrict__ x, _Float16* __restrict__ planes,
                                                                  int64_t plane_stride, int ldp, const float* __restrict__ unscale, int B,
                                                                  int h, int w, int H, int W, int C, int ldx, float sy, float sx) {
    typedef _Float16 h4 __attribute__((ext_vector_type(4)));
    const int CV = C / 4;
    const float s = 1.0f / unscale[0];                 // exact: a power of two
    const int64_t total = (int64_t)B * H * W * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t p = i / CV;
        const int X = (int)(p % W); p /= W;
        const int Y = (int)(p % H);
        const int b = (int)(p / H);
        const Lerp ly = src_index(Y, sy, h), lx = src_index(X, sx, w);
        const float* base = x + (int64_t)b * h * w * ldx + cv * 4;
        const float4 v00 = *reinterpret_cast<const float4*>(base + ((int64_t)ly.i0 * w + lx.i0) * ldx);
        const float4 v01 = *reinterpret_cast<const float4*>(base + ((int64_t)ly.i0 * w + lx.i1) * ldx);
        const float4 v10 = *reinterpret_cast<const float4*>(base + ((int64_t)ly.i1 * w + lx.i0) * ldx);
        const float4 v11 = *reinterpret_cast<const float4*>(base + ((int64_t)ly.i1 * w + lx.i1) * ldx);
        const float a00[4] = {v00.x, v00.y, v00.z, v00.w}, a01[4] = {v01.x, v01.y, v01.z, v01.w};
        const float a10[4] = {v10.x, v10.y, v10.z, v10.w}, a11[4] = {v11.x, v11.y, v11.z, v11.w};
        h4 hi, lo;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // (the expression of bilinear_fwd_kernel: the same value bit for bit)
            const float o = ly.l0 * (lx.l0 * a00[q] + lx.l1 * a01[q]) + ly.l1 * (lx.l0 * a10[q] + lx.l1 * a11[q]);
            const float xs = o * s;
            const _Float16 hh = (_Float16)xs;
            hi[q] = hh;
            lo[q] = (_Float16)(xs - (float)hh);
        }
        _Float16* yp = planes + (((int64_t)b * H + Y) * W + X) * ldp + cv * 4;
        *reinterpret_cast<h4*>(yp) = hi;
        *reinterpret_cast<h4*>(yp + plane_stride) = lo;
    }
}

// gather form of the transpose: every source pixel sums the destination pixels that read it
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void bilinear_bwd_kernel(const TI* __restrict__ dy, TO* __restrict__ dx, int B,
                                                           int h, int w, int H, int W, int C, int lddy, int lddx,
                                                           float sy, float sx) {
    constexpr int V = 4;
    const int CV = C / V;
    const int64_t total = (int64_t)B * h * w * CV;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int cv = (int)(i % CV);
        int64_t p = i / CV;
        const int xs = (int)(p % w); p /= w;
        const int ys = (int)(p % h);
        const int b = (int)(p / h);
        // destination rows whose source coordinate can fall in (ys-1, ys+1)
        const int Y0 = max(0, (int)floorf(((float)ys - 0.5f) / sy - 0.5f) - 1);
        const int Y1 = min(H - 1, (int)ceilf(((float)ys + 1.5f) / sy - 0.5f) + 1);
        const int X0 = max(0, (int)floorf(((float)xs - 0.5f) / sx - 0.5f) - 1);
        const int X1 = min(W - 1, (int)ceilf(((float)xs + 1.5f) / sx - 0.5f) + 1);
        float acc[V] = {0.f, 0.f, 0.f, 0.f};
        for (int Y = Y0; Y <= Y1; ++Y) {
            const Lerp ly = src_index(Y, sy, h);
            const float wy = (ly.i0 == ys ? ly.l0 : 0.f) + (ly.i1 == ys ? ly.l1 : 0.f);
            if (wy == 0.f) continue;
            for (int X = X0; X <= X1; ++X) {
                const Lerp lx = src_index(X, sx, w);
                const float wx = (lx.i0 == xs ? lx.l0 : 0.f) + (lx.i1 == xs ? lx.l1 : 0.f);
                if (wx == 0.f) continue;
                const TI* g = dy + (((int64_t)b * H + Y) * W + X) * lddy + cv * V;
#pragma unroll
                for (int q = 0; q < V; ++q) acc[q] += wy * wx * Elem<TI>::ld(g + q);
            }
        }
        TO* o = dx + (((int64_t)b * h + ys) * w + xs) * lddx + cv * V;
#pragma unroll
        for (int q = 0; q < V; ++q) Elem<TO>::st(o + q, acc[q]);
    }
}

// ---- bf16 -> bf16 fast paths, 16-byte vectors (8 channels) per thread instead of scalar 2-byte accesses
__global__ __launch_bounds__(256) void bilinear_fwd_bf16v_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int B,
                                                                 int h, int w, int H, int W, int C, int ldx, int ldy,
                                                                 float sy, float sx) {
    const int CV = C / 8;
    const int64_t total = (int64_t)B * H * W * CV;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int X = (int)(p % W); p /= W;
    const int Y = (int)(p % H);
    const int b = (int)(p / H);
    const Lerp ly = src_index(Y, sy, h), lx = src_index(X, sx, w);
    const bf16_t* base = x + (int64_t)b * h * w * ldx + cv * 8;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    Vec16<bf16_t>::load(base + ((int64_t)ly.i0 * w + lx.i0) * ldx, v00);
    Vec16<bf16_t>::load(base + ((int64_t)ly.i0 * w + lx.i1) * ldx, v01);
    Vec16<bf16_t>::load(base + ((int64_t)ly.i1 * w + lx.i0) * ldx, v10);
    Vec16<bf16_t>::load(base + ((int64_t)ly.i1 * w + lx.i1) * ldx, v11);
#pragma unroll
    for (int q = 0; q < 8; ++q) o[q] = ly.l0 * (lx.l0 * v00[q] + lx.l1 * v01[q]) + ly.l1 * (lx.l0 * v10[q] + lx.l1 * v11[q]);
    Vec16<bf16_t>::store(y + (((int64_t)b * H + Y) * W + X) * ldy + cv * 8, o);
}

__global__ __launch_bounds__(256) void bilinear_bwd_bf16v_kernel(const bf16_t* __restrict__ dy, bf16_t* __restrict__ dx, int B,
                                                                 int h, int w, int H, int W, int C, int lddy, int lddx,
                                                                 float sy, float sx) {
    const int CV = C / 8;
    const int64_t total = (int64_t)B * h * w * CV;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int xs = (int)(p % w); p /= w;
    const int ys = (int)(p % h);
    const int b = (int)(p / h);
    const int Y0 = max(0, (int)floorf(((float)ys - 0.5f) / sy - 0.5f) - 1);
    const int Y1 = min(H - 1, (int)ceilf(((float)ys + 1.5f) / sy - 0.5f) + 1);
    const int X0 = max(0, (int)floorf(((float)xs - 0.5f) / sx - 0.5f) - 1);
    const int X1 = min(W - 1, (int)ceilf(((float)xs + 1.5f) / sx - 0.5f) + 1);
    float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int Y = Y0; Y <= Y1; ++Y) {
        const Lerp ly = src_index(Y, sy, h);
        const float wy = (ly.i0 == ys ? ly.l0 : 0.f) + (ly.i1 == ys ? ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int X = X0; X <= X1; ++X) {
            const Lerp lx = src_index(X, sx, w);
            const float wx = (lx.i0 == xs ? lx.l0 : 0.f) + (lx.i1 == xs ? lx.l1 : 0.f);
            if (wx == 0.f) continue;
            float g[8];
            Vec16<bf16_t>::load(dy + (((int64_t)b * H + Y) * W + X) * lddy + cv * 8, g);
            const float ww = wy * wx;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] += ww * g[q];
        }
    }
    Vec16<bf16_t>::store(dx + (((int64_t)b * h + ys) * w + xs) * lddx + cv * 8, acc);
}

// max-pool backward, one thread per 2 x 2 block of the input: the four positions share their four candidate windows
// (pooled outputs (a, b) .. (a + 1, b + 1)), so dy / argmax are fetched once per block instead of once per position
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_blk_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ argmax,
                                                              T* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo) {
    constexpr int V = Vec16<T>::N;
    const int CV = C / V, Hb = (H + 1) / 2, Wb = (W + 1) / 2;
    const int64_t total = (int64_t)B * Hb * Wb * CV;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int cv = (int)(i % CV);
    int64_t p = i / CV;
    const int bx = (int)(p % Wb); p /= Wb;
    const int by = (int)(p % Hb);
    const int b = (int)(p / Hb);
    float acc[2][2][V];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int q = 0; q < V; ++q) acc[u][t][q] = 0.f;
    // window (yo, xo) covers input rows 2 yo - 1 .. 2 yo + 1: of this block's rows 2 by, 2 by + 1 the window yo = by takes
    // both (r = 1, 2) and yo = by + 1 takes row 2 by + 1 (r = 0); the same along x
#pragma unroll
    for (int dyo = 0; dyo < 2; ++dyo) {
        const int yo = by + dyo;
        if (yo >= Ho) continue;
#pragma unroll
        for (int dxo = 0; dxo < 2; ++dxo) {
            const int xo = bx + dxo;
            if (xo >= Wo) continue;
            const int64_t o = (((int64_t)b * Ho + yo) * Wo + xo) * C + cv * V;
            float g[V];
            Vec16<T>::load(dy + o, g);
            uint8_t am[V];
            if (V == 8) {
                const uint2 t = *reinterpret_cast<const uint2*>(argmax + o);
                const uint32_t wv[2] = {t.x, t.y};
#pragma unroll
                for (int q = 0; q < 8; ++q) am[q] = (uint8_t)(wv[q >> 2] >> ((q & 3) * 8));
            } else {
                const uint32_t t = *reinterpret_cast<const uint32_t*>(argmax + o);
#pragma unroll
                for (int q = 0; q < 4; ++q) am[q] = (uint8_t)(t >> (q * 8));
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int r = 2 * by + u - (2 * yo - 1);           // tap row of input row 2 by + u in this window
                if (r < 0 || r > 2) continue;
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int sc = 2 * bx + t - (2 * xo - 1);
                    if (sc < 0 || sc > 2) continue;
#pragma unroll
                    for (int q = 0; q < V; ++q) acc[u][t][q] += (am[q] == r * 3 + sc) ? g[q] : 0.f;
                }
            }
        }
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int yi = 2 * by + u;
        if (yi >= H) continue;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int xi = 2 * bx + t;
            if (xi >= W) continue;
            Vec16<T>::store(dx + (((int64_t)b * H + yi) * W + xi) * C + cv * V, acc[u][t]);
        }
    }
}

inline bool vec_ok(int dtype, int a) { return a % (dtype == DML_BF16 ? 8 : 4) == 0; }

}  // namespace

extern "C" int dml_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int B, int H, int W, int C,
                                    int dtype, void* stream) {
    if (!x || !y || B <= 0 || H <= 0 || W <= 0) return DML_EINVAL;
    if (!vec_ok(dtype, C)) return DML_EALIGN;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int V = dtype == DML_BF16 ? 8 : 4;
    // one item per thread: short-lived workgroups sweep the tensor front to back (135 -> 111 us at 384 x 384 x 64 x 16)
    const int grid = grid_for((int64_t)B * Ho * Wo * (C / V), 256, 1 << 20);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == DML_BF16)
        hipLaunchKernelGGL(maxpool_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y,
                           argmax, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, (float*)y,
                           argmax, B, H, W, C, Ho, Wo);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int H, int W, int C,
                                    int dtype, void* stream) {
    if (!dy || !argmax || !dx) return DML_EINVAL;
    if (!vec_ok(dtype, C)) return DML_EALIGN;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int V = dtype == DML_BF16 ? 8 : 4;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t items = (int64_t)B * ((H + 1) / 2) * ((W + 1) / 2) * (C / V);
    if (items >= (1ll << 31) * 256) return DML_EINVAL;
    const dim3 grid((unsigned)((items + 255) / 256));
    if (dtype == DML_BF16)
        hipLaunchKernelGGL(maxpool_bwd_blk_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dy, argmax,
                           (bf16_t*)dx, B, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool_bwd_blk_kernel<float>, grid, dim3(256), 0, st, (const float*)dy, argmax,
                           (float*)dx, B, H, W, C, Ho, Wo);
    DML_LAUNCH_CHECK();
    return 0;
}

static int launch_reduce_hw(const void* x, void* out, int B, int HW, int C, int ldx, int dtype, float scale,
                            void* stream, bool out_f32 = false) {
    if (!x || !out || B <= 0 || HW <= 0) return DML_EINVAL;
    if (!vec_ok(dtype, C) || !vec_ok(dtype, ldx)) return DML_EALIGN;
    const int V = dtype == DML_BF16 ? 8 : 4;
    dim3 grid(B, (C / V + 7) / 8);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == DML_BF16 && out_f32)
        hipLaunchKernelGGL((reduce_hw_kernel<bf16_t, float>), grid, dim3(256), 0, st, (const bf16_t*)x, (float*)out, HW, C,
                           ldx, scale);
    else if (dtype == DML_BF16)
        hipLaunchKernelGGL(reduce_hw_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)out, HW, C,
                           ldx, scale);
    else
        hipLaunchKernelGGL(reduce_hw_kernel<float>, grid, dim3(256), 0, st, (const float*)x, (float*)out, HW, C,
                           ldx, scale);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_global_avgpool_fwd(const void* x, void* y, int B, int HW, int C, int ldx, int dtype,
                                      void* stream) {
    return launch_reduce_hw(x, y, B, HW, C, ldx, dtype, 1.0f / (float)HW, stream);
}
extern "C" int dml_reduce_hw(const void* dz, void* dv, int B, int HW, int C, int lddz, int dtype, void* stream) {
    return launch_reduce_hw(dz, dv, B, HW, C, lddz, dtype, 1.0f, stream);
}
extern "C" int dml_reduce_hw_f32(const void* x, float* out, int B, int HW, int C, int ldx, int dtype, float scale,
                                 void* stream) {
    return launch_reduce_hw(x, out, B, HW, C, ldx, dtype, scale, stream, true);
}

static int launch_broadcast(const void* v, void* z, int B, int HW, int C, int ldz, int dtype, float alpha,
                            bool accum, void* stream) {
    if (!v || !z || B <= 0 || HW <= 0) return DML_EINVAL;
    if (!vec_ok(dtype, C) || !vec_ok(dtype, ldz)) return DML_EALIGN;
    const int V = dtype == DML_BF16 ? 8 : 4;
    const int grid = grid_for((int64_t)B * HW * (C / V), 256);
    hipStream_t st = static_cast<hipStream_t>(stream);
#define GO(T, ACC)                                                                                              \
    hipLaunchKernelGGL((broadcast_hw_kernel<T, ACC>), dim3(grid), dim3(256), 0, st, (const T*)v, (T*)z, B, HW, C, \
                       ldz, alpha)
    if (dtype == DML_BF16) { if (accum) GO(bf16_t, true); else GO(bf16_t, false); }
    else { if (accum) GO(float, true); else GO(float, false); }
#undef GO
    DML_LAUNCH_CHECK();
    return 0;
}
extern "C" int dml_broadcast_hw(const void* v, void* z, int B, int HW, int C, int ldz, int dtype, void* stream) {
    return launch_broadcast(v, z, B, HW, C, ldz, dtype, 1.0f, false, stream);
}
extern "C" int dml_avgpool_bwd_add(const void* dv, void* dx, int B, int HW, int C, int lddx, int dtype,
                                   void* stream) {
    return launch_broadcast(dv, dx, B, HW, C, lddx, dtype, 1.0f / (float)HW, true, stream);
}
extern "C" int dml_avgpool_bwd_set(const void* dv, void* dx, int B, int HW, int C, int lddx, int dtype,
                                   void* stream) {
    return launch_broadcast(dv, dx, B, HW, C, lddx, dtype, 1.0f / (float)HW, false, stream);
}

template <bool BWD>
static int launch_bilinear(const void* src, void* dst, int B, int h, int w, int H, int W, int C, int ld_src,
                           int ld_dst, int dtype, int in_f32, int out_f32, void* stream) {
    if (!src || !dst || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DML_EINVAL;
    if (C % 4) return DML_EALIGN;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int64_t items = BWD ? (int64_t)B * h * w * (C / 4) : (int64_t)B * H * W * (C / 4);
    const int grid = grid_for(items, 256, 256 * 16);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const bool fi = in_f32 || dtype == DML_F32, fo = out_f32 || dtype == DML_F32;
    if (!fi && !fo && C % 8 == 0 && ld_src % 8 == 0 && ld_dst % 8 == 0 && (reinterpret_cast<uintptr_t>(src) & 15) == 0 &&
        (reinterpret_cast<uintptr_t>(dst) & 15) == 0) {
        const int64_t it8 = BWD ? (int64_t)B * h * w * (C / 8) : (int64_t)B * H * W * (C / 8);
        const dim3 g8((unsigned)((it8 + 255) / 256));
        if (BWD)
            hipLaunchKernelGGL(bilinear_bwd_bf16v_kernel, g8, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, h, w, H, W, C,
                               ld_src, ld_dst, sy, sx);
        else
            hipLaunchKernelGGL(bilinear_fwd_bf16v_kernel, g8, dim3(256), 0, st, (const bf16_t*)src, (bf16_t*)dst, B, h, w, H, W, C,
                               ld_src, ld_dst, sy, sx);
        DML_LAUNCH_CHECK();
        return 0;
    }
#define GO(TI, TO)                                                                                           \
    do {                                                                                                     \
        if (BWD)                                                                                             \
            hipLaunchKernelGGL((bilinear_bwd_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)src,  \
                               (TO*)dst, B, h, w, H, W, C, ld_src, ld_dst, sy, sx);                          \
        else                                                                                                 \
            hipLaunchKernelGGL((bilinear_fwd_kernel<TI, TO>), dim3(grid), dim3(256), 0, st, (const TI*)src,  \
                               (TO*)dst, B, h, w, H, W, C, ld_src, ld_dst, sy, sx);                          \
    } while (0)
    if (fi && fo) GO(float, float);
    else if (fi) GO(float, bf16_t);
    else if (fo) GO(bf16_t, float);
    else GO(bf16_t, bf16_t);
#undef GO
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_bilinear_fwd(const void* x, void* y, int B, int h, int w, int H, int W, int C, int ldx, int ldy,
                                int dtype, int in_f32, int out_f32, void* stream) {
    return launch_bilinear<false>(x, y, B, h, w, H, W, C, ldx, ldy, dtype, in_f32, out_f32, stream);
}
extern "C" int dml_bilinear_fwd_planes(const float* x, void* planes, int64_t plane_stride, int ldp, const float* unscale, int B, int h,
                                       int w, int H, int W, int C, int ldx, void* stream) {
    if (!x || !planes || !unscale || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0 || C <= 0 || plane_stride <= 0) return DML_EINVAL;
    if ((C & 3) || (ldx & 3) || (ldp & 3) || (plane_stride & 3) || ldp < C || ldx < C || (reinterpret_cast<uintptr_t>(x) & 15) ||
        (reinterpret_cast<uintptr_t>(planes) & 7))
        return DML_EALIGN;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    const int64_t total = (int64_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(bilinear_fwd_planes_kernel, dim3(grid_for(total, 256, 256 * 32)), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       static_cast<_Float16*>(planes), plane_stride, ldp, unscale, B, h, w, H, W, C, ldx, sy, sx);
    DML_LAUNCH_CHECK();
    return 0;
}
extern "C" int dml_bilinear_bwd(const void* dy, void* dx, int B, int h, int w, int H, int W, int C, int lddy,
                                int lddx, int dtype, int in_f32, int out_f32, void* stream) {
    return launch_bilinear<true>(dy, dx, B, h, w, H, W, C, lddy, lddx, dtype, in_f32, out_f32, stream);
}

// ------------------------------------------------------------------------------------------------
// Pyramid pooling pieces of the anomaly model's decoder (anomaly/models/models.py:586-687 of the reference,
// SURVEY 8(f) rank 2): nn.AdaptiveAvgPool2d(s) on NHWC features, the 13-prototype distance on the 1/8-resolution
// embedding, and F.interpolate(..., size=segSize, bilinear) of an NHWC fp32 map into the NCHW score tensor with the
// multi-scale average of eval_ood_traditional.py:198-210 (scores += scores_tmp / n) folded into the store.
// ------------------------------------------------------------------------------------------------
namespace {
__device__ __forceinline__ int bin_lo(int i, int n, int s) { return (int)(((int64_t)i * n) / s); }
__device__ __forceinline__ int bin_hi(int i, int n, int s) { return (int)((((int64_t)(i + 1)) * n + s - 1) / s); }

// stage 1: per (image, bin, row slice) fp32 sums; thread = one 16-byte channel vector, rows walked in a fixed order
template <typename T>
__global__ __launch_bounds__(256) void adaptive_pool_partial_kernel(const T* __restrict__ x, float* __restrict__ part, int H,
                                                                    int W, int C, int ldx, int S, int RS) {
    constexpr int V = Vec16<T>::N;
    const int cv = blockIdx.y * 256 + threadIdx.x;
    if (cv * V >= C) return;
    const int rs = blockIdx.x % RS, bin = blockIdx.x / RS;
    const int j = bin % S, i = (bin / S) % S, b = bin / (S * S);
    const int y0 = bin_lo(i, H, S), y1 = bin_hi(i, H, S), x0 = bin_lo(j, W, S), x1 = bin_hi(j, W, S);
    const int rows = (y1 - y0 + RS - 1) / RS;
    const int ya = y0 + rs * rows, yb = min(y1, ya + rows);
    float acc[V];
#pragma unroll
    for (int q = 0; q < V; ++q) acc[q] = 0.f;
    for (int y = ya; y < yb; ++y) {
        const T* row = x + (((int64_t)b * H + y) * W) * ldx + cv * V;
#pragma unroll 4
        for (int xx = x0; xx < x1; ++xx) {
            float v[V];
            Vec16<T>::load(row + (int64_t)xx * ldx, v);
#pragma unroll
            for (int q = 0; q < V; ++q) acc[q] += v[q];
        }
    }
    float* o = part + ((int64_t)blockIdx.x * C) + cv * V;
#pragma unroll
    for (int q = 0; q < V; ++q) o[q] = acc[q];
}
template <typename T>
__global__ __launch_bounds__(256) void adaptive_pool_finish_kernel(const float* __restrict__ part, T* __restrict__ y, int bins,
                                                                   int H, int W, int C, int S, int RS) {
    const int64_t total = (int64_t)bins * C;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int c = (int)(t % C), bin = (int)(t / C);
        const int j = bin % S, i = (bin / S) % S;
        const int cnt = (bin_hi(i, H, S) - bin_lo(i, H, S)) * (bin_hi(j, W, S) - bin_lo(j, W, S));
        float s = 0.f;
        for (int r = 0; r < RS; ++r) s += part[((int64_t)bin * RS + r) * C + c];
        Elem<T>::st(y + t, s / (float)cnt);
    }
}

// out[m][k] = -sum_c (e[m][c] - P[k][c])^2 on the low-resolution NHWC embedding (k < K; pad channels of out = 0)
__global__ __launch_bounds__(256) void proto_dist_nhwc_kernel(const float* __restrict__ e, const float* __restrict__ protos,
                                                              float* __restrict__ out, int64_t M, int K, int Kp, int lde,
                                                              int ldo) {
    __shared__ float P[33 * 32];
    for (int t = threadIdx.x; t < K * Kp; t += 256) P[t] = protos[t];
    __syncthreads();
    for (int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x; m < M; m += (int64_t)gridDim.x * 256) {
        float f[32];
        for (int c = 0; c < Kp; ++c) f[c] = e[m * lde + c];
        for (int k = 0; k < Kp; ++k) {
            float d = 0.f;
            if (k < K)
                for (int c = 0; c < Kp; ++c) {
                    const float t = f[c] - P[k * Kp + c];
                    d += t * t;
                }
            out[m * ldo + k] = k < K ? -d : 0.f;
        }
    }
}

// dst[b][c][Y][X] (+)= alpha * bilinear(src[b][.][.][c]); 4 consecutive X per thread, 16-byte stores when aligned
__global__ __launch_bounds__(256) void upsample_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B,
                                                               int h, int w, int ld, int C, int H, int W, float sy, float sx,
                                                               float alpha, int accumulate) {
    const int W4 = (W + 3) / 4;
    const int64_t total = (int64_t)B * C * H * W4;
    const bool vec = (W & 3) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15) == 0;
    for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
        const int xq = (int)(t % W4);
        int64_t p = t / W4;
        const int Y = (int)(p % H); p /= H;
        const int c = (int)(p % C);
        const int b = (int)(p / C);
        const Lerp ly = src_index(Y, sy, h);
        const float* r0 = src + (((int64_t)b * h + ly.i0) * w) * ld + c;
        const float* r1 = src + (((int64_t)b * h + ly.i1) * w) * ld + c;
        float o[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int X = min(xq * 4 + e, W - 1);
            const Lerp lx = src_index(X, sx, w);
            o[e] = alpha * (ly.l0 * (lx.l0 * r0[(int64_t)lx.i0 * ld] + lx.l1 * r0[(int64_t)lx.i1 * ld]) +
                            ly.l1 * (lx.l0 * r1[(int64_t)lx.i0 * ld] + lx.l1 * r1[(int64_t)lx.i1 * ld]));
        }
        float* d = dst + (((int64_t)b * C + c) * H + Y) * W + xq * 4;
        if (vec) {
            float4 v = make_float4(o[0], o[1], o[2], o[3]);
            if (accumulate) {
                const float4 old = *reinterpret_cast<const float4*>(d);
                v.x += old.x; v.y += old.y; v.z += old.z; v.w += old.w;
            }
            *reinterpret_cast<float4*>(d) = v;
        } else {
            for (int e = 0; e < 4 && xq * 4 + e < W; ++e) d[e] = accumulate ? d[e] + o[e] : o[e];
        }
    }
}
}  // namespace

extern "C" int64_t dml_adaptive_avgpool_ws_elems(int B, int H, int W, int C, int S) {
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || S <= 0) return 0;
    const int rows = (H + S - 1) / S + 1;
    const int RS = rows < 16 ? rows : 16;
    return (int64_t)B * S * S * RS * C;
}
extern "C" int dml_adaptive_avgpool_fwd(const void* x, void* y, float* ws, int B, int H, int W, int C, int ldx, int S,
                                        int dtype, void* stream) {
    if (!x || !y || !ws || B <= 0 || H <= 0 || W <= 0 || S <= 0 || S > H || S > W) return DML_EINVAL;
    if (!vec_ok(dtype, C) || !vec_ok(dtype, ldx)) return DML_EALIGN;
    const int rows = (H + S - 1) / S + 1;
    const int RS = rows < 16 ? rows : 16;
    const int V = dtype == DML_BF16 ? 8 : 4;
    const int bins = B * S * S;
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(bins * RS, (C / V + 255) / 256);
    const int fgrid = grid_for((int64_t)bins * C, 256);
    if (dtype == DML_BF16) {
        hipLaunchKernelGGL(adaptive_pool_partial_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)x, ws, H, W, C, ldx, S, RS);
        hipLaunchKernelGGL(adaptive_pool_finish_kernel<bf16_t>, dim3(fgrid), dim3(256), 0, st, ws, (bf16_t*)y, bins, H, W, C, S, RS);
    } else {
        hipLaunchKernelGGL(adaptive_pool_partial_kernel<float>, grid, dim3(256), 0, st, (const float*)x, ws, H, W, C, ldx, S, RS);
        hipLaunchKernelGGL(adaptive_pool_finish_kernel<float>, dim3(fgrid), dim3(256), 0, st, ws, (float*)y, bins, H, W, C, S, RS);
    }
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_proto_dist_nhwc(const float* emb, const float* protos, float* out, int64_t M, int K, int Kp, int lde,
                                   int ldo, void* stream) {
    if (!emb || !protos || !out || M <= 0 || K <= 0 || K > 33 || Kp < 1 || Kp > 32 || K > Kp + 1 || lde < Kp || ldo < Kp)
        return DML_EINVAL;
    if (K > Kp) return DML_EUNSUPPORTED;          // one output channel per prototype inside the padded width
    hipLaunchKernelGGL(proto_dist_nhwc_kernel, dim3(grid_for(M, 256)), dim3(256), 0, static_cast<hipStream_t>(stream), emb,
                       protos, out, M, K, Kp, lde, ldo);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_upsample_nhwc_to_nchw(const float* src, float* dst, int B, int h, int w, int ld, int C, int H, int W,
                                         float alpha, int accumulate, void* stream) {
    if (!src || !dst || B <= 0 || h <= 0 || w <= 0 || C <= 0 || C > ld || H <= 0 || W <= 0) return DML_EINVAL;
    const int64_t items = (int64_t)B * C * H * ((W + 3) / 4);
    hipLaunchKernelGGL(upsample_to_nchw_kernel, dim3(grid_for(items, 256, 256 * 32)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), src, dst, B, h, w, ld, C, H, W, (float)h / (float)H,
                       (float)w / (float)W, alpha, accumulate);
    DML_LAUNCH_CHECK();
    return 0;
}
