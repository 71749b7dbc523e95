// Pixel -> prototype squared-Euclidean-distance head, its backward, the DML loss and the open-world
// scores, for gfx950.  All HBM-bound (about 4 flop/byte): one lane owns 4 consecutive pixels so that
// every global access is a 16-byte vector; prototypes sit in LDS and are read as broadcasts.
//
// Replaces network/utils.py:92-118 (the reference materialises a B x HW x K x C tensor twice),
// utils/loss.py:34-42 / anomaly/models/models.py:42-78 (per-image per-class Python loop with host
// round trips), and the host numpy post-processing of test_embedding.py:339-350,428-445 and
// anomaly/eval_ood_traditional.py:301-305.
#include "common.h"
#include <cstdlib>

namespace {

// non-temporal stores of logits / features: standalone head 0.378 -> 0.358 ms, staged x4 kernel 0.237 -> 0.219 ms
// (tools/bench_dist_variants.py, 768 x 768 x 16)
constexpr bool DIST_NT_DEFAULT = true;
constexpr int MAXC = 32, MAXK = 33;   // embedding dim / prototype count supported by these kernels

// 16-byte store of an output that nobody re-reads from cache (logits / features: 1.2 GB per step, far beyond L2):
// NT = true marks it non-temporal
template <bool NT> __device__ __forceinline__ void store4(float* p, const float4 v) {
    if constexpr (NT) {
        typedef float f32x4_nt __attribute__((ext_vector_type(4)));
        f32x4_nt t = {v.x, v.y, v.z, v.w};
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4_nt*>(p));
    } else {
        *reinterpret_cast<float4*>(p) = v;
    }
}

__device__ __forceinline__ void load_protos(float* sp, const float* __restrict__ protos, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) sp[i] = protos[i];
    __syncthreads();
}

// logits for PX pixels held as f[c][px]; writes NCHW planes / argmax / dissum
template <int PX, int CM>
__device__ __forceinline__ void dist_and_store(const float (&f)[CM][PX], const float* sp, int C, int K,
                                               float* __restrict__ logits, uint8_t* __restrict__ argmax,
                                               float* __restrict__ dissum, int64_t plane, int64_t pix_in_img,
                                               int64_t img_base_logits, int64_t img_pix_base) {
    float best[PX], sum[PX];
    int bi[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) { best[p] = -INFINITY; bi[p] = 0; sum[p] = 0.f; }
    for (int k = 0; k < K; ++k) {
        float d[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) d[p] = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                const float m = sp[k * C + c];
#pragma unroll
                for (int p = 0; p < PX; ++p) {
                    const float t = f[c][p] - m;
                    d[p] += t * t;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const float lg = -d[p];
            d[p] = lg;
            sum[p] += d[p];
            if (lg > best[p]) { best[p] = lg; bi[p] = k; }
        }
        if (logits != nullptr) {
            float* o = logits + img_base_logits + (int64_t)k * plane + pix_in_img;
            if constexpr (PX == 4) *reinterpret_cast<float4*>(o) = make_float4(d[0], d[1], d[2], d[3]);
            else o[0] = d[0];
        }
    }
    if (argmax != nullptr) {
#pragma unroll
        for (int p = 0; p < PX; ++p) argmax[img_pix_base + pix_in_img + p] = (uint8_t)bi[p];
    }
    if (dissum != nullptr) {
#pragma unroll
        for (int p = 0; p < PX; ++p) dissum[img_pix_base + pix_in_img + p] = -sum[p];
    }
}

template <int PX, int CM>
__device__ __forceinline__ void store_feats(const float (&f)[CM][PX], int C, float* __restrict__ feats,
                                            int64_t pix_global) {
    if (feats == nullptr) return;
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        float* o = feats + (pix_global + p) * C;
        if ((C & 3) == 0) {
#pragma unroll
            for (int c = 0; c < CM; c += 4)
                if (c < C) *reinterpret_cast<float4*>(o + c) = make_float4(f[c][p], f[c + 1][p], f[c + 2][p], f[c + 3][p]);
        } else {
#pragma unroll
            for (int c = 0; c < CM; ++c)
                if (c < C) o[c] = f[c][p];
        }
    }
}

// ---- standalone head: x NCHW (what F.interpolate returns) -> logits NCHW, features NHWC
template <int PX, int CM>
__global__ __launch_bounds__(256) void proto_dist_fwd_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ protos,
                                                             float* __restrict__ logits, float* __restrict__ feats,
                                                             uint8_t* __restrict__ argmax,
                                                             float* __restrict__ dissum, int B, int C, int K,
                                                             int64_t HW) {
    __shared__ float sp[MAXK * MAXC];
    load_protos(sp, protos, K * C);
    const int64_t groups_per_img = HW / PX;
    const int64_t total = (int64_t)B * groups_per_img;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / groups_per_img;
        const int64_t pix = (i - b * groups_per_img) * PX;
        float f[CM][PX];
#pragma unroll
        for (int c = 0; c < CM; ++c) {
            if (c < C) {
                const float* src = x + (b * C + c) * HW + pix;
                if constexpr (PX == 4) {
                    const float4 t = *reinterpret_cast<const float4*>(src);
                    f[c][0] = t.x; f[c][1] = t.y; f[c][2] = t.z; f[c][3] = t.w;
                } else {
                    f[c][0] = src[0];
                }
            }
        }
        store_feats<PX, CM>(f, C, feats, b * HW + pix);
        dist_and_store<PX, CM>(f, sp, C, K, logits, argmax, dissum, HW, pix, b * K * HW, b * HW);
    }
}

// ---- per-wave NHWC <-> "4 pixels per lane" transposition through a 16 KB LDS image (C = 16 floats per pixel).
// Lane l owns pixel groups wave_first + l (4 pixels = 256 B = 16 chunks of 16 B).  Global accesses are issued so
// that lane l touches chunk j*64 + l of the wave's contiguous 16 KB: 1 KB of consecutive bytes per instruction.
template <bool NT = false>
__device__ __forceinline__ void wave_store_nhwc16(float4* st, const float (&f)[16][4], float* __restrict__ dst,
                                                  int lane, int64_t wave_first, int64_t total) {
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4)
            st[lane * 16 + ((p * 4 + c4) ^ (lane & 15))] =
                make_float4(f[c4 * 4][p], f[c4 * 4 + 1][p], f[c4 * 4 + 2][p], f[c4 * 4 + 3][p]);
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int q = j * 64 + lane, r = q >> 4;
        const float4 v = st[r * 16 + ((q & 15) ^ (r & 15))];
        if (wave_first + r < total) store4<NT>(dst + (wave_first + r) * 64 + (q & 15) * 4, v);
    }
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void wave_load_nhwc16(float4* st, float (&f)[16][4], const float* __restrict__ src,
                                                 int lane, int64_t wave_first, int64_t total) {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int q = j * 64 + lane, r = q >> 4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (wave_first + r < total) v = *reinterpret_cast<const float4*>(src + (wave_first + r) * 64 + (q & 15) * 4);
        st[r * 16 + ((q & 15) ^ (r & 15))] = v;
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 v = st[lane * 16 + ((p * 4 + c4) ^ (lane & 15))];
            f[c4 * 4][p] = v.x; f[c4 * 4 + 1][p] = v.y; f[c4 * 4 + 2][p] = v.z; f[c4 * 4 + 3][p] = v.w;
        }
    __builtin_amdgcn_wave_barrier();
}

// The same store in two halves (pixels 0-1, then 2-3) through an 8 KB-per-wave image: a 256-thread workgroup then needs
// 32 KB of LDS instead of 64 KB.  Store instruction j of a half covers chunks j*64 + lane of the wave's 8 KB: 128
// contiguous bytes per 8 lanes.  `bound`: first pixel group that must not be written.
template <bool NT>
__device__ __forceinline__ void wave_store_nhwc16_halves(float4* st, const float (&f)[16][4], float* __restrict__ dst,
                                                         int lane, int64_t wave_first, int64_t bound) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int pp = 0; pp < 2; ++pp)
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) {
                const int p = half * 2 + pp;
                st[lane * 8 + ((pp * 4 + c4) ^ (lane & 7))] =
                    make_float4(f[c4 * 4][p], f[c4 * 4 + 1][p], f[c4 * 4 + 2][p], f[c4 * 4 + 3][p]);
            }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int qq = j * 64 + lane, r = qq >> 3;
            const float4 v = st[r * 8 + ((qq & 7) ^ (r & 7))];
            if (wave_first + r < bound) store4<NT>(dst + (wave_first + r) * 64 + half * 32 + (qq & 7) * 4, v);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- fused upsample + head, K = C = 16: one pixel group per lane, features stored through the LDS transpose
template <bool NT>
__global__ __launch_bounds__(256) void upsample_dist_fwd_c16_kernel(const float* __restrict__ e,
                                                                    const float* __restrict__ protos,
                                                                    float* __restrict__ logits,
                                                                    float* __restrict__ feats, int B, int h, int w,
                                                                    int H, int W, float sy, float sx) {
    constexpr int C = 16, K = 16;
    __shared__ __attribute__((aligned(16))) float4 stage[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int WG = W / 4;
    const int64_t HW = (int64_t)H * W, groups_per_img = HW / 4, total = (int64_t)B * groups_per_img;
    const int64_t wave_first = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int64_t i = wave_first + lane;
    const bool active = i < total;
    const int64_t ic = active ? i : total - 1;
    const int b = (int)(ic / groups_per_img);
    const int64_t g = ic - (int64_t)b * groups_per_img;
    const int Y = (int)(g / WG), xg = (int)(g - (int64_t)Y * WG);
    float sY = sy * ((float)Y + 0.5f) - 0.5f;
    sY = sY < 0.f ? 0.f : sY;
    const int y0 = min((int)sY, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
    const float ly1 = sY - (float)y0, ly0 = 1.f - ly1;
    const float* r0 = e + ((int64_t)b * h + y0) * w * C;
    const float* r1 = e + ((int64_t)b * h + y1) * w * C;
    float f[C][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int X = xg * 4 + p;
        float sX = sx * ((float)X + 0.5f) - 0.5f;
        sX = sX < 0.f ? 0.f : sX;
        const int x0 = min((int)sX, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float lx1 = sX - (float)x0, lx0 = 1.f - lx1;
#pragma unroll
        for (int c = 0; c < C; c += 4) {
            const float4 a = *reinterpret_cast<const float4*>(r0 + x0 * C + c);
            const float4 bq = *reinterpret_cast<const float4*>(r0 + x1 * C + c);
            const float4 cq = *reinterpret_cast<const float4*>(r1 + x0 * C + c);
            const float4 d = *reinterpret_cast<const float4*>(r1 + x1 * C + c);
            f[c][p] = ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * cq.x + lx1 * d.x);
            f[c + 1][p] = ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * cq.y + lx1 * d.y);
            f[c + 2][p] = ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * cq.z + lx1 * d.z);
            f[c + 3][p] = ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * cq.w + lx1 * d.w);
        }
    }
    if (feats != nullptr) wave_store_nhwc16<NT>(stage[wave], f, feats, lane, wave_first, total);
    const int64_t pix = g * 4;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float m = protos[k * C + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float t = f[c][p] - m;
                d[p] += t * t;
            }
        }
        if (logits != nullptr && active)
            store4<NT>(logits + ((int64_t)b * K + k) * HW + pix, make_float4(-d[0], -d[1], -d[2], -d[3]));
    }
}

// ---- the same for an exact x4 upsample (H = 4h, W = 4w: the DeepLabV3+ head, network/utils.py:88) with the low-resolution
// operand staged in LDS.  A workgroup owns a band of up to four output rows that interpolate between the SAME two
// low-resolution rows (rows 4j+2 .. 4j+5 lie between rows j and j+1) and 256 output columns: 2 x 66 low-resolution
// pixels = 8.4 KB are fetched once, coalesced, instead of 64 gathered 16-byte loads per lane; each lane then blends the
// two rows for its three source columns and forms its 4 pixels with three weights per pixel.  The NHWC copy goes out
// through an 8 KB-per-wave transposition in two halves, so a workgroup needs 40 KB of LDS (4 per CU, the gathered
// kernel's 64 KB allow 2).
template <bool NT>
__global__ __launch_bounds__(256) void upsample4_dist_fwd_c16_kernel(const float* __restrict__ e,
                                                                     const float* __restrict__ protos,
                                                                     float* __restrict__ logits,
                                                                     float* __restrict__ feats, int B, int h, int w) {
    constexpr int C = 16, K = 16, COLS = 66;
    __shared__ __attribute__((aligned(16))) float4 low[2][COLS][4];       // [row][column][16-byte chunk, swizzled]
    __shared__ __attribute__((aligned(16))) float4 stage[4][512];         // per wave: 64 lanes x 2 pixels x 64 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int H = 4 * h, W = 4 * w;
    const int segs = (W + 255) / 256;
    int bid = blockIdx.x;
    const int seg = bid % segs;
    bid /= segs;
    const int jb = bid % (h + 1) - 1, b = bid / (h + 1);                  // band -1 .. h-1
    const int r0 = max(jb, 0), r1 = min(jb + 1, h - 1);
    const int Q0 = seg * 64;                                              // first low-resolution column of the segment
    for (int idx = tid; idx < 2 * COLS * 4; idx += 256) {
        const int row = idx / (COLS * 4), rem = idx - row * (COLS * 4);
        const int ci = rem >> 2, c4 = rem & 3;
        const int col = min(max(Q0 - 1 + ci, 0), w - 1);
        const float4 v = *reinterpret_cast<const float4*>(e + (((int64_t)b * h + (row ? r1 : r0)) * w + col) * C + c4 * 4);
        low[row][ci][c4 ^ (ci & 3)] = v;
    }
    __syncthreads();
    const int Y = 4 * jb + 2 + wave;
    if (Y < 0 || Y >= H) return;                                          // (whole waves: no barrier below)
    float sY = 0.25f * ((float)Y + 0.5f) - 0.5f;
    sY = sY < 0.f ? 0.f : sY;
    const int y0 = min((int)sY, h - 1);
    const float ly1 = sY - (float)y0, ly0 = 1.f - ly1;
    // y0 is r0 except on the last band's clamp, y1 = y0 + 1 clamped is r1 -- or both rows coincide
    const int ra = (y0 == r0) ? 0 : 1, rb = (min(y0 + 1, h - 1) == r1) ? 1 : 0;
    const int q = Q0 + lane;                                              // this lane's low-resolution column
    const bool active = q < w;
    // rows blended for the three source columns q-1, q, q+1 (LDS columns lane, lane+1, lane+2)
    float T[3][C];
#pragma unroll
    for (int s3 = 0; s3 < 3; ++s3) {
        const int ci = lane + s3;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            const float4 a = low[ra][ci][c4 ^ (ci & 3)], bq = low[rb][ci][c4 ^ (ci & 3)];
            T[s3][c4 * 4] = ly0 * a.x + ly1 * bq.x;
            T[s3][c4 * 4 + 1] = ly0 * a.y + ly1 * bq.y;
            T[s3][c4 * 4 + 2] = ly0 * a.z + ly1 * bq.z;
            T[s3][c4 * 4 + 3] = ly0 * a.w + ly1 * bq.w;
        }
    }
    float f[C][4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int X = 4 * q + p;
        float sX = 0.25f * ((float)X + 0.5f) - 0.5f;
        sX = sX < 0.f ? 0.f : sX;
        const int x0 = min((int)sX, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
        const float l1 = sX - (float)x0, l0 = 1.f - l1;
        const float wl = (x0 == q - 1 ? l0 : 0.f) + (x1 == q - 1 ? l1 : 0.f);
        const float wc = (x0 == q ? l0 : 0.f) + (x1 == q ? l1 : 0.f);
        const float wr = (x0 == q + 1 ? l0 : 0.f) + (x1 == q + 1 ? l1 : 0.f);
#pragma unroll
        for (int c = 0; c < C; ++c) f[c][p] = wl * T[0][c] + wc * T[1][c] + wr * T[2][c];
    }
    const int64_t HW = (int64_t)H * W;
    const int64_t row_first = (((int64_t)b * H + Y) * W) / 4;             // first pixel group of this output row
    const int64_t wave_first = row_first + Q0, row_end = row_first + w;
    if (feats != nullptr) wave_store_nhwc16_halves<NT>(stage[wave], f, feats, lane, wave_first, row_end);
    if (logits == nullptr || !active) return;
    const int64_t pix = (int64_t)Y * W + 4 * q;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float m = protos[k * C + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float t = f[c][p] - m;
                d[p] += t * t;
            }
        }
        store4<NT>(logits + ((int64_t)b * K + k) * HW + pix, make_float4(-d[0], -d[1], -d[2], -d[3]));
    }
}

// ---- backward of the head, K = C = 16: df = -2 sum_k g_k (f - m_k) (+ gfeats); NHWC tensors through the transpose
__global__ __launch_bounds__(256) void proto_dist_bwd_c16_kernel(const float* __restrict__ glogits,
                                                                 const float* __restrict__ gfeats,
                                                                 const float* __restrict__ feats,
                                                                 const float* __restrict__ protos,
                                                                 float* __restrict__ df, int B, int64_t HW) {
    constexpr int C = 16, K = 16;
    __shared__ __attribute__((aligned(16))) float4 stage[4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t groups_per_img = HW / 4, total = (int64_t)B * groups_per_img;
    const int64_t wave_first = ((int64_t)blockIdx.x * 4 + wave) * 64;
    const int64_t i = wave_first + lane;
    const bool active = i < total;
    const int64_t ic = active ? i : total - 1;
    const int64_t b = ic / groups_per_img;
    const int64_t pix = (ic - b * groups_per_img) * 4;
    float gs[4] = {0.f, 0.f, 0.f, 0.f};
    float o[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) o[c][p] = 0.f;
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        const float4 t = *reinterpret_cast<const float4*>(glogits + (b * K + k) * HW + pix);
        const float g[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int p = 0; p < 4; ++p) gs[p] += g[p];
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float m = protos[k * C + c];
#pragma unroll
            for (int p = 0; p < 4; ++p) o[c][p] += g[p] * m;              // sum_k g_k m_kc
        }
    }
    float f[C][4];
    wave_load_nhwc16(stage[wave], f, feats, lane, wave_first, total);
#pragma unroll
    for (int c = 0; c < C; ++c)
#pragma unroll
        for (int p = 0; p < 4; ++p) o[c][p] = -2.f * (gs[p] * f[c][p] - o[c][p]);
    if (gfeats != nullptr) {
        wave_load_nhwc16(stage[wave], f, gfeats, lane, wave_first, total);
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int p = 0; p < 4; ++p) o[c][p] += f[c][p];
    }
    wave_store_nhwc16(stage[wave], o, df, lane, wave_first, total);
}

// ---- standalone head, K = C = 16 fast path (the reference's configuration).  Per lane: 4 consecutive pixels,
// 16-byte loads of the 16 channel planes, 16-byte stores of the 16 logit planes; the NHWC copy of the features is
// transposed through a per-wave 16 KB LDS image (XOR-swizzled 16-byte chunks) so that every store instruction
// writes 1 KB of consecutive addresses instead of 64 scattered 16-byte pieces.  Prototypes are read with a
// wave-uniform index (scalar loads), no LDS, no block barrier.
template <bool NT>
__global__ __launch_bounds__(256) void proto_dist_fwd_c16_kernel(const float* __restrict__ x,
                                                                 const float* __restrict__ protos,
                                                                 float* __restrict__ logits,
                                                                 float* __restrict__ feats,
                                                                 uint8_t* __restrict__ argmax,
                                                                 float* __restrict__ dissum, int B, int64_t HW) {
    constexpr int C = 16, K = 16;
    __shared__ __attribute__((aligned(16))) float4 stage[4][512];       // per wave: 64 lanes x 2 px x 64 B (two halves)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t groups_per_img = HW / 4;
    const int64_t total = (int64_t)B * groups_per_img;
    const int64_t wave_first = ((int64_t)blockIdx.x * 4 + wave) * 64;    // first pixel group of this wave
    const int64_t i = wave_first + lane;
    const bool active = i < total;
    const int64_t ic = active ? i : total - 1;
    const int64_t b = ic / groups_per_img;
    const int64_t pix = (ic - b * groups_per_img) * 4;
    float f[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float4 t = *reinterpret_cast<const float4*>(x + (b * C + c) * HW + pix);
        f[c][0] = t.x; f[c][1] = t.y; f[c][2] = t.z; f[c][3] = t.w;
    }
    // NHWC features: pixel group g of the flattened (image, pixel / 4) order sits at feats + g * 64 floats
    if (feats != nullptr) wave_store_nhwc16_halves<NT>(stage[wave], f, feats, lane, wave_first, total);
    float best[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, sum[4] = {0.f, 0.f, 0.f, 0.f};
    int bi[4] = {0, 0, 0, 0};
#pragma unroll 4
    for (int k = 0; k < K; ++k) {
        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float m = protos[k * C + c];                            // wave-uniform -> scalar load
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const float t = f[c][p] - m;
                d[p] += t * t;
            }
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            d[p] = -d[p];
            sum[p] += d[p];
            if (d[p] > best[p]) { best[p] = d[p]; bi[p] = k; }
        }
        if (logits != nullptr && active)
            store4<NT>(logits + (b * K + k) * HW + pix, make_float4(d[0], d[1], d[2], d[3]));
    }
    if (active) {
        if (argmax != nullptr)
#pragma unroll
            for (int p = 0; p < 4; ++p) argmax[b * HW + pix + p] = (uint8_t)bi[p];
        if (dissum != nullptr)
            *reinterpret_cast<float4*>(dissum + b * HW + pix) = make_float4(-sum[0], -sum[1], -sum[2], -sum[3]);
    }
}

// ---- fused: bilinear upsample of the low-resolution embedding e[B,h,w,C] + head at [H,W]
template <int PX, int CM>
__global__ __launch_bounds__(256) void upsample_dist_fwd_kernel(const float* __restrict__ e,
                                                                const float* __restrict__ protos,
                                                                float* __restrict__ logits,
                                                                float* __restrict__ feats,
                                                                uint8_t* __restrict__ argmax,
                                                                float* __restrict__ dissum, int B, int h, int w,
                                                                int C, int K, int H, int W, float sy, float sx) {
    __shared__ float sp[MAXK * MAXC];
    load_protos(sp, protos, K * C);
    const int WG = W / PX;
    const int64_t HW = (int64_t)H * W;
    const int64_t total = (int64_t)B * H * WG;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int xg = (int)(i % WG);
        int64_t t = i / WG;
        const int Y = (int)(t % H);
        const int b = (int)(t / H);
        float sY = sy * ((float)Y + 0.5f) - 0.5f;
        sY = sY < 0.f ? 0.f : sY;
        const int y0 = min((int)sY, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float ly1 = sY - (float)y0, ly0 = 1.f - ly1;
        const float* r0 = e + ((int64_t)b * h + y0) * w * C;
        const float* r1 = e + ((int64_t)b * h + y1) * w * C;
        float f[CM][PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int X = xg * PX + p;
            float sX = sx * ((float)X + 0.5f) - 0.5f;
            sX = sX < 0.f ? 0.f : sX;
            const int x0 = min((int)sX, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
            const float lx1 = sX - (float)x0, lx0 = 1.f - lx1;
            if ((C & 3) == 0) {
#pragma unroll
                for (int c = 0; c < CM; c += 4) {
                    if (c < C) {
                        const float4 a = *reinterpret_cast<const float4*>(r0 + x0 * C + c);
                        const float4 bq = *reinterpret_cast<const float4*>(r0 + x1 * C + c);
                        const float4 cq = *reinterpret_cast<const float4*>(r1 + x0 * C + c);
                        const float4 d = *reinterpret_cast<const float4*>(r1 + x1 * C + c);
                        f[c][p] = ly0 * (lx0 * a.x + lx1 * bq.x) + ly1 * (lx0 * cq.x + lx1 * d.x);
                        f[c + 1][p] = ly0 * (lx0 * a.y + lx1 * bq.y) + ly1 * (lx0 * cq.y + lx1 * d.y);
                        f[c + 2][p] = ly0 * (lx0 * a.z + lx1 * bq.z) + ly1 * (lx0 * cq.z + lx1 * d.z);
                        f[c + 3][p] = ly0 * (lx0 * a.w + lx1 * bq.w) + ly1 * (lx0 * cq.w + lx1 * d.w);
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (c < C)
                        f[c][p] = ly0 * (lx0 * r0[x0 * C + c] + lx1 * r0[x1 * C + c]) +
                                  ly1 * (lx0 * r1[x0 * C + c] + lx1 * r1[x1 * C + c]);
                }
            }
        }
        const int64_t pix = (int64_t)Y * W + xg * PX;
        store_feats<PX, CM>(f, C, feats, (int64_t)b * HW + pix);
        dist_and_store<PX, CM>(f, sp, C, K, logits, argmax, dissum, HW, pix, (int64_t)b * K * HW, (int64_t)b * HW);
    }
}

// ---- backward of the head: df = -2 sum_k g_k (f - m_k) (+ gfeats)
template <int PX, int CM>
__global__ __launch_bounds__(256) void proto_dist_bwd_kernel(const float* __restrict__ glogits,
                                                             const float* __restrict__ gfeats,
                                                             const float* __restrict__ feats,
                                                             const float* __restrict__ protos,
                                                             float* __restrict__ df, int B, int C, int K,
                                                             int64_t HW) {
    __shared__ float sp[MAXK * MAXC];
    load_protos(sp, protos, K * C);
    const int64_t groups_per_img = HW / PX;
    const int64_t total = (int64_t)B * groups_per_img;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / groups_per_img;
        const int64_t pix = (i - b * groups_per_img) * PX;
        float gs[PX];                 // sum_k g_k
        float gm[CM][PX];           // sum_k g_k m_kc
#pragma unroll
        for (int p = 0; p < PX; ++p) gs[p] = 0.f;
#pragma unroll
        for (int c = 0; c < CM; ++c)
#pragma unroll
            for (int p = 0; p < PX; ++p) gm[c][p] = 0.f;
        for (int k = 0; k < K; ++k) {
            float g[PX];
            const float* src = glogits + (b * K + k) * HW + pix;
            if constexpr (PX == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src);
                g[0] = t.x; g[1] = t.y; g[2] = t.z; g[3] = t.w;
            } else {
                g[0] = src[0];
            }
#pragma unroll
            for (int p = 0; p < PX; ++p) gs[p] += g[p];
#pragma unroll
            for (int c = 0; c < CM; ++c) {
                if (c < C) {
                    const float m = sp[k * C + c];
#pragma unroll
                    for (int p = 0; p < PX; ++p) gm[c][p] += g[p] * m;
                }
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            const int64_t o = (b * HW + pix + p) * C;
            if ((C & 3) == 0) {
#pragma unroll
                for (int c = 0; c < CM; c += 4) {
                    if (c < C) {
                        const float4 fv = *reinterpret_cast<const float4*>(feats + o + c);
                        float4 v;
                        v.x = -2.f * (gs[p] * fv.x - gm[c][p]);
                        v.y = -2.f * (gs[p] * fv.y - gm[c + 1][p]);
                        v.z = -2.f * (gs[p] * fv.z - gm[c + 2][p]);
                        v.w = -2.f * (gs[p] * fv.w - gm[c + 3][p]);
                        if (gfeats != nullptr) {
                            const float4 gq = *reinterpret_cast<const float4*>(gfeats + o + c);
                            v.x += gq.x; v.y += gq.y; v.z += gq.z; v.w += gq.w;
                        }
                        *reinterpret_cast<float4*>(df + o + c) = v;
                    }
                }
            } else {
#pragma unroll
                for (int c = 0; c < CM; ++c) {
                    if (c < C) {
                        float v = -2.f * (gs[p] * feats[o + c] - gm[c][p]);
                        if (gfeats != nullptr) v += gfeats[o + c];
                        df[o + c] = v;
                    }
                }
            }
        }
    }
}

// ---- argmax / max-softmax-probability
__global__ __launch_bounds__(256) void argmax_msp_kernel(const float* __restrict__ logits,
                                                         int64_t* __restrict__ preds, float* __restrict__ msp,
                                                         int B, int K, int64_t HW) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, pix = i - b * HW;
        const float* src = logits + b * K * HW + pix;
        float best = src[0];
        int bi = 0;
        for (int k = 1; k < K; ++k) {
            const float v = src[(int64_t)k * HW];
            if (v > best) { best = v; bi = k; }
        }
        if (preds) preds[i] = bi;
        if (msp) {
            float den = 0.f;
            for (int k = 0; k < K; ++k) den += expf(src[(int64_t)k * HW] - best);
            msp[i] = 1.f - 1.f / den;
        }
    }
}

// ---- dissum score: clip(-sum_k logit_k), then per-image min-max normalisation
__device__ __forceinline__ void atomic_min_f(float* addr, float v) {
    if (v >= 0.f) atomicMin(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMax(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f(float* addr, float v) {
    if (v >= 0.f) atomicMax(reinterpret_cast<int*>(addr), __float_as_int(v));
    else atomicMin(reinterpret_cast<unsigned*>(addr), __float_as_uint(v));
}
__global__ void minmax_init_kernel(float* work, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) { work[2 * i] = INFINITY; work[2 * i + 1] = -INFINITY; }
}
__global__ __launch_bounds__(256) void dissum_kernel(const float* __restrict__ logits, float* __restrict__ score,
                                                     float* work, int K, int64_t HW, float clip, int inclusive) {
    __shared__ float smin[4], smax[4];
    const int b = blockIdx.y;
    float lo = INFINITY, hi = -INFINITY;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < HW;
         pix += (int64_t)gridDim.x * blockDim.x) {
        const float* src = logits + (int64_t)b * K * HW + pix;
        float s = 0.f;
        for (int k = 0; k < K; ++k) s += src[(int64_t)k * HW];
        s = -s;
        if (inclusive ? (s >= clip) : (s > clip)) s = clip;
        score[(int64_t)b * HW + pix] = s;
        lo = fminf(lo, s);
        hi = fmaxf(hi, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        lo = fminf(lo, __shfl_xor(lo, o, 64));
        hi = fmaxf(hi, __shfl_xor(hi, o, 64));
    }
    if ((threadIdx.x & 63) == 0) { smin[threadIdx.x >> 6] = lo; smax[threadIdx.x >> 6] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 4; ++i) { lo = fminf(lo, smin[i]); hi = fmaxf(hi, smax[i]); }
        lo = fminf(smin[0], lo); hi = fmaxf(smax[0], hi);
        if (lo <= hi) { atomic_min_f(work + 2 * b, lo); atomic_max_f(work + 2 * b + 1, hi); }
    }
}
__global__ __launch_bounds__(256) void minmax_norm_kernel(float* __restrict__ score, const float* work,
                                                          int64_t HW) {
    const int b = blockIdx.y;
    const float lo = work[2 * b], hi = work[2 * b + 1];
    const float den = hi - lo;
    for (int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; pix < HW;
         pix += (int64_t)gridDim.x * blockDim.x) {
        float* p = score + (int64_t)b * HW + pix;
        *p = (*p - lo) / den;
    }
}

__global__ __launch_bounds__(256) void novel_relabel_kernel(const float* __restrict__ feats,
                                                            const float* __restrict__ logits,
                                                            const float* __restrict__ proto,
                                                            int64_t* __restrict__ preds, int B, int C, int K,
                                                            int64_t HW, float thresh, int64_t new_label) {
    const int64_t total = (int64_t)B * HW;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / HW, pix = i - b * HW;
        float d = 0.f;
        for (int c = 0; c < C; ++c) {
            const float t = feats[i * C + c] - proto[c];
            d += t * t;
        }
        d = -d;
        const float* src = logits + b * K * HW + pix;
        float best = src[0];
        for (int k = 1; k < K; ++k) best = fmaxf(best, src[(int64_t)k * HW]);
        if (d > thresh && d > best) preds[i] = new_label;
    }
}

// ---- DML loss
// block partial = (sum nll, #valid, sum -logit_y over valid, #correct)
template <int PX>
__global__ __launch_bounds__(256) void loss_fwd_kernel(const float* __restrict__ logits,
                                                       const int64_t* __restrict__ labels,
                                                       float* __restrict__ partials, int B, int K, int64_t HW,
                                                       int64_t ignore_index) {
    __shared__ float sh[4][4];
    float s_nll = 0.f, s_cnt = 0.f, s_var = 0.f, s_ok = 0.f;
    const int64_t gpi = HW / PX, total = (int64_t)B * gpi;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / gpi, pix = (i - b * gpi) * PX;
        int64_t lab[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) lab[p] = labels[b * HW + pix + p];
        const float* src = logits + b * K * HW + pix;
        float mx[PX], own[PX], den[PX];
        int bi[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) { mx[p] = -INFINITY; own[p] = 0.f; den[p] = 0.f; bi[p] = 0; }
        for (int k = 0; k < K; ++k) {
            float v[PX];
            if constexpr (PX == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)k * HW);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                v[0] = src[(int64_t)k * HW];
            }
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                // online log-sum-exp: one pass over the K planes
                if (v[p] > mx[p]) { den[p] = den[p] * expf(mx[p] - v[p]) + 1.f; mx[p] = v[p]; bi[p] = k; }
                else den[p] += expf(v[p] - mx[p]);
                if ((int64_t)k == lab[p]) own[p] = v[p];
            }
        }
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            if (lab[p] == ignore_index) continue;
            s_nll += (mx[p] - own[p]) + logf(den[p]);
            s_cnt += 1.f;
            s_var += -own[p];
            s_ok += (bi[p] == (int)lab[p]) ? 1.f : 0.f;
        }
    }
    s_nll = wave_sum(s_nll); s_cnt = wave_sum(s_cnt); s_var = wave_sum(s_var); s_ok = wave_sum(s_ok);
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        sh[wv][0] = s_nll; sh[wv][1] = s_cnt; sh[wv][2] = s_var; sh[wv][3] = s_ok;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        partials[(int64_t)blockIdx.x * 4 + q] = sh[0][q] + sh[1][q] + sh[2][q] + sh[3][q];
    }
}
__global__ __launch_bounds__(256) void loss_sum_kernel(const float* __restrict__ partials, int nblocks,
                                                       double* sums, double inv_hw) {
    __shared__ double sh[4][4];
    double acc[4] = {0, 0, 0, 0};
    for (int i = threadIdx.x; i < nblocks; i += blockDim.x)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += (double)partials[(int64_t)i * 4 + q];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = wave_sum_d(acc[q]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 4; ++q) sh[threadIdx.x >> 6][q] = acc[q];
    __syncthreads();
    if (threadIdx.x < 4) {
        const int q = threadIdx.x;
        double v = sh[0][q] + sh[1][q] + sh[2][q] + sh[3][q];
        if (q == 2) v *= inv_hw;       // VAR = sum_i (1/HW_i) sum_valid(-logit_y), HW_i identical in a batch
        sums[q] = v;
    }
}
__global__ void loss_finalize_kernel(const double* sums, float* loss, float alpha, float n_images) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const double ce = sums[0] / sums[1];          // NaN when nothing is valid, as torch
        const double n = n_images > 0.f ? (double)n_images : sums[4];      // data parallel: global image count on the device
        *loss = (float)((ce + (double)alpha * sums[2]) / n);
    }
}
template <int PX>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const float* __restrict__ logits,
                                                       const int64_t* __restrict__ labels,
                                                       const double* __restrict__ sums,
                                                       const float* __restrict__ gout,
                                                       float* __restrict__ glogits, int B, int K, int64_t HW,
                                                       int64_t ignore_index, float alpha, float n_images_arg) {
    const float go = gout ? *gout : 1.f;
    const float n_images = n_images_arg > 0.f ? n_images_arg : (float)sums[4];
    const float w_ce = go / ((float)sums[1] * n_images);
    const float w_var = go * alpha / ((float)HW * n_images);
    const int64_t gpi = HW / PX, total = (int64_t)B * gpi;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t b = i / gpi, pix = (i - b * gpi) * PX;
        int64_t lab[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) lab[p] = labels[b * HW + pix + p];
        const float* src = logits + b * K * HW + pix;
        float* dst = glogits + b * K * HW + pix;
        float mx[PX], den[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) { mx[p] = -INFINITY; den[p] = 0.f; }
        for (int k = 0; k < K; ++k) {
            float v[PX];
            if constexpr (PX == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)k * HW);
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                v[0] = src[(int64_t)k * HW];
            }
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                if (v[p] > mx[p]) { den[p] = den[p] * expf(mx[p] - v[p]) + 1.f; mx[p] = v[p]; }
                else den[p] += expf(v[p] - mx[p]);
            }
        }
        float inv[PX];
#pragma unroll
        for (int p = 0; p < PX; ++p) inv[p] = lab[p] == ignore_index ? 0.f : w_ce / den[p];
        for (int k = 0; k < K; ++k) {
            float v[PX], g[PX];
            if constexpr (PX == 4) {
                const float4 t = *reinterpret_cast<const float4*>(src + (int64_t)k * HW);   // L2-resident re-read
                v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
            } else {
                v[0] = src[(int64_t)k * HW];
            }
#pragma unroll
            for (int p = 0; p < PX; ++p) {
                g[p] = expf(v[p] - mx[p]) * inv[p];
                if ((int64_t)k == lab[p] && lab[p] != ignore_index) g[p] -= w_ce + w_var;
            }
            if constexpr (PX == 4) *reinterpret_cast<float4*>(dst + (int64_t)k * HW) = make_float4(g[0], g[1], g[2], g[3]);
            else dst[(int64_t)k * HW] = g[0];
        }
    }
}


// ---- fused backward of loss + distance head + final x4 upsample (C = K = 16, H = 4h, W = 4w).
// The unfused chain moves 394 B per full-resolution pixel (loss_bwd writes d loss / d logits, proto_dist_bwd reads it and
// writes d loss / d features, bilinear_bwd reads that); none of the two intermediates is needed by anyone else.  Here a
// workgroup owns a TJ x TI tile of the LOW-resolution embedding gradient.  Phase 1: for the (4 TJ + 4) x (4 TI + 8)
// full-resolution pixels that reach the tile, one work item = 4 consecutive pixels of a row: features (64 B) and label in,
// logits recomputed (-|f - m_k|^2, same arithmetic as the forward), softmax, d loss / d logits, d loss / d features, and
// the row's bilinear weights folded at once into three low-resolution column partials (left neighbour, own, right
// neighbour) that go to an LDS row buffer in three conflict-free sub-phases (deterministic: no atomics).  Phase 2: each
// thread gathers the <= 8 rows that reach its low-resolution pixel for 8 channels and writes 16 bytes.
// Traffic: 72 B per pixel (+ halo re-reads that hit in L2) + the low-resolution result.
constexpr int HB_TJ = 7, HB_TI = 14;                 // tile of the low-resolution map per workgroup
constexpr int HB_ROWS = 4 * HB_TJ + 4;               // 32 full-resolution rows reach it
constexpr int HB_GRPS = HB_TI + 2;                   // 16 groups of 4 pixels per row (one halo group each side)

template <typename TO>
__global__ __launch_bounds__(256) void head_bwd_fused_c16_kernel(
    const float* __restrict__ feats, const int64_t* __restrict__ labels, const double* __restrict__ sums,
    const float* __restrict__ gout, const float* __restrict__ protos, TO* __restrict__ de, int B, int h, int w,
    int64_t ignore_index, float alpha, float n_images_arg) {
    constexpr int C = 16, K = 16;
    __shared__ __attribute__((aligned(16))) float t[HB_ROWS][HB_GRPS][C];      // 32 KB: x-reduced rows
    const int tid = threadIdx.x;
    const int H = 4 * h, W = 4 * w;
    const int tiles_x = (w + HB_TI - 1) / HB_TI, tiles_y = (h + HB_TJ - 1) / HB_TJ;
    int bid = blockIdx.x;
    const int tx = bid % tiles_x;
    bid /= tiles_x;
    const int ty = bid % tiles_y, b = bid / tiles_y;
    const int j0 = ty * HB_TJ, i0 = tx * HB_TI;
    const int Y0 = 4 * j0 - 2;                         // first full-resolution row of the region
    const int q0 = i0 - 1;                             // first pixel group of the region
    const float go = gout ? *gout : 1.f;
    const float n_images = n_images_arg > 0.f ? n_images_arg : (float)sums[4];
    const float w_ce = go / ((float)sums[1] * n_images);
    const float w_var = go * alpha / ((float)((int64_t)H * W) * n_images);
    // DIAGONAL prototype matrix (the reference's centers are 3 * I, network/utils.py:103-106): the distances collapse to
    // |f|^2 - 2 d_k f_k + d_k^2, whose |f|^2 cancels in the softmax, and sum_k g_k m_k to g_c d_c: ~250 VALU instructions per
    // pixel instead of ~1000 (the general path is VALU-bound at 18 % of the HBM roof).  Checked here, wave-uniformly, from the
    // 256 scalars themselves -- no flag in the ABI, any other matrix takes the general path.
    bool diag = true;
    float dk[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const float v = protos[k * C + c];
            if (c == k) dk[k] = v;
            else diag = diag && (v == 0.f);
        }
    }

    // two passes of 256 work items (16 rows x 16 groups each); a pass owns its rows of `t`, so the passes do not interact
#pragma unroll 1
    for (int it = 0; it < 2; ++it) {
        const int id = it * 256 + tid;
        const int r = id / HB_GRPS, g = id - r * HB_GRPS;
        const int Y = Y0 + r, q = q0 + g;
        float pl[C], pc[C], pr[C];                     // column partials: left neighbour (q - 1), own (q), right (q + 1)
#pragma unroll
        for (int c = 0; c < C; ++c) { pl[c] = 0.f; pc[c] = 0.f; pr[c] = 0.f; }
        if (Y >= 0 && Y < H && q >= 0 && q < w) {
            // halo groups only reach the tile through their inner two pixels
            const int p_lo = (g == 0) ? 2 : 0, p_hi = (g == HB_GRPS - 1) ? 2 : 4;
            const float* frow = feats + (((int64_t)b * H + Y) * W + 4 * q) * C;
            const int64_t* lrow = labels + ((int64_t)b * H + Y) * W + 4 * q;
            // the next pixel's features / label are requested before the current pixel's ~1000 VALU instructions
            float4 nf[4];
            int64_t nlab = lrow[p_lo];
#pragma unroll
            for (int c4 = 0; c4 < 4; ++c4) nf[c4] = *reinterpret_cast<const float4*>(frow + p_lo * C + c4 * 4);
#pragma unroll 1
            for (int p = p_lo; p < p_hi; ++p) {
                // keep the prototype loads (256 wave-uniform scalars) inside the iteration, 4 prototypes at a time:
                // hoisted out of the loop they need 256 SGPRs and spill
                asm volatile("" ::: "memory");
                float f[C];
#pragma unroll
                for (int c4 = 0; c4 < 4; ++c4) {
                    f[c4 * 4] = nf[c4].x; f[c4 * 4 + 1] = nf[c4].y; f[c4 * 4 + 2] = nf[c4].z; f[c4 * 4 + 3] = nf[c4].w;
                }
                const int64_t lab = nlab;
                if (p + 1 < p_hi) {
                    nlab = lrow[p + 1];
#pragma unroll
                    for (int c4 = 0; c4 < 4; ++c4) nf[c4] = *reinterpret_cast<const float4*>(frow + (p + 1) * C + c4 * 4);
                }
                float lg[K];
                float mx = -INFINITY;
                const bool valid = lab != ignore_index;
                float gs = 0.f, gm[C];
                if (diag) {
                    // logit_k - logit_j = (2 d_k f_k - d_k^2) - (2 d_j f_j - d_j^2): softmax over t_k, |f|^2 drops out
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        lg[k] = dk[k] * (2.f * f[k] - dk[k]);
                        mx = fmaxf(mx, lg[k]);
                    }
                    float den = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        lg[k] = __expf(lg[k] - mx);          // v_exp_f32 (2 ulp): the softmax weights carry 1e-7 relative
                        den += lg[k];
                    }
                    const float inv = valid ? w_ce / den : 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        float gk = lg[k] * inv;
                        if (valid && (int64_t)k == lab) gk -= w_ce + w_var;
                        gs += gk;
                        gm[k] = gk * dk[k];                  // sum_k g_k m_k has one term per channel
                    }
                } else {
#pragma unroll 4
                    for (int k = 0; k < K; ++k) {          // prototypes: wave-uniform addresses -> scalar loads
                        float d = 0.f;
#pragma unroll
                        for (int c = 0; c < C; ++c) {
                            const float u = f[c] - protos[k * C + c];
                            d += u * u;
                        }
                        lg[k] = -d;
                        mx = fmaxf(mx, lg[k]);
                    }
                    float den = 0.f;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        lg[k] = expf(lg[k] - mx);
                        den += lg[k];
                    }
                    const float inv = valid ? w_ce / den : 0.f;
                    // d loss / d features = -2 sum_k g_k (f - m_k) = -2 (f sum_k g_k - sum_k g_k m_k)
#pragma unroll
                    for (int c = 0; c < C; ++c) gm[c] = 0.f;
#pragma unroll 4
                    for (int k = 0; k < K; ++k) {
                        float gk = lg[k] * inv;
                        if (valid && (int64_t)k == lab) gk -= w_ce + w_var;
                        gs += gk;
#pragma unroll
                        for (int c = 0; c < C; ++c) gm[c] += gk * protos[k * C + c];
                    }
                }
                // bilinear weights of full-resolution column X = 4 q + p (align_corners = False, scale 1/4)
                const int X = 4 * q + p;
                float sX = 0.25f * ((float)X + 0.5f) - 0.5f;
                sX = sX < 0.f ? 0.f : sX;
                const int x0 = min((int)sX, w - 1), x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float l1 = sX - (float)x0, l0 = 1.f - l1;
                // columns x0 / x1 are q - 1, q or q + 1
                const float wl = (x0 == q - 1 ? l0 : 0.f) + (x1 == q - 1 ? l1 : 0.f);
                const float wc = (x0 == q ? l0 : 0.f) + (x1 == q ? l1 : 0.f);
                const float wr = (x0 == q + 1 ? l0 : 0.f) + (x1 == q + 1 ? l1 : 0.f);
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const float dfc = -2.f * (gs * f[c] - gm[c]);
                    pl[c] += wl * dfc;
                    pc[c] += wc * dfc;
                    pr[c] += wr * dfc;
                }
            }
        }
        // three sub-phases: every (row, column) cell of this pass's rows is written by exactly one thread in each
#pragma unroll
        for (int c = 0; c < C; c += 4)
            *reinterpret_cast<float4*>(&t[r][g][c]) = make_float4(pc[c], pc[c + 1], pc[c + 2], pc[c + 3]);
        __syncthreads();
        if (g + 1 < HB_GRPS) {
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                float4 v = *reinterpret_cast<float4*>(&t[r][g + 1][c]);
                v.x += pr[c]; v.y += pr[c + 1]; v.z += pr[c + 2]; v.w += pr[c + 3];
                *reinterpret_cast<float4*>(&t[r][g + 1][c]) = v;
            }
        }
        __syncthreads();
        if (g >= 1) {
#pragma unroll
            for (int c = 0; c < C; c += 4) {
                float4 v = *reinterpret_cast<float4*>(&t[r][g - 1][c]);
                v.x += pl[c]; v.y += pl[c + 1]; v.z += pl[c + 2]; v.w += pl[c + 3];
                *reinterpret_cast<float4*>(&t[r][g - 1][c]) = v;
            }
        }
    }
    __syncthreads();

    // phase 2: low-resolution pixel (j, i) x 8 channels per thread (98 pixels x 2 halves = 196 threads)
    const int px = tid >> 1, half = tid & 1;
    if (px >= HB_TJ * HB_TI) return;
    const int jj = px / HB_TI, ii = px - jj * HB_TI;
    const int j = j0 + jj, i = i0 + ii;
    if (j >= h || i >= w) return;
    float acc[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[c] = 0.f;
    for (int dy = 0; dy < 8; ++dy) {
        const int Y = 4 * j - 2 + dy;
        if (Y < 0 || Y >= H) continue;
        float sY = 0.25f * ((float)Y + 0.5f) - 0.5f;
        sY = sY < 0.f ? 0.f : sY;
        const int y0 = min((int)sY, h - 1), y1 = y0 + (y0 < h - 1 ? 1 : 0);
        const float l1 = sY - (float)y0;
        const float wy = (y0 == j ? 1.f - l1 : 0.f) + (y1 == j ? l1 : 0.f);
        if (wy == 0.f) continue;
        const float* src = &t[Y - Y0][ii + 1][half * 8];
        const float4 a = *reinterpret_cast<const float4*>(src), c4 = *reinterpret_cast<const float4*>(src + 4);
        acc[0] += wy * a.x; acc[1] += wy * a.y; acc[2] += wy * a.z; acc[3] += wy * a.w;
        acc[4] += wy * c4.x; acc[5] += wy * c4.y; acc[6] += wy * c4.z; acc[7] += wy * c4.w;
    }
    TO* dst = de + (((int64_t)b * h + j) * w + i) * C + half * 8;
    if constexpr (sizeof(TO) == 2) {
        *reinterpret_cast<uint4*>(dst) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]),
                                                    pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    } else {
        *reinterpret_cast<float4*>(dst) = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(dst + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

}  // namespace

extern "C" int dml_proto_dist_fwd(const float* x_nchw, const float* protos, float* logits, float* feats,
                                  uint8_t* argmax, float* dissum, int B, int C, int K, int H, int W,
                                  void* stream) {
    if (!x_nchw || !protos || B <= 0 || H <= 0 || W <= 0) return DML_EINVAL;
    if (C <= 0 || C > MAXC || K <= 0 || K > MAXK) return DML_EUNSUPPORTED;
    const int64_t HW = (int64_t)H * W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (HW % 4 == 0 && C == 16 && K == 16) {
        const int64_t groups = (int64_t)B * HW / 4;
        constexpr bool nt = DIST_NT_DEFAULT;
        if (nt)
            hipLaunchKernelGGL(proto_dist_fwd_c16_kernel<true>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                               x_nchw, protos, logits, feats, argmax, dissum, B, HW);
        else
            hipLaunchKernelGGL(proto_dist_fwd_c16_kernel<false>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st,
                               x_nchw, protos, logits, feats, argmax, dissum, B, HW);
    } else if (HW % 4 == 0) {
        const int grid = grid_for((int64_t)B * HW / 4, 256, 256 * 16);
        if (C <= 16) hipLaunchKernelGGL((proto_dist_fwd_kernel<4, 16>), dim3(grid), dim3(256), 0, st, x_nchw, protos, logits, feats,
                           argmax, dissum, B, C, K, HW);
        else hipLaunchKernelGGL((proto_dist_fwd_kernel<4, 32>), dim3(grid), dim3(256), 0, st, x_nchw, protos, logits, feats,
                           argmax, dissum, B, C, K, HW);
    } else {
        const int grid = grid_for((int64_t)B * HW, 256, 256 * 16);
        if (C <= 16) hipLaunchKernelGGL((proto_dist_fwd_kernel<1, 16>), dim3(grid), dim3(256), 0, st, x_nchw, protos, logits, feats,
                           argmax, dissum, B, C, K, HW);
        else hipLaunchKernelGGL((proto_dist_fwd_kernel<1, 32>), dim3(grid), dim3(256), 0, st, x_nchw, protos, logits, feats,
                           argmax, dissum, B, C, K, HW);
    }
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_upsample_dist_fwd(const float* e, const float* protos, float* logits, float* feats,
                                     uint8_t* argmax, float* dissum, int B, int h, int w, int C, int K, int H,
                                     int W, void* stream) {
    if (!e || !protos || B <= 0 || h <= 0 || w <= 0 || H <= 0 || W <= 0) return DML_EINVAL;
    if (C <= 0 || C > MAXC || K <= 0 || K > MAXK) return DML_EUNSUPPORTED;
    const float sy = (float)h / (float)H, sx = (float)w / (float)W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    constexpr bool staged = true;      // (the gathered variant it replaced: profiles/r02_dist_variants.txt)
    if (staged && H == 4 * h && W == 4 * w && C == 16 && K == 16 && argmax == nullptr && dissum == nullptr) {
        constexpr bool nt4 = DIST_NT_DEFAULT;
        const int64_t blocks = (int64_t)B * (h + 1) * ((W + 255) / 256);
        if (blocks >= (1ll << 31)) return DML_EINVAL;
        if (nt4)
            hipLaunchKernelGGL(upsample4_dist_fwd_c16_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, st, e, protos, logits,
                               feats, B, h, w);
        else
            hipLaunchKernelGGL(upsample4_dist_fwd_c16_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, st, e, protos, logits,
                               feats, B, h, w);
    } else if (W % 4 == 0 && C == 16 && K == 16 && argmax == nullptr && dissum == nullptr) {
        const int64_t groups = (int64_t)B * H * (W / 4);
        constexpr bool nt = DIST_NT_DEFAULT;
        if (nt)
            hipLaunchKernelGGL(upsample_dist_fwd_c16_kernel<true>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, e,
                               protos, logits, feats, B, h, w, H, W, sy, sx);
        else
            hipLaunchKernelGGL(upsample_dist_fwd_c16_kernel<false>, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, e,
                               protos, logits, feats, B, h, w, H, W, sy, sx);
    } else if (W % 4 == 0) {
        const int grid = grid_for((int64_t)B * H * (W / 4), 256, 256 * 16);
        if (C <= 16) hipLaunchKernelGGL((upsample_dist_fwd_kernel<4, 16>), dim3(grid), dim3(256), 0, st, e, protos, logits, feats,
                           argmax, dissum, B, h, w, C, K, H, W, sy, sx);
        else hipLaunchKernelGGL((upsample_dist_fwd_kernel<4, 32>), dim3(grid), dim3(256), 0, st, e, protos, logits, feats,
                           argmax, dissum, B, h, w, C, K, H, W, sy, sx);
    } else {
        const int grid = grid_for((int64_t)B * H * W, 256, 256 * 16);
        if (C <= 16) hipLaunchKernelGGL((upsample_dist_fwd_kernel<1, 16>), dim3(grid), dim3(256), 0, st, e, protos, logits, feats,
                           argmax, dissum, B, h, w, C, K, H, W, sy, sx);
        else hipLaunchKernelGGL((upsample_dist_fwd_kernel<1, 32>), dim3(grid), dim3(256), 0, st, e, protos, logits, feats,
                           argmax, dissum, B, h, w, C, K, H, W, sy, sx);
    }
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_proto_dist_bwd(const float* glogits, const float* gfeats, const float* feats,
                                  const float* protos, float* df, int B, int C, int K, int H, int W,
                                  void* stream) {
    if (!glogits || !feats || !protos || !df) return DML_EINVAL;
    if (C <= 0 || C > MAXC || K <= 0 || K > MAXK) return DML_EUNSUPPORTED;
    const int64_t HW = (int64_t)H * W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (HW % 4 == 0 && C == 16 && K == 16) {
        const int64_t groups = (int64_t)B * HW / 4;
        hipLaunchKernelGGL(proto_dist_bwd_c16_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, st, glogits,
                           gfeats, feats, protos, df, B, HW);
    } else if (HW % 4 == 0) {
        const int grid = grid_for((int64_t)B * HW / 4, 256, 256 * 16);
        if (C <= 16) hipLaunchKernelGGL((proto_dist_bwd_kernel<4, 16>), dim3(grid), dim3(256), 0, st, glogits, gfeats, feats, protos,
                           df, B, C, K, HW);
        else hipLaunchKernelGGL((proto_dist_bwd_kernel<4, 32>), dim3(grid), dim3(256), 0, st, glogits, gfeats, feats, protos,
                           df, B, C, K, HW);
    } else {
        const int grid = grid_for((int64_t)B * HW, 256, 256 * 16);
        if (C <= 16) hipLaunchKernelGGL((proto_dist_bwd_kernel<1, 16>), dim3(grid), dim3(256), 0, st, glogits, gfeats, feats, protos,
                           df, B, C, K, HW);
        else hipLaunchKernelGGL((proto_dist_bwd_kernel<1, 32>), dim3(grid), dim3(256), 0, st, glogits, gfeats, feats, protos,
                           df, B, C, K, HW);
    }
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_head_bwd_fused(const float* feats, const int64_t* labels, const double* sums, const float* gout,
                                  const float* protos, void* de, int B, int h, int w, int C, int K, int H, int W,
                                  int64_t ignore_index, float alpha, float n_images, int dtype, void* stream) {
    if (!feats || !labels || !sums || !protos || !de || B <= 0 || h <= 0 || w <= 0) return DML_EINVAL;
    if (dtype != DML_F32 && dtype != DML_BF16) return DML_EINVAL;
    if (C != 16 || K != 16 || H != 4 * h || W != 4 * w) return DML_EUNSUPPORTED;
    const int tiles = ((w + HB_TI - 1) / HB_TI) * ((h + HB_TJ - 1) / HB_TJ);
    if ((int64_t)tiles * B >= (1ll << 31)) return DML_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == DML_BF16)
        hipLaunchKernelGGL(head_bwd_fused_c16_kernel<bf16_t>, dim3(tiles * B), dim3(256), 0, st, feats, labels, sums, gout,
                           protos, static_cast<bf16_t*>(de), B, h, w, ignore_index, alpha, n_images);
    else
        hipLaunchKernelGGL(head_bwd_fused_c16_kernel<float>, dim3(tiles * B), dim3(256), 0, st, feats, labels, sums, gout,
                           protos, static_cast<float*>(de), B, h, w, ignore_index, alpha, n_images);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_argmax_msp(const float* logits, int64_t* preds, float* msp, int B, int K, int H, int W,
                              void* stream) {
    if (!logits || B <= 0 || K <= 0) return DML_EINVAL;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(argmax_msp_kernel, dim3(grid_for(B * HW, 256, 256 * 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), logits, preds, msp, B, K, HW);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_dissum_score(const float* logits, float* score, float* work, int B, int K, int H, int W,
                                float clip, int inclusive, void* stream) {
    if (!logits || !score || !work || B <= 0 || K <= 0) return DML_EINVAL;
    const int64_t HW = (int64_t)H * W;
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(minmax_init_kernel, dim3((B + 63) / 64), dim3(64), 0, st, work, B);
    dim3 grid(grid_for(HW, 256, 1024), B);
    hipLaunchKernelGGL(dissum_kernel, grid, dim3(256), 0, st, logits, score, work, K, HW, clip, inclusive);
    hipLaunchKernelGGL(minmax_norm_kernel, grid, dim3(256), 0, st, score, work, HW);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_novel_relabel(const float* feats, const float* logits, const float* proto, int64_t* preds,
                                 int B, int C, int K, int H, int W, float thresh, int64_t new_label,
                                 void* stream) {
    if (!feats || !logits || !proto || !preds) return DML_EINVAL;
    const int64_t HW = (int64_t)H * W;
    hipLaunchKernelGGL(novel_relabel_kernel, dim3(grid_for(B * HW, 256, 256 * 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), feats, logits, proto, preds, B, C, K, HW, thresh,
                       new_label);
    DML_LAUNCH_CHECK();
    return 0;
}

#define DML_LOSS_BLOCKS 2048

extern "C" int dml_loss_fwd(const float* logits, const int64_t* labels, double* sums, float* block_partials,
                            int B, int K, int H, int W, int64_t ignore_index, void* stream) {
    if (!logits || !labels || !sums || !block_partials || B <= 0 || K <= 0) return DML_EINVAL;
    const int64_t HW = (int64_t)H * W;
    const bool v4 = (HW % 4) == 0;
    const int grid = grid_for(B * HW / (v4 ? 4 : 1), 256, DML_LOSS_BLOCKS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (v4)
        hipLaunchKernelGGL(loss_fwd_kernel<4>, dim3(grid), dim3(256), 0, st, logits, labels, block_partials, B, K, HW,
                           ignore_index);
    else
        hipLaunchKernelGGL(loss_fwd_kernel<1>, dim3(grid), dim3(256), 0, st, logits, labels, block_partials, B, K, HW,
                           ignore_index);
    hipLaunchKernelGGL(loss_sum_kernel, dim3(1), dim3(256), 0, st, block_partials, grid, sums, 1.0 / (double)HW);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_loss_finalize(const double* sums, float* loss, float alpha, float n_images, void* stream) {
    if (!sums || !loss) return DML_EINVAL;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), sums, loss,
                       alpha, n_images);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_loss_bwd(const float* logits, const int64_t* labels, const double* sums, const float* gout,
                            float* glogits, int B, int K, int H, int W, int64_t ignore_index, float alpha,
                            float n_images, void* stream) {
    if (!logits || !labels || !sums || !glogits) return DML_EINVAL;
    const int64_t HW = (int64_t)H * W;
    if (HW % 4 == 0)
        hipLaunchKernelGGL(loss_bwd_kernel<4>, dim3(grid_for(B * HW / 4, 256, 256 * 32)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), logits, labels, sums, gout, glogits, B, K, HW,
                           ignore_index, alpha, n_images);
    else
        hipLaunchKernelGGL(loss_bwd_kernel<1>, dim3(grid_for(B * HW, 256, 256 * 32)), dim3(256), 0,
                           static_cast<hipStream_t>(stream), logits, labels, sums, gout, glogits, B, K, HW,
                           ignore_index, alpha, n_images);
    DML_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// confusion matrix (metrics/stream_metrics.py:49-55 of the reference): per-workgroup LDS histogram, int64 flush
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void confusion_kernel(const int64_t* __restrict__ lt, const int64_t* __restrict__ lp,
                                                        unsigned long long* __restrict__ hist, int64_t count, int n) {
    extern __shared__ uint32_t bins[];
    const int nb = n * n;
    for (int i = threadIdx.x; i < nb; i += 256) bins[i] = 0u;
    __syncthreads();
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    const int64_t pairs = count >> 1;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < pairs; i += (int64_t)gridDim.x * 256) {
        const ll2 t = reinterpret_cast<const ll2*>(lt)[i], p = reinterpret_cast<const ll2*>(lp)[i];
        if (t.x >= 0 && t.x < n && p.x >= 0 && p.x < n) atomicAdd(&bins[(int)t.x * n + (int)p.x], 1u);
        if (t.y >= 0 && t.y < n && p.y >= 0 && p.y < n) atomicAdd(&bins[(int)t.y * n + (int)p.y], 1u);
    }
    if ((count & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
        const int64_t t = lt[count - 1], p = lp[count - 1];
        if (t >= 0 && t < n && p >= 0 && p < n) atomicAdd(&bins[(int)t * n + (int)p], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nb; i += 256)
        if (bins[i]) atomicAdd(hist + i, (unsigned long long)bins[i]);
}
}  // namespace

extern "C" int dml_confusion_update(const int64_t* label_true, const int64_t* label_pred, int64_t* hist, int64_t count,
                                    int n_classes, void* stream) {
    if (!label_true || !label_pred || !hist || count < 0 || n_classes <= 0 || n_classes > 64) return DML_EINVAL;
    if ((reinterpret_cast<uintptr_t>(label_true) | reinterpret_cast<uintptr_t>(label_pred)) & 15) return DML_EALIGN;
    if (count == 0) return 0;
    // a workgroup's uint32 bins cannot overflow: it sees at most count / grid + 512 elements
    int64_t blocks = (count / 2 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    if (count / blocks >= (1ll << 31)) return DML_EUNSUPPORTED;
    hipLaunchKernelGGL(confusion_kernel, dim3((int)blocks), dim3(256), sizeof(uint32_t) * n_classes * n_classes,
                       static_cast<hipStream_t>(stream), label_true, label_pred, reinterpret_cast<unsigned long long*>(hist),
                       count, n_classes);
    DML_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------------
// masked feature sum for the few-shot prototypes (test_embedding.py:413-425 of the reference)
// ------------------------------------------------------------------------------------------------
namespace {
__global__ __launch_bounds__(256) void class_feature_sum_kernel(const float* __restrict__ feats, const int64_t* __restrict__ labels,
                                                                int64_t n_px, int C, int64_t class_id, double* __restrict__ sums,
                                                                unsigned long long* __restrict__ count) {
    __shared__ double sh[MAXC];
    __shared__ unsigned long long shn;
    if (threadIdx.x < MAXC) sh[threadIdx.x] = 0.0;
    if (threadIdx.x == 0) shn = 0ull;
    __syncthreads();
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    unsigned long long n = 0;
    for (int64_t p = (int64_t)blockIdx.x * 256 + threadIdx.x; p < n_px; p += (int64_t)gridDim.x * 256) {
        if (labels[p] != class_id) continue;
        ++n;
        const float* f = feats + p * C;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) acc[c] += f[c];
    }
    if (n) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) atomicAdd(&sh[c], (double)acc[c]);
        atomicAdd(&shn, n);
    }
    __syncthreads();
    if (threadIdx.x < C && sh[threadIdx.x] != 0.0) atomicAdd(sums + threadIdx.x, sh[threadIdx.x]);
    if (threadIdx.x == 0 && shn) atomicAdd(count, shn);
}
}  // namespace

extern "C" int dml_class_feature_sum(const float* feats, const int64_t* labels, int64_t n_px, int C, int64_t class_id,
                                     double* sums, unsigned long long* count, void* stream) {
    if (!feats || !labels || !sums || !count || n_px <= 0 || C <= 0 || C > MAXC) return DML_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(sums, 0, sizeof(double) * C, st) != hipSuccess || hipMemsetAsync(count, 0, 8, st) != hipSuccess)
        return DML_EINVAL;
    hipLaunchKernelGGL(class_feature_sum_kernel, dim3(grid_for(n_px, 256, 1024)), dim3(256), 0, st, feats, labels, n_px, C,
                       class_id, sums, count);
    DML_LAUNCH_CHECK();
    return 0;
}
