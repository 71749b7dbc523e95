// Implicit-GEMM convolution on MFMA for gfx950 (MI355X).
//
//   forward : y[m][n]  = sum_{r,s,c} x[src(m,r,s)][c] * w[n][r][s][c]          m = (b,yo,xo)
//   dgrad   : dx[m][n] = sum_{r,s,c} dy[srcT(m,r,s)][c] * wt[n][r][s][c]       m = (b,yi,xi), n = Cin
//   wgrad   : dw[n][r][s][c] += sum_m dy[m][n] * x[src(m,r,s)][c]
//
// Replaces nn.Conv2d at network/backbone/resnet.py:24-32,139 and network/utils.py:11-23,311,322,
// 337-352 of the reference (cuDNN there).  Layout NHWC / KRSC so that the GEMM K dimension is
// contiguous in both operands; 128-row pixel tiles; bf16 (v_mfma_f32_16x16x32_bf16) or exact fp32
// (v_mfma_f32_16x16x4_f32) with fp32 accumulation.  The MFMA "row" operand is the weight tile, fed in a permuted
// row order, so that every lane ends up with 16 consecutive output channels of one pixel (16-byte stores).
// Epilogue options, all on the fp32 accumulators: per-channel BatchNorm partial statistics (training forward),
// BatchNorm + residual + ReLU (inference), bias, accumulate, and -- data gradient -- the BN-backward partial sums
// of the tensor being written.
#include "common.h"
#include <cstdlib>

namespace {

constexpr int BK = 32;
constexpr int NTHREADS = 256;

typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));

struct ConvArgs {
    const void* x;
    const void* w;
    void* y;
    const float* bias;
    float* stats;
    int B, Hi, Wi, C, ldx;
    int Ho, Wo, N, ldy;
    int R, S, stride, dil, pad;
    int M, Ktot;
    int y_f32, accum;
    int nblk_n, nblk_m;
    FastDiv div_wo, div_howo, div_c;
    float* dbg;      // tuning builds only (ABL == 3): per-wave phase timings
    uint32_t x_bytes, w_bytes;   // operand extents for the buffer descriptors (fast path: both < 2^31)
    int w_tiled;                 // weights in the tile-major layout (DmlConvDesc::w_tiled): LDS-DMA kernels only
    int ws_min_tiles;            // DmlConvDesc::ws_min_tiles
    // f32_split == 2: operands as two fp16 planes of the power-of-two-scaled fp32 tensors (DmlConvDesc::x_planes ...)
    const void* x_planes;
    const void* w_planes;
    const float* x_unscale;      // device scalars: 1 / scale of the operand (dml_h2_split)
    const float* w_unscale;
    uint32_t x_plane_bytes, w_plane_bytes;      // distance from the hi plane to the lo plane
    // data-gradient mode: BN-backward partial sums of the tensor being written (DmlConvDesc::bnr_*)
    const void* bnr_y;
    const uint8_t* bnr_mask;
    const float* bnr_mean;
    const float* bnr_invstd;
    float* bnr_partials;
    float* bnr_gmax;
    int bnr_ldy, bnr_relu;
    // forward mode: inference epilogue (DmlConvDesc::post_*)
    const float* post_scale;
    const float* post_shift;
    const float* post_mean;
    const void* post_res;
    int post_ldres, post_relu;
    // K-split of the remainder tiles (DmlConvDesc::tail_*): tiles >= tail_full are computed by tail_q workgroups each
    float* tail_ws;
    int* tail_cnt;
    int tail_full, tail_q;
    int64_t tail_ws_elems_;      // host side only: capacities of the two buffers
    int tail_cnt_len_;
    // data-gradient mode: masked residual gradient added in the epilogue (DmlConvDesc::res_*)
    const void* res_dz;
    const uint8_t* res_mask;
    int res_ld;
    // data-gradient mode: fp32 sum of the earlier producers of this gradient, added before the one rounding (DmlConvDesc::acc32)
    const float* acc32;
    int acc32_ld;
    int f32_split;      // fp32 tensors: products through the three-term bf16 split (DmlConvDesc::f32_split)
    // bit 0: the epilogue's bf16 output stores carry the non-temporal hint, bit 1: its partial-statistics stores (launch_conv)
    int nt_out;
    int half_stagger;      // conv_ws_half_kernel: start delay of a CU's second workgroup, 10 ns ticks (launch_conv)
    // one parity class of a stride-2 data gradient as a stride-1 launch on dY's grid (DmlConvDesc::sub_grid): the launch's rows are
    // the pixels (b, 2 y2 + sub_y, 2 x2 + sub_x) of the B x 2 Ho x 2 Wo tensors y / res_dz / bnr_y / masks (ws_out_row)
    int pad_x;             // padding along the width (== pad unless DmlConvDesc::pad_w_set)
    int sub_grid, sub_y, sub_x;
    int bnr_inc;           // fused BN-backward sums over the launch's increment, not the stored total (DmlConvDesc::bnr_inc)
};

// physical pixel row of row m of the launch in the epilogue's tensors (y, accumulate / identity operand, bnr_y, masks); a row beyond
// M maps to the first row beyond those tensors (buffer accesses with the tensor's size as range drop it / read zeros, pointer
// accesses are guarded by their callers)
__device__ __forceinline__ uint32_t ws_out_row(const ConvArgs& a, const uint32_t m) {
    if (!a.sub_grid) return m;
    if (m >= (uint32_t)a.M) return 4u * (uint32_t)a.M;
    const uint32_t b = fdiv(m, a.div_howo);
    const uint32_t rem = m - b * (uint32_t)(a.Ho * a.Wo);
    const uint32_t y2 = fdiv(rem, a.div_wo);
    const uint32_t x2 = rem - y2 * (uint32_t)a.Wo;
    return ((b * 2u * (uint32_t)a.Ho + 2u * y2 + (uint32_t)a.sub_y) * 2u * (uint32_t)a.Wo) + 2u * x2 + (uint32_t)a.sub_x;
}
// rows of those tensors (range of the epilogue's buffer descriptors)
__device__ __forceinline__ uint32_t ws_out_rows(const ConvArgs& a) { return a.sub_grid ? 4u * (uint32_t)a.M : (uint32_t)a.M; }

constexpr int WGRAD_DEPTH = 2;      // K steps per barrier of the 256 x 256 weight-gradient kernel (one: 3-12 % slower)

// bijective XCD-aware remap: consecutive logical tiles land on the same XCD (private L2)
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + slot;
}

template <typename T> __device__ __forceinline__ int swz_chunk(int row, int chunk);
// bf16: 64-byte rows, ds_read_b128 lane groups {0-3,12-15,20-27}... -> conflict-free permutation
template <> __device__ __forceinline__ int swz_chunk<bf16_t>(int row, int chunk) {
    return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3);
}
// fp32: 128-byte rows, 8 chunks of 16 B
template <> __device__ __forceinline__ int swz_chunk<float>(int row, int chunk) { return chunk ^ (row & 7); }

// ------------------------------------------------------------------------------------------------
// Weight-tile row order.  MFMA tile i of a wave's TN = 16 NT channels does not take channels 16 i .. 16 i + 15:
// its row rho = 4 g + e (g = rho >> 2 is the accumulator's lane group, e the register) is channel
// g * (4 NT) + 4 i + e, so that after NT tiles every lane holds CL = 4 NT CONSECUTIVE channels of one pixel and the
// epilogue stores whole 16-byte vectors straight from the accumulators (a wave writes full 128-byte rows; no
// LDS staging, no barriers).  In LDS the tile stays in plain channel order; the fragment read just picks row
// b_row(i, rho), and the chunk swizzle is keyed on rho (recovered from the row by b_rho) so the bank pattern of
// the fragment reads is the conflict-free one of swz_chunk.
// ------------------------------------------------------------------------------------------------
// NT = 4 (wave tile 64 channels): the lane's 16 channels are two runs of 8, 32 apart -- tiles 2p, 2p+1 of lane group g are
// channels 32 p + 8 g .. 32 p + 8 g + 7 -- so that ONE store instruction (8 channels = 16 bytes per lane) writes 64
// contiguous bytes per pixel row (the four lane groups side by side) instead of four 16-byte pieces 32 bytes apart.
// (Throughput-neutral -- the L2 merges either pattern, and the store-heavy 1x1 layers sit at the ~2.7 TB/s the chip
// sustains for writes: 64 -> 256 channels at 192 x 192 writes 302 MB in 111 us -- but the epilogue needs 20-30 fewer
// VGPRs with it.)
template <int NT> __device__ __forceinline__ int frag_chan(int i, int g) {      // first channel of tile i in lane group g
    return NT == 4 ? (i >> 1) * 32 + g * 8 + (i & 1) * 4 : g * (4 * NT) + i * 4;
}
template <int NT> __device__ __forceinline__ int b_row(int i, int rho) { return frag_chan<NT>(i, rho >> 2) + (rho & 3); }
template <int NT> __device__ __forceinline__ int b_rho(int row) {
    return NT == 4 ? ((row >> 3) & 3) * 4 + (row & 3) : ((row / (4 * NT)) & 3) * 4 + (row & 3);
}

template <int CTRL> __device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
// sum over the 16 lanes of a DPP row (every lane gets the total): quad xor 1, xor 2, half mirror, row mirror
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_mov<0xB1>(v);
    v += dpp_mov<0x4E>(v);
    v += dpp_mov<0x141>(v);
    v += dpp_mov<0x140>(v);
    return v;
}

// 16-byte store, optionally with the non-temporal hint.  Why the hint: a conv output of 75-300 MB passes through 32 MB of
// L2; with plain stores the lines stay dirty until capacity evicts them, the evictions stall the allocating stores, the
// stalled stores block the CU's memory pipeline and the operand loads of the other workgroups queue behind them -- the
// store time ADDS to the compute time instead of hiding under it (1x1 256 -> 1024 at 48 x 48: 53 us, 30 us without the
// stores, 36 us with the stores aimed at an L2-resident region).  Non-temporal stores stream to memory: 36.4 us.
typedef unsigned int u32x4_st __attribute__((ext_vector_type(4)));
typedef float f32x4_st __attribute__((ext_vector_type(4)));
// (inline asm: with __builtin_nontemporal_store in one arm of a run-time branch the optimiser merges the two stores to the
// same address into one plain store and the hint is gone)
__device__ __forceinline__ void st16(void* p, uint32_t a, uint32_t b, uint32_t c, uint32_t d, const bool nt) {
    const u32x4_st v = {a, b, c, d};
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<u32x4_st*>(p) = v;
}
__device__ __forceinline__ void st16f(float* p, float a, float b, float c, float d, const bool nt) {
    const f32x4_st v = {a, b, c, d};
    if (nt) asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(v) : "memory");
    else *reinterpret_cast<f32x4_st*>(p) = v;
}

// ------------------------------------------------------------------------------------------------
// shared epilogue: BN partial statistics, bias, accumulate, fp32 / storage-dtype stores
// acc[i][j][e] = out[m = mw0 + j*16 + (lane&15)][n = nw0 + frag_chan<NT>(i, lane>>4) + e]
// ------------------------------------------------------------------------------------------------
template <typename T, int NT, int MT, int MODE, bool WIDE_MASK = true, bool STATS_ONLY = false>
__device__ __forceinline__ void conv_epilogue(f32x4 (&acc)[NT][MT], const ConvArgs& a, const int mw0, const int nw0,
                                              const int lr, const int lq) {
    constexpr int TM = MT * 16;
    constexpr int CL = NT * 4;
    auto ch = [&](int c) { return nw0 + frag_chan<NT>(c >> 2, lq) + (c & 3); };      // channel of the lane's c-th value

    if (a.stats != nullptr) {
        const int cnt = min(TM, max(0, a.M - mw0));
        if (cnt > 0) {
            const float inv = 1.0f / (float)cnt;
            const int grp = mw0 / TM;
            float* sg = a.stats + (int64_t)grp * a.N * 2;
            const bool pair_ok = (a.N & 1) == 0;
            // whole wave tile inside N: the partials leave in ONE store instruction -- lane (lq, lr = 2 i + h) writes the
            // (sum, M2) pairs of channels 2h, 2h+1 of tile i, 32 lanes x 16 B covering the wave's 64 channels contiguously --
            // instead of two 64-byte stores from four lanes per tile (8 instructions per wave: at ~1 us per 128 x 128 tile
            // the statistics cost 13-34 us per launch on the large maps)
            const bool gather = pair_ok && nw0 + NT * 16 <= a.N;
            float S[NT][4], Q[NT][4];
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const bool v = (mw0 + j * 16 + lr) < a.M;
#pragma unroll
                    for (int q = 0; q < 4; ++q) s[q] += v ? acc[i][j][q] : 0.f;
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] = row16_sum(s[q]);
                float m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const bool v = (mw0 + j * 16 + lr) < a.M;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float dlt = acc[i][j][q] - s[q] * inv;
                        m2[q] += v ? dlt * dlt : 0.f;
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) m2[q] = row16_sum(m2[q]);
#pragma unroll
                for (int q = 0; q < 4; ++q) { S[i][q] = s[q]; Q[i][q] = m2[q]; }
                if (lr == 0 && !gather) {
                    const int n = ch(i * 4);
                    float* sp = sg + (int64_t)n * 2 - i * 8;
                    if (pair_ok && n + 3 < a.N) {
                        st16f(sp + i * 8, s[0], m2[0], s[1], m2[1], (a.nt_out & 2) != 0);
                        st16f(sp + i * 8 + 4, s[2], m2[2], s[3], m2[3], (a.nt_out & 2) != 0);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (n + q < a.N) {
                                sp[(i * 4 + q) * 2] = s[q];
                                sp[(i * 4 + q) * 2 + 1] = m2[q];
                            }
                    }
                }
            }
            if (gather && lr < 2 * NT) {
                float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int h = 0; h < 2; ++h)
                        if (lr == 2 * i + h) { v0 = S[i][2 * h]; v1 = Q[i][2 * h]; v2 = S[i][2 * h + 1]; v3 = Q[i][2 * h + 1]; }
                const int n = nw0 + frag_chan<NT>(lr >> 1, lq) + (lr & 1) * 2;
                st16f(sg + (int64_t)n * 2, v0, v1, v2, v3, (a.nt_out & 2) != 0);
            }
        }
    }

    if constexpr (STATS_ONLY) return;      // (the caller stores the tile itself: conv_epilogue_rows)

    float bv[CL];
#pragma unroll
    for (int c = 0; c < CL; ++c) bv[c] = (MODE == 0 && a.bias != nullptr && ch(c) < a.N) ? a.bias[ch(c)] : 0.f;

    const bool out_f32 = a.y_f32 || sizeof(T) == 4;
    const uintptr_t yb = reinterpret_cast<uintptr_t>(a.y);
    const bool v4_ok = ((a.N & 3) == 0) && ((a.ldy & 3) == 0) && ((yb & (out_f32 ? 15 : 7)) == 0);
    const bool v8_ok = ((a.N & 7) == 0) && ((a.ldy & 7) == 0) && ((yb & 15) == 0);

    if constexpr (sizeof(T) == 2 && CL >= 8) {
        if (!out_f32 && v8_ok) {
            // bf16 rows of 16-byte vectors: 8-channel group outer, rows inner, so that the optional BN-backward sums
            // of the group (below) live in 32 registers.
            // Data gradient whose result is the output gradient dz of a BatchNorm(+ReLU): emit that BN's backward
            // partial sums (sum g, sum g * xhat per 64 rows, g = dz * relu') from the bf16-rounded values being
            // stored -- exactly what dml_bn_bwd_reduce would compute from the stored tensor -- so the separate pass
            // over dz / y / mask disappears.
            const bool bnr = MODE == 1 && a.bnr_partials != nullptr;
            const bool resm = MODE == 1 && a.res_dz != nullptr;
            constexpr bool a32 = MODE == 2;      // its own instantiation: the hot MODE 1 kernels keep their register budget
            const bool post = MODE == 0 && a.post_scale != nullptr;
            // ReLU-mask bits of the wave's whole 64 x 64 tile in ONE instruction per mask: lane L fetches row L's 64 bits (8
            // bytes), and lane (lr, lq) later takes byte 4 g + lq of row 16 j + lr through a lane permute -- instead of one
            // byte-load instruction per (j, g), each of which costs the texture addresser as much as a 1 KB load
            // (profiles/r03_epi_pmc.txt).  Wave tiles of 64 channels inside N only; the others load bytes.
            // (WIDE_MASK = false: the 256-row tile's kernel, at its 256-register limit, keeps the byte loads)
            const bool wide_mask = WIDE_MASK && MODE == 1 && NT == 4 && MT == 4 && (a.N & 63) == 0 && nw0 + 64 <= a.N;
            uint32_t rm_lo = 0xffffffffu, rm_hi = 0xffffffffu, bm_lo = 0xffffffffu, bm_hi = 0xffffffffu;
            if (MODE == 1 && wide_mask) {
                const int mrow = mw0 + lq * 16 + lr;              // lane L = lq * 16 + lr holds row L of the tile
                if (mrow < a.M) {
                    const int64_t mo = (int64_t)mrow * (a.N >> 3) + (nw0 >> 3);
                    if (resm) {
                        const uint2 v = *reinterpret_cast<const uint2*>(a.res_mask + mo);
                        rm_lo = v.x; rm_hi = v.y;
                    }
                    if (bnr && a.bnr_relu) {
                        const uint2 v = *reinterpret_cast<const uint2*>(a.bnr_mask + mo);
                        bm_lo = v.x; bm_hi = v.y;
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < CL / 8; ++g) {
                const int n8 = ch(g * 8);
                if (n8 >= a.N) continue;
                float r1[8], r2[8], rmu[8], ris[8];      // MODE 1: BN-backward sums; MODE 0: inference BN coefficients
                if (post) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        r1[e] = a.post_scale[n8 + e];
                        r2[e] = a.post_shift[n8 + e];
                        rmu[e] = a.post_mean[n8 + e];
                    }
                }
                if (bnr) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        r1[e] = 0.f;
                        r2[e] = 0.f;
                        rmu[e] = a.bnr_mean[n8 + e];
                        ris[e] = a.bnr_invstd[n8 + e];
                    }
                }
                // all of the group's loads (accumulate operand, BN input, mask bytes) are issued before the first
                // use: one memory round trip per group instead of one per row
                uint4 told[MT], ty[MT];      // (acc32: the row's eight floats, in told and ty -- never together with bnr)
                uint32_t bits[MT], rbits[MT];
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int m = mw0 + j * 16 + lr;
                    told[j] = make_uint4(0, 0, 0, 0);
                    ty[j] = make_uint4(0, 0, 0, 0);
                    bits[j] = 0xffu;
                    rbits[j] = 0xffu;
                    if (m < a.M) {
                        if (resm) {
                            told[j] = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.res_dz) +
                                                                      (int64_t)m * a.res_ld + n8);
                            if (!wide_mask) rbits[j] = a.res_mask[(int64_t)m * (a.N >> 3) + (n8 >> 3)];
                        }
                        if (post && a.post_res != nullptr)
                            told[j] = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.post_res) +
                                                                      (int64_t)m * a.post_ldres + n8);
                        else if (a32) {
                            // (32-bit element offset from the uniform base: the staging tensors stay below 2^32 bytes)
                            const uint32_t ao = (uint32_t)m * (uint32_t)a.acc32_ld + (uint32_t)n8;
                            told[j] = *reinterpret_cast<const uint4*>(a.acc32 + ao);
                            ty[j] = *reinterpret_cast<const uint4*>(a.acc32 + ao + 4);
                        } else if (a.accum)
                            told[j] = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.y) + (int64_t)m * a.ldy + n8);
                        if (bnr) {
                            ty[j] = *reinterpret_cast<const uint4*>(static_cast<const bf16_t*>(a.bnr_y) +
                                                                    (int64_t)m * a.bnr_ldy + n8);
                            if (a.bnr_relu && !wide_mask) bits[j] = a.bnr_mask[(int64_t)m * (a.N >> 3) + (n8 >> 3)];
                        }
                    }
                }
                if (MODE == 1 && wide_mask) {
                    // (executed by every lane: the permute reads another lane's register)
#pragma unroll
                    for (int j = 0; j < MT; ++j) {
                        const int src = j * 16 + lr;
                        if (resm) rbits[j] = ((uint32_t)__shfl((int)(g == 0 ? rm_lo : rm_hi), src) >> (lq * 8)) & 0xffu;
                        if (bnr && a.bnr_relu) bits[j] = ((uint32_t)__shfl((int)(g == 0 ? bm_lo : bm_hi), src) >> (lq * 8)) & 0xffu;
                    }
                }
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int m = mw0 + j * 16 + lr;
                    if (m >= a.M) continue;
                    float w[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) w[e] = acc[(g * 8 + e) >> 2][j][e & 3] + bv[g * 8 + e];
                    if (post) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] = (w[e] - rmu[e]) * r1[e] + r2[e];
                    }
                    bf16_t* yp = static_cast<bf16_t*>(a.y) + (int64_t)m * a.ldy + n8;
                    if (a32) {             // fp32 running sum of the earlier producers: this launch rounds the total once
                        w[0] += __uint_as_float(told[j].x); w[1] += __uint_as_float(told[j].y);
                        w[2] += __uint_as_float(told[j].z); w[3] += __uint_as_float(told[j].w);
                        w[4] += __uint_as_float(ty[j].x); w[5] += __uint_as_float(ty[j].y);
                        w[6] += __uint_as_float(ty[j].z); w[7] += __uint_as_float(ty[j].w);
                    } else {
                        const uint32_t tt[4] = {told[j].x, told[j].y, told[j].z, told[j].w};      // zeros unless accumulating
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (resm) {          // masked residual gradient: a cleared ReLU bit drops the element
                                w[2 * e] += (rbits[j] >> (2 * e)) & 1u ? __uint_as_float(tt[e] << 16) : 0.f;
                                w[2 * e + 1] += (rbits[j] >> (2 * e + 1)) & 1u ? __uint_as_float(tt[e] & 0xffff0000u) : 0.f;
                            } else {
                                w[2 * e] += __uint_as_float(tt[e] << 16);
                                w[2 * e + 1] += __uint_as_float(tt[e] & 0xffff0000u);
                            }
                        }
                    }
                    if (post && a.post_relu) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) w[e] = w[e] > 0.f ? w[e] : 0.f;
                    }
                    const uint32_t pk[4] = {pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]), pack_bf16x2(w[4], w[5]),
                                            pack_bf16x2(w[6], w[7])};
                    st16(yp, pk[0], pk[1], pk[2], pk[3], (a.nt_out & 1) != 0);
                    if (bnr) {
                        const uint32_t yy[4] = {ty[j].x, ty[j].y, ty[j].z, ty[j].w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float g0 = (bits[j] >> (2 * e)) & 1u ? __uint_as_float(pk[e] << 16) : 0.f;
                            const float g1 = (bits[j] >> (2 * e + 1)) & 1u ? __uint_as_float(pk[e] & 0xffff0000u) : 0.f;
                            r1[2 * e] += g0;
                            r2[2 * e] += g0 * (__uint_as_float(yy[e] << 16) - rmu[2 * e]) * ris[2 * e];
                            r1[2 * e + 1] += g1;
                            r2[2 * e + 1] += g1 * (__uint_as_float(yy[e] & 0xffff0000u) - rmu[2 * e + 1]) * ris[2 * e + 1];
                        }
                    }
                }
                if (bnr && mw0 < a.M) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        r1[e] = row16_sum(r1[e]);
                        r2[e] = row16_sum(r2[e]);
                    }
                    if (lr < 4) {            // lane lr writes the pairs of channels 2 lr, 2 lr + 1: one instruction per group
                        float* pp = a.bnr_partials + ((int64_t)(mw0 / TM) * a.N + n8) * 2;
                        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (lr == e) { v0 = r1[2 * e]; v1 = r2[2 * e]; v2 = r1[2 * e + 1]; v3 = r2[2 * e + 1]; }
                        st16f(pp + 4 * lr, v0, v1, v2, v3, (a.nt_out & 2) != 0);
                    }
                }
            }
            return;
        }
    }

#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int m = mw0 + j * 16 + lr;
        if (m >= a.M) continue;
        float v[CL];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) v[i * 4 + q] = acc[i][j][q] + bv[i * 4 + q];
        if (MODE == 0 && a.post_scale != nullptr) {          // inference epilogue on the generic path (fp32, odd shapes)
#pragma unroll
            for (int c = 0; c < CL; ++c) {
                if (ch(c) >= a.N) continue;
                float t = (v[c] - a.post_mean[ch(c)]) * a.post_scale[ch(c)] + a.post_shift[ch(c)];
                if (a.post_res != nullptr)
                    t += Elem<T>::ld(static_cast<const T*>(a.post_res) + (int64_t)m * a.post_ldres + ch(c));
                v[c] = a.post_relu ? (t > 0.f ? t : 0.f) : t;
            }
        }
        // yp + g * 4 below addresses the lane's g-th run of four channels: rebase per run through ch()
        const int64_t off = (int64_t)m * a.ldy;
        if (out_f32) {
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const int n = ch(g * 4);
                float* yp = static_cast<float*>(a.y) + off + n - g * 4;
                if (n >= a.N) continue;
                if (v4_ok) {
                    float4 o = make_float4(v[g * 4], v[g * 4 + 1], v[g * 4 + 2], v[g * 4 + 3]);
                    if (a.accum) {
                        const float4 t = *reinterpret_cast<const float4*>(yp + g * 4);
                        o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
                    }
                    *reinterpret_cast<float4*>(yp + g * 4) = o;
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < a.N) yp[g * 4 + q] = a.accum ? yp[g * 4 + q] + v[g * 4 + q] : v[g * 4 + q];
                }
            }
        } else {
#pragma unroll
            for (int g = 0; g < NT; ++g) {
                const int n = ch(g * 4);
                bf16_t* yp = static_cast<bf16_t*>(a.y) + off + n - g * 4;
                if (n >= a.N) continue;
                float* w = v + g * 4;
                if (v4_ok) {
                    if (a.accum) {
                        const uint2 t = *reinterpret_cast<const uint2*>(yp + g * 4);
                        w[0] += __uint_as_float(t.x << 16);
                        w[1] += __uint_as_float(t.x & 0xffff0000u);
                        w[2] += __uint_as_float(t.y << 16);
                        w[3] += __uint_as_float(t.y & 0xffff0000u);
                    }
                    *reinterpret_cast<uint2*>(yp + g * 4) = make_uint2(pack_bf16x2(w[0], w[1]), pack_bf16x2(w[2], w[3]));
                } else {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if (n + q < a.N) yp[g * 4 + q] = f32_to_bf16(a.accum ? bf16_to_f32(yp[g * 4 + q]) + w[q] : w[q]);
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// forward / data-gradient kernel
// ------------------------------------------------------------------------------------------------
template <typename T, int BN, bool ALIGNED, int MODE, int ABL = 0>   // ABL: tuning ablations (tools/bench_conv.py)
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(3))) void conv_igemm_kernel(const ConvArgs a) {
    constexpr int BM = 128;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int KV = BK / VEC;                 // 16-byte chunks per tile row
    constexpr int RPP = NTHREADS / KV;           // rows staged per pass
    constexpr int A_LD = BM / RPP;
    constexpr int B_LD = (BN + RPP - 1) / RPP;
    constexpr int TM = 64, TN = BN / 2;          // wave tile (2 x 2 waves)
    constexpr int MT = TM / 16, NT = TN / 16;

    __shared__ __attribute__((aligned(256))) T smem[2 * (BM + BN) * BK];
    auto As = [&](int buf) -> T* { return smem + buf * (BM + BN) * BK; };
    auto Bs = [&](int buf) -> T* { return smem + buf * (BM + BN) * BK + BM * BK; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    uint64_t t_start = 0;
    if constexpr (ABL == 3) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_start)::"memory");
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int blk_m = tile / a.nblk_n, blk_n = tile - blk_m * a.nblk_n;
    const int m0 = blk_m * BM, n0 = blk_n * BN;

    const T* __restrict__ X = static_cast<const T*>(a.x);
    const T* __restrict__ Wt = static_cast<const T*>(a.w);

    const int kvec = tid % KV, prow = tid / KV;
    constexpr int ES = (int)sizeof(T);
    constexpr uint32_t OOB = 0x80000000u;          // beyond any descriptor: the load returns zeros

    // ---- operand addressing.
    // ALIGNED (C % 32 == 0: a K tile never straddles a filter tap): everything per-thread is loop invariant -- a
    // signed byte offset of the row's tap-(0,0) pixel, a bit per filter tap saying whether that tap lands inside
    // the image (and, for the stride-2 data gradient, on an even position), and the weight row offset.  Per K step
    // the tap / channel advance is SCALAR; padding and tile edges are an out-of-range buffer offset, so the loads
    // are branch-free: 3 VALU per activation load, none per weight load.
    // Otherwise (stem 7x7 with 8 channels, odd channel counts, > 32 taps, > 2 GiB tensors): generic 64-bit gather.
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const int sh2 = (MODE != 0 && a.stride == 2) ? 1 : 0;

    int a_iy[A_LD], a_ix[A_LD], a_img[A_LD];       // generic path: pixel decomposition
    int a_base[A_LD];                               // fast path
    uint32_t a_mask[A_LD], b_base[B_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + prow + j * RPP;
        a_base[j] = 0;
        a_mask[j] = 0;
        if (m < a.M) {
            const uint32_t b = fdiv((uint32_t)m, a.div_howo);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
            const uint32_t yo = fdiv(rem, a.div_wo);
            const uint32_t xo = rem - yo * (uint32_t)a.Wo;
            if (MODE == 0) {
                a_iy[j] = (int)yo * a.stride - a.pad;
                a_ix[j] = (int)xo * a.stride - a.pad;
            } else {
                a_iy[j] = (int)yo + a.pad;
                a_ix[j] = (int)xo + a.pad;
            }
            a_img[j] = (int)b * a.Hi;
            if constexpr (ALIGNED) {
                const int by = MODE == 0 ? a_iy[j] : (a_iy[j] >> sh2), bx = MODE == 0 ? a_ix[j] : (a_ix[j] >> sh2);
                a_base[j] = (((a_img[j] + by) * a.Wi + bx) * a.ldx + kvec * VEC) * ES;
                uint32_t mk = 0;
                for (int r = 0, t = 0; r < a.R; ++r)
                    for (int q = 0; q < a.S; ++q, ++t) {
                        bool ok;
                        if (MODE == 0) {
                            const int ys = a_iy[j] + r * a.dil, xs = a_ix[j] + q * a.dil;
                            ok = ((unsigned)ys < (unsigned)a.Hi) && ((unsigned)xs < (unsigned)a.Wi);
                        } else {
                            const int ty = a_iy[j] - r * a.dil, tx = a_ix[j] - q * a.dil;
                            ok = (sh2 == 0 || (((ty | tx) & 1) == 0)) && ty >= 0 && tx >= 0 &&
                                 ((ty >> sh2) < a.Hi) && ((tx >> sh2) < a.Wi);
                        }
                        mk |= ok ? (1u << t) : 0u;
                    }
                a_mask[j] = mk;
            }
        } else {
            a_iy[j] = -(1 << 28);
            a_ix[j] = -(1 << 28);
            a_img[j] = 0;
        }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = prow + j * RPP, n = n0 + row;
        b_base[j] = (row < BN && n < a.N) ? (uint32_t)((n * a.Ktot + kvec * VEC) * ES) : OOB;
    }
    // scalar byte offset of filter tap (r, q) relative to tap (0, 0)
    auto tap_delta = [&](int r, int q) -> int {
        if (MODE == 0) return ((r * a.dil) * a.Wi + q * a.dil) * a.ldx * ES;
        return -((((r * a.dil) >> sh2) * a.Wi + ((q * a.dil) >> sh2)) * a.ldx) * ES;
    };

    uint4 a_reg[A_LD], b_reg[B_LD];
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

    int c0_staged = 0;          // channel offset of the tile currently held in a_reg (ABL == 4 only)
    auto load_tiles = [&](int kt, int tap_r, int tap_s, int c0) {
        c0_staged = c0;
        if constexpr (ALIGNED) {
            const uint32_t tapbit = 1u << (tap_r * a.S + tap_s);
            const int soff = tap_delta(tap_r, tap_s) + c0 * ES;
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                const uint32_t voff = (a_mask[j] & tapbit) ? (uint32_t)(a_base[j] + soff) : OOB;
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff, 0, 0);
                a_reg[j] = make_uint4(v.x, v.y, v.z, v.w);
            }
#pragma unroll
            for (int j = 0; j < B_LD; ++j) {
                const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)b_base[j], kt * BK * ES, 0);
                b_reg[j] = make_uint4(v.x, v.y, v.z, v.w);
            }
            return;
        }
        // ---- A operand (gathered activations)
        int r = tap_r, s = tap_s, c = c0 + kvec * VEC;
        bool kvalid = true;
        if (!ALIGNED) {
            const int k = kt * BK + kvec * VEC;
            kvalid = k < a.Ktot;
            const uint32_t tap = fdiv((uint32_t)k, a.div_c);
            c = k - (int)tap * a.C;
            r = (int)tap / a.S;
            s = (int)tap - r * a.S;
        }
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            int ys, xs;
            bool ok = kvalid;
            if (MODE == 0) {
                ys = a_iy[j] + r * a.dil;
                xs = a_ix[j] + s * a.dil;
            } else {
                const int ty = a_iy[j] - r * a.dil, tx = a_ix[j] - s * a.dil;
                if (a.stride == 2) {
                    ok = ok && (((ty | tx) & 1) == 0);
                    ys = ty >> 1;
                    xs = tx >> 1;
                } else {
                    ys = ty;
                    xs = tx;
                }
            }
            ok = ok && ((unsigned)ys < (unsigned)a.Hi) && ((unsigned)xs < (unsigned)a.Wi);
            if (ok) {
                const int64_t off = ((int64_t)(a_img[j] + ys) * a.Wi + xs) * a.ldx + c;
                a_reg[j] = *reinterpret_cast<const uint4*>(X + off);
            } else {
                a_reg[j] = make_uint4(0, 0, 0, 0);
            }
        }
        // ---- B operand (weights, K contiguous)
        const int kk = kt * BK + kvec * VEC;
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int n = n0 + prow + j * RPP;
            const bool ok = (prow + j * RPP < BN) && n < a.N && (ALIGNED || kk < a.Ktot);
            if (ok)
                b_reg[j] = *reinterpret_cast<const uint4*>(Wt + (int64_t)n * a.Ktot + kk);
            else
                b_reg[j] = make_uint4(0, 0, 0, 0);
        }
    };
    auto store_tiles = [&](int buf) {
        if constexpr (ABL == 4 && sizeof(T) == 2) {
            // feasibility probe for "normalise on load" (DESIGN.md, next round): a per-channel affine + ReLU applied to the
            // activation tile between its global load and the LDS write, as a consumer conv would apply the producer's
            // BatchNorm instead of reading a normalised copy.  Padding stays zero (all-zero pieces are left alone).
            const int cb = (c0_staged + kvec * VEC) % a.C;
            float sc[8], sh[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { sc[e] = a.bnr_mean[cb + e]; sh[e] = a.bnr_invstd[cb + e]; }      // probe: fields unused in forward mode
#pragma unroll
            for (int j = 0; j < A_LD; ++j) {
                uint32_t u[4] = {a_reg[j].x, a_reg[j].y, a_reg[j].z, a_reg[j].w};
                const bool pad = (u[0] | u[1] | u[2] | u[3]) == 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = fmaxf(fmaf(__uint_as_float(u[e] << 16), sc[2 * e], sh[2 * e]), 0.f);
                    const float hi = fmaxf(fmaf(__uint_as_float(u[e] & 0xffff0000u), sc[2 * e + 1], sh[2 * e + 1]), 0.f);
                    u[e] = pad ? 0u : pack_bf16x2(lo, hi);
                }
                a_reg[j] = make_uint4(u[0], u[1], u[2], u[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int row = prow + j * RPP;
            *reinterpret_cast<uint4*>(As(buf) + row * BK + swz_chunk<T>(row, kvec) * VEC) = a_reg[j];
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int row = prow + j * RPP;
            if ((BN % RPP) == 0 || row < BN)
                *reinterpret_cast<uint4*>(Bs(buf) + row * BK + swz_chunk<T>(b_rho<NT>(row), kvec) * VEC) = b_reg[j];
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int KT = (a.Ktot + BK - 1) / BK;
    int tap_r = 0, tap_s = 0, c0 = 0;
    auto advance = [&]() {
        if (ALIGNED) {
            c0 += BK;
            if (c0 >= a.C) {
                c0 = 0;
                if (++tap_s == a.S) { tap_s = 0; ++tap_r; }
            }
        }
    };

    load_tiles(0, tap_r, tap_s, c0);
    store_tiles(0);
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    int cur = 0;
    // ABL == 3: s_memtime stamps at the points where lgkmcnt(0) is harmless; deltas averaged over the K loop
    uint32_t ph[6] = {0, 0, 0, 0, 0, 0};
    uint64_t tprev = 0;
    auto stamp = [&](int i) {
        if constexpr (ABL == 3) {
            uint64_t t;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            __builtin_amdgcn_sched_barrier(0);
            if (i >= 0) ph[i] += (uint32_t)(t - tprev);
            tprev = t;
        }
    };
    stamp(-1);
    const uint64_t t_loop = tprev;
    for (int kt = 0; kt < KT; ++kt) {
        const bool has_next = kt + 1 < KT;
        if (has_next && ABL != 1) {
            advance();
            load_tiles(kt + 1, tap_r, tap_s, c0);
        }
        stamp(0);      // global loads issued
        const T* as = As(cur) + (wm * TM) * BK;
        const T* bs = Bs(cur) + (wn * TN) * BK;
        if constexpr (sizeof(T) == 2) {
            mfma_bf16x8 bf[NT], af[MT];
#pragma unroll
            for (int i = 0; i < NT; ++i)
                bf[i] = *reinterpret_cast<const mfma_bf16x8*>(bs + b_row<NT>(i, lr) * BK + swz_chunk<T>(lr, lq) * 8);
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = j * 16 + lr;
                af[j] = *reinterpret_cast<const mfma_bf16x8*>(as + row * BK + swz_chunk<T>(row, lq) * 8);
            }
            stamp(1);  // fragments in registers
            if constexpr (ABL == 2) {
#pragma unroll
                for (int i = 0; i < NT; ++i) asm volatile("" ::"v"(bf[i]));
#pragma unroll
                for (int j = 0; j < MT; ++j) asm volatile("" ::"v"(af[j]));
            } else {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[i], af[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float bf[NT], af[MT];
#pragma unroll
                for (int i = 0; i < NT; ++i) bf[i] = bs[b_row<NT>(i, lr) * BK + swz_chunk<T>(lr, kk) * 4 + lq];
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const int row = j * 16 + lr;
                    af[j] = as[row * BK + swz_chunk<T>(row, kk) * 4 + lq];
                }
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[i], af[j], acc[i][j], 0, 0, 0);
            }
        }
        stamp(2);      // MFMAs issued
        if constexpr (ABL == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp(3);      // next tile arrived in registers
        if (has_next) store_tiles(cur ^ 1);
        stamp(4);      // LDS writes drained
        __syncthreads();
        stamp(5);      // barrier released
        cur ^= 1;
    }
    const uint64_t t_end_loop = tprev;

    // Cut the accumulators' live ranges here: without it hipcc keeps the MFMA results un-tied through the
    // branchy epilogue and re-copies all 64 AGPRs (v_accvgpr_mov + s_nop) in EVERY K step (-35 % throughput).
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
    conv_epilogue<T, NT, MT, MODE>(acc, a, m0 + wm * TM, n0 + wn * TN, lr, lq);
    if constexpr (ABL == 3) {
        uint64_t t_end;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end)::"memory");
        if (a.dbg != nullptr && lane == 0) {
            float* o = a.dbg + ((int64_t)blockIdx.x * 4 + wave) * 8;
#pragma unroll
            for (int i = 0; i < 6; ++i) o[i] = (float)ph[i] / (float)KT;
            o[6] = (float)(uint32_t)(t_loop - t_start);
            o[7] = (float)(uint32_t)(t_end - t_end_loop);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 tensors, products on the bf16 matrix cores: three-term split.
// The exact mode's v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate.  Here every fp32 operand value x is split once,
// on its way from the staging registers into LDS, into hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid) (24 mantissa
// bits, each residual exact in fp32), LDS holds three bf16 planes per operand tile, and a 16 x 16 x 32 block is the six
// bf16 MFMAs whose terms are above 2^-26 of the product: lo*hi', hi*lo', mid*mid', mid*hi', hi*mid', hi*hi' (small terms
// first; every bf16 x bf16 product is exact in the fp32 accumulator).  96 MFMA cycles per block instead of 256; results at
// fp32 rounding level (NOT the reference's fp32 arithmetic bit for bit: DmlConvDesc.f32_split).
// One LDS buffer of 3 x (BM + BN) x 32 bf16 = 48 KB, two barriers per K step, global loads of the next tile in flight in
// registers meanwhile; 256-register budget = two workgroups per CU.  ALIGNED shapes only (C % 32 == 0).
// ------------------------------------------------------------------------------------------------
template <int BN, int MODE>
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2))) void conv_igemm_x3_kernel(const ConvArgs a) {
    typedef float T;
    constexpr int BM = 128, ES = 4;
    constexpr int KV = BK / 4;                   // 16-byte chunks (4 floats) per tile row
    constexpr int RPP = NTHREADS / KV;           // rows staged per pass
    constexpr int A_LD = BM / RPP, B_LD = BN / RPP;
    constexpr int TM = 64, TN = BN / 2, MT = TM / 16, NT = TN / 16;
    constexpr int PA = BM * BK, PB = BN * BK;    // bf16 elements per plane
    constexpr uint32_t OOB = 0x80000000u;
    __shared__ __attribute__((aligned(256))) bf16_t smem[3 * (PA + PB)];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tile = xcd_remap(blockIdx.x, gridDim.x);
    const int blk_m = tile / a.nblk_n, blk_n = tile - blk_m * a.nblk_n;
    const int m0 = blk_m * BM, n0 = blk_n * BN;
    const int kvec = tid % KV, prow = tid / KV;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)a.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)a.w_bytes, 0x00020000);
    const int sh2 = (MODE != 0 && a.stride == 2) ? 1 : 0;
    int a_base[A_LD];
    uint32_t a_mask[A_LD], b_base[B_LD];
#pragma unroll
    for (int j = 0; j < A_LD; ++j) {
        const int m = m0 + prow + j * RPP;
        a_base[j] = 0;
        a_mask[j] = 0;
        if (m < a.M) {
            const uint32_t b = fdiv((uint32_t)m, a.div_howo);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
            const uint32_t yo = fdiv(rem, a.div_wo);
            const uint32_t xo = rem - yo * (uint32_t)a.Wo;
            const int iy = MODE == 0 ? (int)yo * a.stride - a.pad : (int)yo + a.pad;
            const int ix = MODE == 0 ? (int)xo * a.stride - a.pad : (int)xo + a.pad;
            const int by = MODE == 0 ? iy : (iy >> sh2), bx = MODE == 0 ? ix : (ix >> sh2);
            a_base[j] = ((((int)b * a.Hi + by) * a.Wi + bx) * a.ldx + kvec * 4) * ES;
            uint32_t mk = 0;
            for (int r = 0, t = 0; r < a.R; ++r)
                for (int q = 0; q < a.S; ++q, ++t) {
                    bool ok;
                    if (MODE == 0) {
                        const int ys = iy + r * a.dil, xs = ix + q * a.dil;
                        ok = ((unsigned)ys < (unsigned)a.Hi) && ((unsigned)xs < (unsigned)a.Wi);
                    } else {
                        const int ty = iy - r * a.dil, tx = ix - q * a.dil;
                        ok = (sh2 == 0 || (((ty | tx) & 1) == 0)) && ty >= 0 && tx >= 0 && ((ty >> sh2) < a.Hi) &&
                             ((tx >> sh2) < a.Wi);
                    }
                    mk |= ok ? (1u << t) : 0u;
                }
            a_mask[j] = mk;
        }
    }
#pragma unroll
    for (int j = 0; j < B_LD; ++j) {
        const int row = prow + j * RPP, n = n0 + row;
        b_base[j] = (n < a.N) ? (uint32_t)((n * a.Ktot + kvec * 4) * ES) : OOB;
    }
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t a_reg[A_LD], b_reg[B_LD];
    int tap_r = 0, tap_s = 0, c0 = 0;
    const int KT = a.Ktot / BK;
    // loads of K step kt (kt >= KT: every lane out of range -- zeros, no memory access: keeps the loop branch-free), in three
    // parts so that each staging register can be refilled right after its split has been stored
    uint32_t ld_tapbit = 0, ld_boff = 0;
    int ld_soff = 0;
    bool ld_valid = false;
    auto tap_setup = [&](int kt) {
        ld_valid = kt < KT;
        ld_tapbit = ld_valid ? 1u << ((tap_r * a.S + tap_s) & 31) : 0u;
        ld_soff = (MODE == 0 ? ((tap_r * a.dil) * a.Wi + tap_s * a.dil) * a.ldx
                             : -((((tap_r * a.dil) >> sh2) * a.Wi + ((tap_s * a.dil) >> sh2)) * a.ldx)) * ES + c0 * ES;
        ld_boff = ld_valid ? (uint32_t)(kt * BK * ES) : 0u;
        c0 += BK;
        if (c0 >= a.C) {
            c0 = 0;
            if (++tap_s == a.S) { tap_s = 0; ++tap_r; }
        }
    };
    auto load_a = [&](int j) {
        if (j < A_LD) {
            const uint32_t voff = (a_mask[j] & ld_tapbit) ? (uint32_t)(a_base[j] + ld_soff) : OOB;
            a_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)voff, 0, 0);
        }
    };
    auto load_b = [&](int j) {
        if (j < B_LD) b_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_w, (int)(ld_valid ? b_base[j] : OOB), (int)ld_boff, 0);
    };
    auto load_tiles = [&](int kt) {
        tap_setup(kt);
#pragma unroll
        for (int j = 0; j < A_LD; ++j) load_a(j);
#pragma unroll
        for (int j = 0; j < B_LD; ++j) load_b(j);
    };
    // x -> (hi, mid, lo), four values at a time, packed as bf16 pairs; the 8-byte piece is half a 16-byte bf16 chunk
    auto split_store = [&](const u32x4_t v, bf16_t* plane0, int plane_elems, int off) {
        const float x[4] = {__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        uint32_t h[2], m[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
            const float r0 = x[2 * e] - __uint_as_float(h[e] << 16), r1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
            m[e] = pack_bf16x2(r0, r1);
            const float s0 = r0 - __uint_as_float(m[e] << 16), s1 = r1 - __uint_as_float(m[e] & 0xffff0000u);
            l[e] = pack_bf16x2(s0, s1);
        }
        *reinterpret_cast<uint2*>(plane0 + off) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(plane0 + plane_elems + off) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2*>(plane0 + 2 * plane_elems + off) = make_uint2(l[0], l[1]);
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int j = 0; j < A_LD; ++j) {
            const int row = prow + j * RPP;
            split_store(a_reg[j], smem, PA, row * BK + swz_chunk<bf16_t>(row, kvec >> 1) * 8 + (kvec & 1) * 4);
        }
#pragma unroll
        for (int j = 0; j < B_LD; ++j) {
            const int row = prow + j * RPP;
            split_store(b_reg[j], smem + 3 * PA, PB, row * BK + swz_chunk<bf16_t>(b_rho<NT>(row), kvec >> 1) * 8 + (kvec & 1) * 4);
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    load_tiles(0);
    store_tiles();
    __syncthreads();
    load_tiles(1);                                  // stays in registers until it is split under the MFMAs of K step 0
    const int lr = lane & 15, lq = lane >> 4;
    const bf16_t* as = smem + (wm * TM) * BK;
    const bf16_t* bs = smem + 3 * PA + (wn * TN) * BK;
    for (int kt = 0; kt < KT; ++kt) {
        // all 24 fragments of this K step first (96 registers) ...
        mfma_bf16x8 af[3][MT], bf[3][NT];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int row = j * 16 + lr;
                af[p][j] = *reinterpret_cast<const mfma_bf16x8*>(as + p * PA + row * BK + swz_chunk<bf16_t>(row, lq) * 8);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
                bf[p][i] = *reinterpret_cast<const mfma_bf16x8*>(bs + p * PB + b_row<NT>(i, lr) * BK + swz_chunk<bf16_t>(lr, lq) * 8);
        }
        __syncthreads();                        // ... so that the one LDS buffer is free while the MFMAs run
        auto mm = [&](int pb, int pa) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[pb][i], af[pa][j], acc[i][j], 0, 0, 0);
        };
        // six MFMA groups (small terms first), the split + store of the NEXT tile's eight staged chunks in between: ~2.4 VALU /
        // LDS-write instructions per MFMA, which issue in the shadow of the 16-cycle matrix operations
        auto st_a = [&](int j) {
            if (j < A_LD) {
                const int row = prow + j * RPP;
                split_store(a_reg[j], smem, PA, row * BK + swz_chunk<bf16_t>(row, kvec >> 1) * 8 + (kvec & 1) * 4);
            }
        };
        auto st_b = [&](int j) {
            if (j < B_LD) {
                const int row = prow + j * RPP;
                split_store(b_reg[j], smem + 3 * PA, PB, row * BK + swz_chunk<bf16_t>(b_rho<NT>(row), kvec >> 1) * 8 + (kvec & 1) * 4);
            }
        };
        // each staging register is refilled (K step kt + 2) as soon as its split has been stored: the load has the rest of
        // this step and the next step's fragment reads to land
        tap_setup(kt + 2);
        mm(2, 0); st_a(0); st_a(1); load_a(0); load_a(1);             // lo(w) * hi(x)
        mm(1, 1); st_a(2); st_a(3); load_a(2); load_a(3);             // mid * mid
        mm(1, 0); st_b(0); st_b(1); load_b(0); load_b(1);             // mid * hi
        mm(0, 2); st_b(2); st_b(3); load_b(2); load_b(3);             // hi * lo
        mm(0, 1);                                                     // hi * mid
        mm(0, 0);                                                     // hi * hi
        // pin the interleaving: one MFMA, then up to three VALU and one LDS write, a global load every eighth
#pragma unroll
        for (int q = 0; q < 16 * NT * MT / 4; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (q % (2 * NT) == 2 * NT - 1) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
    conv_epilogue<T, NT, MT, MODE>(acc, a, m0 + wm * TM, n0 + wn * TN, lr, lq);
}

// ------------------------------------------------------------------------------------------------
// bf16 forward / data-gradient kernel, LDS-DMA version (requires C % 32 == 0).
// Operand tiles go HBM -> LDS with buffer_load ... lds (no VGPR round trip, no ds_write): an NST-stage ring (3: 48 KB,
// three workgroups per CU), tiles issued NST-1 K-steps ahead, counted vmcnt waits and ONE barrier per K-step.  Zero padding / masked rows
// come from the buffer descriptor's bounds check (an out-of-range voffset writes zeros to LDS).  The LDS image
// is lane-linear, so the bank-conflict swizzle is applied to the per-lane SOURCE address.
// ------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// BM = 128: 4 waves (2 x 2), three workgroups per CU.  BM = 256: 8 waves (4 x 2) on the same 64 x 64 wave tile -- the
// weight tile is shared by twice the rows, so a K step moves 24 KB into LDS for 2.1 MFLOP (85 FLOP/B against 64) -- two
// workgroups per CU (72 KB of LDS each, 128-register budget).
// TMW = 128 (with BM = 256): 4 waves (2 x 2) on a 128 x 64 WAVE tile -- 32 MFMAs per wave and K step against 12 KB of
// fragment reads (the 64 x 64 wave tile: 16 against 8 KB), half the barriers per FLOP; 128 accumulator registers, two
// workgroups per CU = two waves per SIMD.
// PH (tuning builds only, dml_debug_conv_ablate 5): s_memtime stamps around the phases of a K step, per-wave sums to a.dbg
template <int BN, int MODE, int NST, int WPE = 3, int BM = 128, int TMW = 64, bool PH = false>
__global__ __launch_bounds__(BM / TMW * 128) __attribute__((amdgpu_waves_per_eu(WPE))) void conv_igemm_dma_kernel(const ConvArgs a, const uint32_t x_bytes,
                                                                  const uint32_t w_bytes) {
    typedef bf16_t T;
    constexpr int WAVES = BM / TMW * 2, NTH = WAVES * 64, LA = NST - 1;
    constexpr int STAGE = (BM + BN) * BK;          // elements per ring stage
    constexpr int TM = TMW, TN = BN / 2, MT = TM / 16, NT = TN / 16;
    constexpr int A_I = BM / 16 / WAVES, B_I = BN / 16 / WAVES;   // DMA instructions per wave per tile (1 KiB = 16 rows each)
    static_assert(B_I >= 1, "tile too narrow for this many waves");
    constexpr int NI = A_I + B_I;
    constexpr uint32_t OOB = 0x80000000u;

    __shared__ __attribute__((aligned(1024))) T smem[NST * STAGE];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // whole tiles first (XCD-aware order), then the K-split parts of the remainder tiles
    int tile, kpart = 0, kparts = 1;
    if (a.tail_q > 1) {
        if ((int)blockIdx.x < a.tail_full) {
            tile = xcd_remap(blockIdx.x, a.tail_full);
        } else {
            const int part = (int)blockIdx.x - a.tail_full;
            tile = a.tail_full + part / a.tail_q;
            kpart = part - (tile - a.tail_full) * a.tail_q;
            kparts = a.tail_q;
        }
    } else {
        tile = xcd_remap(blockIdx.x, gridDim.x);
    }
    const int blk_m = tile / a.nblk_n, blk_n = tile - blk_m * a.nblk_n;
    const int m0 = blk_m * BM, n0 = blk_n * BN;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.w), 0, (int)w_bytes, 0x00020000);

    // lane -> (row within a 16-row DMA piece, 16-byte chunk); source chunk carries the XOR swizzle
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((0x78 >> (((lane >> 4) & 3) * 2)) & 3);

    // per piece and lane: signed byte offset of the row's tap-(0,0) pixel (+ swizzled chunk) and one validity bit per
    // filter tap (image border, stride-2 parity of the data gradient) -- the per-K-step part is scalar (see
    // conv_igemm_kernel)
    const int sh2 = (MODE != 0 && a.stride == 2) ? 1 : 0;
    int a_base[A_I];
    uint32_t a_mask[A_I];
#pragma unroll
    for (int jj = 0; jj < A_I; ++jj) {
        const int m = m0 + (wave * A_I + jj) * 16 + prow;
        a_base[jj] = 0;
        a_mask[jj] = 0;
        if (m < a.M) {
            const uint32_t b = fdiv((uint32_t)m, a.div_howo);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
            const uint32_t yo = fdiv(rem, a.div_wo);
            const uint32_t xo = rem - yo * (uint32_t)a.Wo;
            const int iy = MODE == 0 ? (int)yo * a.stride - a.pad : (int)yo + a.pad;
            const int ix = MODE == 0 ? (int)xo * a.stride - a.pad : (int)xo + a.pad;
            const int by = MODE == 0 ? iy : (iy >> sh2), bx = MODE == 0 ? ix : (ix >> sh2);
            a_base[jj] = ((((int)b * a.Hi + by) * a.Wi + bx) * a.ldx + lchunk * 8) * 2;
            uint32_t mk = 0;
            for (int r = 0, t = 0; r < a.R; ++r)
                for (int q = 0; q < a.S; ++q, ++t) {
                    bool ok;
                    if (MODE == 0) {
                        const int ys = iy + r * a.dil, xs = ix + q * a.dil;
                        ok = ((unsigned)ys < (unsigned)a.Hi) && ((unsigned)xs < (unsigned)a.Wi);
                    } else {
                        const int ty = iy - r * a.dil, tx = ix - q * a.dil;
                        ok = (sh2 == 0 || (((ty | tx) & 1) == 0)) && ty >= 0 && tx >= 0 && ((ty >> sh2) < a.Hi) &&
                             ((tx >> sh2) < a.Wi);
                    }
                    mk |= ok ? (1u << t) : 0u;
                }
            a_mask[jj] = mk;
        }
    }
    uint32_t b_off[B_I];
#pragma unroll
    for (int jj = 0; jj < B_I; ++jj) {
        const int row = (wave * B_I + jj) * 16 + prow, n = n0 + row;
        const int bchunk = swz_chunk<T>(b_rho<NT>(row), lane & 3);      // weight rows: swizzle keyed on the MFMA row
        // tile-major weights [N / 64][K / 32][64][32]: the instruction's 16 rows x 64 bytes are one contiguous KB
        b_off[jj] = n >= a.N ? OOB
                    : a.w_tiled ? (uint32_t)(((int64_t)(n >> 6) * (a.Ktot / BK) * 2048 + (n & 63) * 32 + bchunk * 8) * 2)
                                : (uint32_t)(((int64_t)n * a.Ktot + bchunk * 8) * 2);
    }
    const uint32_t b_kstep = a.w_tiled ? 64u * BK * 2u : BK * 2u;       // bytes from one K step of a weight row to the next

    const int KTall = a.Ktot / BK;
    const int kbeg = (int)((int64_t)kpart * KTall / kparts), kend = (int)((int64_t)(kpart + 1) * KTall / kparts);
    int ir = 0, is = 0, ic0 = 0;       // filter tap / channel offset of the next tile to issue
    if (kbeg > 0) {
        const int per_tap = a.C / BK, tap = kbeg / per_tap;
        ic0 = (kbeg - tap * per_tap) * BK;
        ir = tap / a.S;
        is = tap - ir * a.S;
    }
    auto issue = [&](int kt, int stage) {
        T* sbase = smem + stage * STAGE;
        const uint32_t tapbit = 1u << (ir * a.S + is);
        const int soff = (MODE == 0 ? ((ir * a.dil) * a.Wi + is * a.dil) * a.ldx
                                    : -((((ir * a.dil) >> sh2) * a.Wi + ((is * a.dil) >> sh2)) * a.ldx)) * 2 + ic0 * 2;
#pragma unroll
        for (int jj = 0; jj < A_I; ++jj) {
            const uint32_t voff = (a_mask[jj] & tapbit) ? (uint32_t)(a_base[jj] + soff) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_ptr_t)(sbase + (wave * A_I + jj) * 16 * BK), 16, voff, 0, 0, 0);
        }
#pragma unroll
        for (int jj = 0; jj < B_I; ++jj) {    // OOB + K offset stays past the descriptor's range (tensors < 2^31 bytes)
            const uint32_t voff = b_off[jj] + (uint32_t)kt * b_kstep;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_ptr_t)(sbase + BM * BK + (wave * B_I + jj) * 16 * BK), 16, voff, 0, 0, 0);
        }
        ic0 += BK;
        if (ic0 >= a.C) {
            ic0 = 0;
            if (++is == a.S) { is = 0; ++ir; }
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    const int KT = kend - kbeg;
#pragma unroll
    for (int t = 0; t < LA; ++t)
        if (t < KT) issue(kbeg + t, t);

    const int lr = lane & 15, lq = lane >> 4;
    uint64_t ph_t[5] = {0, 0, 0, 0, 0}, ph_sum[4] = {0, 0, 0, 0}, ph_begin = 0;
    auto stamp = [&](int i) {
        if constexpr (PH) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_t[i])::"memory");
    };
    if constexpr (PH) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_begin)::"memory");
    for (int kt = 0; kt < KT; ++kt) {
        stamp(0);
        // tiles kt .. min(kt + LA - 1, KT - 1) are in flight; tile kt must have landed
        if (LA > 2 && kt + 2 < KT) wait_vmcnt<(LA > 2 ? 2 : 0) * NI>();
        else if (LA > 1 && kt + 1 < KT) wait_vmcnt<(LA > 1 ? 1 : 0) * NI>();
        else wait_vmcnt<0>();
        stamp(1);
        __builtin_amdgcn_s_barrier();
        stamp(2);
        if (kt + LA < KT) issue(kbeg + kt + LA, (kt + LA) % NST);
        stamp(3);
        const T* as = smem + (kt % NST) * STAGE + (wm * TM) * BK;
        const T* bs = smem + (kt % NST) * STAGE + BM * BK + (wn * TN) * BK;
        mfma_bf16x8 bf[NT], af[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
            bf[i] = *reinterpret_cast<const mfma_bf16x8*>(bs + b_row<NT>(i, lr) * BK + swz_chunk<T>(lr, lq) * 8);
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int row = j * 16 + lr;
            af[j] = *reinterpret_cast<const mfma_bf16x8*>(as + row * BK + swz_chunk<T>(row, lq) * 8);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf[i], af[j], acc[i][j], 0, 0, 0);
        if constexpr (PH) {
            stamp(4);          // (after the MFMAs were ISSUED: the matrix pipe may still be working)
#pragma unroll
            for (int q = 0; q < 4; ++q) ph_sum[q] += ph_t[q + 1] - ph_t[q];
        }
    }
    if constexpr (PH) {
        uint64_t ph_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ph_end)::"memory");
        if (a.dbg != nullptr && lane == 0) {
            float* o = a.dbg + ((int64_t)blockIdx.x * WAVES + wave) * 8;
            const float inv = 1.0f / (float)(KT > 0 ? KT : 1);
            o[0] = (float)ph_sum[0] * inv;      // vmcnt wait
            o[1] = (float)ph_sum[1] * inv;      // barrier
            o[2] = (float)ph_sum[2] * inv;      // DMA issue
            o[3] = (float)ph_sum[3] * inv;      // fragment reads + MFMA issue
            o[4] = (float)(ph_end - ph_begin) * inv;      // whole loop per step (includes the stamps themselves)
            o[5] = (float)KT;
            o[6] = 0.f;
            o[7] = 0.f;
        }
    }
    // Cut the accumulators' live ranges here: without it hipcc keeps the MFMA results un-tied through the
    // branchy epilogue and re-copies all 64 AGPRs (v_accvgpr_mov + s_nop) in EVERY K step (-35 % throughput).
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
    if (kparts > 1) {
        // K-split remainder tile: park this part's fp32 accumulators (lane-linear: one 4 KB row per (i, j)), publish,
        // and let the part that arrives last add all parts in part order -- deterministic -- and finish the tile.
        const int rt = tile - a.tail_full;
        // write-through (sc1) slab stores and sc1 loads instead of release / acquire fences: a release fence writes back
        // every dirty line of this XCD's L2 -- the other workgroups' output tiles included (first version: -2.3 % on the
        // whole step).  cdna_hip_programming.md, split-K seam: sc1 stores -> drained -> barrier -> relaxed ticket; the
        // last arriver reads with sc1 loads.
        typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
        const __amdgpu_buffer_rsrc_t rs_ws = __builtin_amdgcn_make_buffer_rsrc(
            a.tail_ws + (int64_t)rt * kparts * (BM * BN), 0, kparts * BM * BN * 4, 0x00020000);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, acc[i][j]), rs_ws,
                                                       (kpart * BM * BN + ((i * MT + j) * NTH + tid) * 4) * 4, 0, 16);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        int* flag = reinterpret_cast<int*>(smem);            // the operand ring is free now
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(a.tail_cnt + rt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == kparts - 1)
                __hip_atomic_store(a.tail_cnt + rt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // ready for the next launch
            *flag = ticket;
        }
        __syncthreads();
        if (*flag != kparts - 1) return;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < kparts; ++p) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(
                        rs_ws, (p * BM * BN + ((i * MT + j) * NTH + tid) * 4) * 4, 0, 16);
                    acc[i][j] += __builtin_bit_cast(f32x4, v);
                }
        }
    }
    if constexpr (MT == 4) {
        conv_epilogue<T, NT, MT, MODE>(acc, a, m0 + wm * TM, n0 + wn * TN, lr, lq);
    } else {
        // 128-row wave tile: the epilogue's statistics / BN-backward partials are per DML_STAT_ROWS = 64 rows, so it runs
        // once per 64-row half (which also halves its register footprint)
#pragma unroll
        for (int h = 0; h < MT / 4; ++h) {
            f32x4 sub[NT][4];
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) sub[i][j] = acc[i][h * 4 + j];
            conv_epilogue<T, NT, 4, MODE, false>(sub, a, m0 + wm * TM + h * 64, n0 + wn * TN, lr, lq);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// bf16 forward / data-gradient kernel, WAVE-SPECIALISED version (round 4; tools/probe_ws_gemm.hip is its GEMM-only probe,
// profiles/r04_ws_probe_*.txt the measurements behind every choice below).
// What bounded the ring kernel above: every wave both issues its share of the K step's buffer_load ... lds instructions and
// consumes the step -- a wave issues in order, so the 84-170 cycles the texture addresser takes to accept each of its four
// DMA instructions are cycles in which it issues no MFMA, and three co-resident workgroups only partly fill the holes
// (profiles/r03_dma_phases.txt).  Here the two jobs belong to different waves of one workgroup per CU:
//   * NLD LOADER waves own the vector-memory path.  Loader l issues pieces l, l + NLD, ... (1 KB = 16 tile rows x 64 B each)
//     of every K step into an NST-stage LDS ring, keeps D stages in flight behind the one it publishes (counted vmcnt) and
//     announces "stage landed" through a counter in LDS.  A single wave gets one DMA instruction accepted per ~100 cycles,
//     the addresser takes one per ~30 from several waves: three to four loaders saturate it (1 loader 455, 2: 906, 3: 997
//     TFLOP/s on layer3's 3x3).  The DMA instruction is inline asm: hipcc must not see an LDS-DMA in flight, or it drains
//     vmcnt(0) before each of the loader's own LDS accesses (the flag polls).
//   * 4 CONSUMER waves (one per SIMD) never touch vector memory inside the K loop: poll the counter (the poll for step
//     k + 1 is issued in the middle of step k's MFMAs), read fragments, issue MFMAs back to back.  The fragments of step
//     k + 1 are fetched under the MFMAs of step k, register by register as they become free (A fragment j right after the
//     four MFMAs that used it; the weight fragments and the last A fragment alternate between two register sets).
//   * No s_barrier in the K loop; the ring decouples the two sides.  WAR: a consumer announces "every read of stage g has
//     been ISSUED" (DS executes one wave's operations in order, so the flag write lands after them); a loader overwrites a
//     stage only when all four consumers have announced it.
//   * Tile = (144 MW) x (64 NW) pixels x channels on 144 x 64 WAVE tiles (36 MFMAs per wave and K step against 13 KB of
//     fragment reads; the 64 x 64 wave tile of the ring kernel: 16 against 8 KB).  144 rows because the maps of this network
//     at 16 images are 36 864 / 147 456 / 589 824 pixels: 256 CUs x 144 rows x {1, 4, 16} -- one 144 x 256 tile per CU fills the
//     chip exactly where 128-row tiles leave a quarter of a round (layer3: 576 tiles on 768 slots) and 256 x 256 tiles 44 %
//     of it.  One workgroup per CU; workgroups are persistent and walk the tile list (m fastest: the CUs work on the same
//     weight rows at the same time), the loaders run ahead into the next tile while the consumers store the finished one.
//   * BatchNorm partial statistics / BN-backward partial sums come per 48 rows (three groups per wave tile): the epilogue is
//     conv_epilogue on 48-row sub-tiles, and the finalize kernels take the group height (dml_conv_stat_rows).
// Weights must be the tile-major copy (w_tiled).  MODE as conv_igemm_dma_kernel.
// ------------------------------------------------------------------------------------------------
// fp32 output of a 48 x 64 sub-tile as WHOLE 256-byte rows: straight from the accumulators a store instruction writes, per pixel
// row, four 16-byte pieces 32 bytes apart (the lane's channel runs, frag_chan) -- 16 rows x 4 pieces per instruction, and the
// chip takes a 144 x 256 fp32 tile per CU (37.7 MB per launch) in 14-16 us, every K loop waiting behind it.  Staged through
// 12 KB of LDS (row-major, 16-byte chunk index XOR row: conflict-free both ways) every store instruction -- and every read
// of the accumulate / residual operands -- covers four pixel rows x 256 contiguous bytes: 5-6 us (profiles/r04_h2_epilogue.txt).
// BatchNorm statistics come from the accumulators as before (conv_epilogue, STATS_ONLY); accumulate and the stores happen on the
// row side (launches with a bias or the inference epilogue post_* keep conv_epilogue).  `stage`: this wave's own 12 KB.
// BatchNorm partial statistics of a WHOLE 144 x 64 wave tile of the two-plane forward kernel: (sum, M2 about the tile mean) per channel over
// its 144 rows, from the accumulators, before the row epilogue stores them sub-tile by sub-tile.  Round 6: with 48-row groups the
// cross-lane reductions (a DPP row sum per channel and sub-tile, twice) made the statistics 15-28 % of the short-K 1x1 forward
// launches (tools/bench_h2.py with and without `stats`: 1x1 256 -> 1024 at 48 x 48 94.9 vs 80.1 us, 64 -> 256 at 192 x 192 177.6 vs
// 128.4); per wave tile they are a third as many, and the finalize kernels fold a third of the partial rows.
// acc3[h][i][jj] = rows mw0 + 48 h + 16 jj + (lane & 15), channels nw0 + frag_chan<4>(i, lane >> 4) .. + 3
template <int NT, int NS>
__device__ __forceinline__ void ws_tile_stats(const f32x4 (&acc3)[NS][NT][3], const ConvArgs& a, const int mw0, const int nw0,
                                              const int lr, const int lq) {
    static_assert(NT == 4, "64-channel wave tile");
    constexpr int TM = NS * 48;
    const int cnt = min(TM, max(0, a.M - mw0));
    if (cnt <= 0) return;
    const float inv = 1.0f / (float)cnt;
    float* sg = a.stats + (int64_t)(mw0 / TM) * a.N * 2;
    float S[NT][4], Q[NT][4];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NS; ++h)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const bool v = (mw0 + h * 48 + jj * 16 + lr) < a.M;
#pragma unroll
                for (int q = 0; q < 4; ++q) s[q] += v ? acc3[h][i][jj][q] : 0.f;
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = row16_sum(s[q]);
        float m2[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int h = 0; h < NS; ++h)
#pragma unroll
            for (int jj = 0; jj < 3; ++jj) {
                const bool v = (mw0 + h * 48 + jj * 16 + lr) < a.M;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float dlt = acc3[h][i][jj][q] - s[q] * inv;
                    m2[q] += v ? dlt * dlt : 0.f;
                }
            }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            S[i][q] = s[q];
            Q[i][q] = row16_sum(m2[q]);
        }
    }
    // one store instruction per wave: lane (lq, lr = 2 i + h) writes the (sum, M2) pairs of channels 2 h, 2 h + 1 of tile i (conv_epilogue)
    if (lr < 2 * NT && nw0 + NT * 16 <= a.N) {
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                if (lr == 2 * i + h) { v0 = S[i][2 * h]; v1 = Q[i][2 * h]; v2 = S[i][2 * h + 1]; v3 = Q[i][2 * h + 1]; }
        const int n = nw0 + frag_chan<NT>(lr >> 1, lq) + (lr & 1) * 2;
        st16f(sg + (int64_t)n * 2, v0, v1, v2, v3, (a.nt_out & 2) != 0);
    }
}

constexpr int WS_STAT_ROWS_C = 48;             // (= WS_STAT_ROWS, declared below)
template <int NT, int MODE, bool OPS = true>      // OPS = false: launches with epilogue operands go to conv_epilogue_rows_ops
__device__ __forceinline__ void conv_epilogue_rows(f32x4 (&acc)[NT][3], const ConvArgs& a, const int mw0, const int nw0,
                                                   const int lane, char* stage, char* mstage, const bool do_stats = true) {
    static_assert(NT == 4, "64-channel wave tile");
    const int lr = lane & 15, lq = lane >> 4;
    if (MODE == 0 && a.stats != nullptr && do_stats) conv_epilogue<float, NT, 3, MODE, false, true>(acc, a, mw0, nw0, lr, lq);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int row = j * 16 + lr, chunk = frag_chan<NT>(i, lq) >> 2;
            *reinterpret_cast<f32x4*>(stage + row * 256 + ((chunk ^ (row & 15)) << 4)) = acc[i][j];
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own writes, read back by other lanes
    const int c0 = nw0 + (lane & 15) * 4;
    if (c0 >= a.N) return;                                       // (a last, half-empty 128-wide block)
    // the accumulate operand: the gradient's earlier value (accum), or the identity branch's share of it -- the block output's
    // gradient res_dz under its ReLU mask (DmlConvDesc.res_*, one mask byte per four channels)
    const bool resm = OPS && MODE == 1 && a.res_dz != nullptr;
    const bool has_old = OPS && (a.accum != 0 || resm);
    const bool bnr = OPS && MODE == 1 && a.bnr_partials != nullptr;
    float* const yb = static_cast<float*>(a.y);
    if (!OPS || (!has_old && !bnr)) {
        // all twelve LDS reads first (into the registers the sub-tile's accumulators just left), then twelve stores back to back.
        // (One path with the accumulate operand selected per element made every store wait vmcnt(0) -- stores count there too on
        // gfx950 -- i.e. for the store before it: 36 serialised round trips per tile, as slow as the scattered stores this function
        // replaces; hence separate code paths on wave-uniform branches.)
        float4 o[12];
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int r = k * 4 + (lane >> 4);
            o[k] = *reinterpret_cast<const float4*>(stage + r * 256 + (((lane & 15) ^ (r & 15)) << 4));
        }
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const int m = mw0 + k * 4 + (lane >> 4);
            if (m < a.M) st16f(yb + (int64_t)(MODE == 1 ? ws_out_row(a, (uint32_t)m) : (uint32_t)m) * a.ldy + c0, o[k].x, o[k].y, o[k].z, o[k].w,
                               (a.nt_out & 1) != 0);
        }
        return;
    }
    // With an accumulate operand (the gradient has earlier producers) and / or the BatchNorm-backward sums of the tensor being
    // written (DmlConvDesc.bnr_*: sum g, sum g * xhat per 48 rows and channel, g = dz * [ReLU mask bit] -- what dml_bn_bwd_reduce
    // would compute from the stored tensor, without its pass over dz and y): groups of four rows, the operand loads of the NEXT
    // group issued before this group's stores, so that waiting for them does not wait for the stores.
    auto body = [&](auto ho, auto bn) {
        constexpr bool HO = decltype(ho)::value, BNR = decltype(bn)::value;
        float4 old[2][4], yv[2][4];
        float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1, mu = r1, is = r1;
        uint32_t gmx = 0;
        if (BNR) {
            mu = *reinterpret_cast<const float4*>(a.bnr_mean + c0);
            is = *reinterpret_cast<const float4*>(a.bnr_invstd + c0);
        }
        // ReLU masks of the sub-tile (one byte per four channels, low nibble): lane L < 48 fetches row L's 16 mask bytes -- this wave's
        // 64 channels -- of each mask with ONE 16-byte load, packs them (identity-branch mask: low nibble, BatchNorm mask: high
        // nibble) and parks the 768 bytes in the wave's mask area behind the ring; the row loop reads its byte back.  Per-lane byte
        // loads were one VMEM instruction per mask and four rows -- 24 of the ~60 of a sub-tile -- and the texture addresser pays
        // per instruction, not per byte.
        const bool use_rm = HO && resm, use_bm = BNR && a.bnr_relu != 0;
        // byte of row (4 c + (lane >> 4)), chunk (lane & 15) = mstage[64 c + lane]: one base, compile-time offsets (computed here, from
        // a lane index the optimiser cannot see through -- hoisted to the head of the kernel the twelve row offsets get spilled, and a
        // scratch reload inside the row loop waits vmcnt(0), i.e. for every store in flight)
        int lane_m = lane;
        asm volatile("" : "+v"(lane_m));
        const char* const mrd = mstage + lane_m;
        if (use_rm || use_bm) {
            if (lane < WS_STAT_ROWS_C) {
                const int m = mw0 + lane;
                uint4 rv = make_uint4(0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu, 0x0f0f0f0fu), bv = rv;
                if (m < a.M) {
                    const int64_t mo = (int64_t)ws_out_row(a, (uint32_t)m) * (a.N >> 2) + (nw0 >> 2);
                    if (use_rm) rv = *reinterpret_cast<const uint4*>(a.res_mask + mo);
                    if (use_bm) bv = *reinterpret_cast<const uint4*>(a.bnr_mask + mo);
                }
                uint4 pk;
                pk.x = (rv.x & 0x0f0f0f0fu) | ((bv.x & 0x0f0f0f0fu) << 4);
                pk.y = (rv.y & 0x0f0f0f0fu) | ((bv.y & 0x0f0f0f0fu) << 4);
                pk.z = (rv.z & 0x0f0f0f0fu) | ((bv.z & 0x0f0f0f0fu) << 4);
                pk.w = (rv.w & 0x0f0f0f0fu) | ((bv.w & 0x0f0f0f0fu) << 4);
                *reinterpret_cast<uint4*>(mstage + lane * 16) = pk;
            }
        }
        auto load_ops = [&](const int g4) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int m = mw0 + (g4 * 4 + u) * 4 + (lane >> 4);
                old[g4 & 1][u] = make_float4(0.f, 0.f, 0.f, 0.f);
                yv[g4 & 1][u] = old[g4 & 1][u];
                if (m < a.M) {
                    const int64_t pm = (int64_t)ws_out_row(a, (uint32_t)m);
                    if (HO) {
                        if (resm) old[g4 & 1][u] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.res_dz) + pm * a.res_ld + c0);
                        else old[g4 & 1][u] = *reinterpret_cast<const float4*>(yb + pm * a.ldy + c0);
                    }
                    if (BNR) yv[g4 & 1][u] = *reinterpret_cast<const float4*>(static_cast<const float*>(a.bnr_y) + pm * a.bnr_ldy + c0);
                }
            }
        };
        load_ops(0);
#pragma unroll
        for (int g4 = 0; g4 < 3; ++g4) {
            if (g4 + 1 < 3) load_ops(g4 + 1);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int r = (g4 * 4 + u) * 4 + (lane >> 4), m = mw0 + r;
                if (m >= a.M) continue;
                float4 o = *reinterpret_cast<const float4*>(stage + r * 256 + (((lane & 15) ^ (r & 15)) << 4));
                uint32_t pkb = 0xffu;
                if (use_rm || use_bm) pkb = *reinterpret_cast<const uint8_t*>(mrd + (g4 * 4 + u) * 64);
                if (HO) {
                    float4 t = old[g4 & 1][u];
                    if (resm) {
                        t.x = (pkb & 1u) ? t.x : 0.f; t.y = (pkb & 2u) ? t.y : 0.f; t.z = (pkb & 4u) ? t.z : 0.f; t.w = (pkb & 8u) ? t.w : 0.f;
                    }
                    o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
                }
                st16f(yb + (int64_t)ws_out_row(a, (uint32_t)m) * a.ldy + c0, o.x, o.y, o.z, o.w, (a.nt_out & 1) != 0);
                if (BNR) {
                    const uint32_t bits = pkb >> 4;
                    const float4 y4 = yv[g4 & 1][u];
                    const float g0 = (bits & 1u) ? o.x : 0.f, g1 = (bits & 2u) ? o.y : 0.f, g2 = (bits & 4u) ? o.z : 0.f,
                                g3 = (bits & 8u) ? o.w : 0.f;
                    r1.x += g0; r1.y += g1; r1.z += g2; r1.w += g3;
                    r2.x += g0 * (y4.x - mu.x) * is.x; r2.y += g1 * (y4.y - mu.y) * is.y;
                    r2.z += g2 * (y4.z - mu.z) * is.z; r2.w += g3 * (y4.w - mu.w) * is.w;
                    gmx = max(max(gmx, __float_as_uint(g0) & 0x7fffffffu), max(__float_as_uint(g1) & 0x7fffffffu,
                              max(__float_as_uint(g2) & 0x7fffffffu, __float_as_uint(g3) & 0x7fffffffu)));
                }
            }
        }
        if (BNR) {
            // the four lane groups (lane >> 4) hold the same four channels: fold, then lanes 0 .. 15 write their channels' pairs
            float v[8] = {r1.x, r2.x, r1.y, r2.y, r1.z, r2.z, r1.w, r2.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += __shfl_xor(v[e], 16, 64);
                v[e] += __shfl_xor(v[e], 32, 64);
            }
            if (lane < 16 && mw0 < a.M) {
                float* pp = a.bnr_partials + ((int64_t)(mw0 / WS_STAT_ROWS_C) * a.N + c0) * 2;
                st16f(pp, v[0], v[1], v[2], v[3], false);
                st16f(pp + 4, v[4], v[5], v[6], v[7], false);
            }
            if (a.bnr_gmax != nullptr) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) gmx = max(gmx, (uint32_t)__shfl_xor((int)gmx, o, 64));
                if (lane == 0 && gmx != 0)
                    atomicMax(reinterpret_cast<uint32_t*>(a.bnr_gmax) + ((blockIdx.x * 16u + (uint32_t)(mw0 / WS_STAT_ROWS_C) + (uint32_t)(nw0 >> 6)) & 1023u), gmx);
            }
        }
    };
    if constexpr (OPS) {
        if (bnr) {
            if (has_old) body(std::true_type{}, std::true_type{});
            else body(std::false_type{}, std::true_type{});
        } else {
            body(std::true_type{}, std::false_type{});
        }
    }
}

// conv_epilogue_rows for a WHOLE wave tile (NS 48-row sub-tiles) of a data gradient WITH epilogue operands (identity-branch gradient /
// accumulate, fused BatchNorm-backward sums): one software pipeline over all 12 NS row quads instead of one per sub-tile.  The operand
// loads of a launch like the data gradient of 1x1 1024 -> 256 (dz and y in, dx out: 490 MB) ran at 4 TB/s = the bytes a CU had in
// flight (two groups of four row quads per wave) over the loaded HBM latency, and every sub-tile began with an exposed round trip
// (its masks and its first group).  Here every row quad has its own operand registers (a ring of 12, indices compile-time in the
// unrolled quad loop), the loads run WS_EPI_P quads ahead ACROSS the sub-tile boundaries of the rolled sub-tile loop, and the masks
// of sub-tile h + 1 are fetched during sub-tile h.
#ifndef DML_WS_EPI_P
#define DML_WS_EPI_P 8
#endif
#ifndef DML_WS_EPI_OPS
#define DML_WS_EPI_OPS 1                       // 0: the per-sub-tile pipeline of conv_epilogue_rows for launches with operands too (A/B)
#endif
template <int I, int N, class F>
__device__ __forceinline__ void ws_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        ws_static_for<I + 1, N>(f);
    }
}
template <int NT, int NS, bool HO, bool BNR>
__device__ __forceinline__ void conv_epilogue_rows_ops(f32x4 (&acc3)[NS][NT][3], const ConvArgs& a, const int mwt, const int nw0,
                                                       const int lane_in, char* stage0, char* stage1, char* mstage) {
    static_assert(NT == 4, "64-channel wave tile");
    constexpr int P = DML_WS_EPI_P;
    static_assert(P >= 1 && P <= 11, "ring of 12 row quads");
    typedef unsigned int u32x4_b __attribute__((ext_vector_type(4)));
    constexpr uint32_t OOB = 0x80000000u;
    // (a lane index the optimiser cannot see through: hoisted out of the tile loop, the per-quad addresses below get spilled)
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    // Every global access of the quad loop is a BUFFER access with the tensor's byte size as range: rows beyond M read zeros and
    // their stores are dropped by the address unit, an access that must not happen gets an offset 2^31 bytes further (all these
    // tensors are below 2^31 bytes, conv_ws_planes_eligible).  No branch around any memory instruction, so the loop is straight-line
    // code whose vmcnt waits the compiler counts exactly; under `if (m < M)` / `if (h + 1 < NS)` branches it must assume the
    // fewest instructions behind a load and ends up draining the queue.
    const int c0 = nw0 + (lane & 15) * 4;
    const uint32_t coff = c0 < a.N ? (uint32_t)c0 * 4u : OOB;
    const bool resm = a.res_dz != nullptr;
    const uint32_t y_row = (uint32_t)a.ldy * 4u;
    const uint32_t o_row = resm ? (uint32_t)a.res_ld * 4u : y_row, b_row = (uint32_t)a.bnr_ldy * 4u, m_row = (uint32_t)(a.N >> 2);
    const bool use_rm = HO && resm, use_bm = BNR && a.bnr_relu != 0;
    // (sub_grid: the launch's rows are a parity class of tensors with four times as many pixels, ws_out_row; a row beyond M maps
    // beyond every range)
    const uint32_t prow_n = ws_out_rows(a);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, (int)(prow_n * y_row), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
        resm ? const_cast<void*>(a.res_dz) : a.y, 0, HO ? (int)(prow_n * o_row) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_b = __builtin_amdgcn_make_buffer_rsrc(
        BNR ? const_cast<void*>(static_cast<const void*>(a.bnr_y)) : a.y, 0, BNR ? (int)(prow_n * b_row) : 0, 0x00020000);
    // (a mask that is not in use: a zero-sized range -- the loads return zeros without touching memory -- and constant bits instead)
    const __amdgpu_buffer_rsrc_t rs_rm = __builtin_amdgcn_make_buffer_rsrc(
        use_rm ? const_cast<void*>(static_cast<const void*>(a.res_mask)) : a.y, 0, use_rm ? (int)(prow_n * m_row) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_bm = __builtin_amdgcn_make_buffer_rsrc(
        use_bm ? const_cast<void*>(static_cast<const void*>(a.bnr_mask)) : a.y, 0, use_bm ? (int)(prow_n * m_row) : 0, 0x00020000);
    const int ngroups = (a.M + WS_STAT_ROWS_C - 1) / WS_STAT_ROWS_C;
    const __amdgpu_buffer_rsrc_t rs_p = __builtin_amdgcn_make_buffer_rsrc(
        BNR ? static_cast<void*>(a.bnr_partials) : a.y, 0, BNR ? (int)((uint32_t)ngroups * (uint32_t)a.N * 8u) : 0, 0x00020000);
    const uint32_t nomask = use_bm ? 0u : 0xfu;
    u32x4_b old[12], yv[12];
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), is = mu;
    if (BNR && c0 < a.N) {
        mu = *reinterpret_cast<const float4*>(a.bnr_mean + c0);
        is = *reinterpret_cast<const float4*>(a.bnr_invstd + c0);
    }
    // ReLU masks of a sub-tile: lane L fetches row L's 16 mask bytes (this wave's 64 channels) of each mask; lanes < 48 park them,
    // packed (identity-branch mask: low nibble, BatchNorm mask: high nibble), in the wave's mask area (as conv_epilogue_rows);
    // the masks of sub-tile h + 1 are fetched during sub-tile h
    u32x4_b rv, bv;
    auto load_masks = [&](const int h) {
        const uint32_t mo = ws_out_row(a, (uint32_t)(mwt + h * WS_STAT_ROWS_C + lane)) * m_row + (uint32_t)(nw0 >> 2);
        rv = __builtin_amdgcn_raw_buffer_load_b128(rs_rm, mo, 0, 0);
        bv = __builtin_amdgcn_raw_buffer_load_b128(rs_bm, mo, 0, 0);
    };
    // operands of row quad Q of the wave tile (rows mwt + 4 Q + (lane >> 4)) into its ring slot
    auto load_quad = [&](auto Qc) {
        constexpr int Q = decltype(Qc)::value, slot = Q % 12;
        if constexpr (Q < 12 * NS) {
            const uint32_t m = ws_out_row(a, (uint32_t)(mwt + Q * 4 + (lane >> 4)));
            if (HO) old[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_o, m * o_row + coff, 0, 0);
            if (BNR) yv[slot] = __builtin_amdgcn_raw_buffer_load_b128(rs_b, m * b_row + coff, 0, 0);
        }
    };
    // accumulators -> LDS rows: sub-tiles 0 and 1 at once (two staging areas: the slots of the tile's last TWO K stages), sub-tile 2
    // into area 0 when sub-tile 0 has been read -- 48 accumulator registers live under the operand ring instead of 96
    auto stage_acc = [&](f32x4 (&sub)[NT][3], char* st) {
        const int lr = lane & 15, lq = lane >> 4;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int row = j * 16 + lr, chunk = frag_chan<NT>(i, lq) >> 2;
                *reinterpret_cast<f32x4*>(st + row * 256 + ((chunk ^ (row & 15)) << 4)) = sub[i][j];
            }
    };
    stage_acc(acc3[0], stage0);
    if constexpr (NS == 3) stage_acc(acc3[1], stage1);
    __builtin_amdgcn_sched_barrier(0);      // (the ring's first loads after the staging: 96 accumulator registers are free by then)
    load_masks(0);
    ws_static_for<0, P>(load_quad);
    uint32_t gmx = 0;
    // (the sub-tile loop unrolled: rolled, the ring slots still in flight at its back edge get copied from register to register there,
    // and every copy waits for its load -- the queue drained once per sub-tile)
    ws_static_for<0, NS>([&](auto hc) {
        constexpr int h = decltype(hc)::value;
        asm volatile("" : "+v"(lane));      // (per-quad LDS offsets recomputed per sub-tile, not kept live across sub-tiles)
        const int mw0 = mwt + h * WS_STAT_ROWS_C;
        const char* const stage = (h & 1) ? stage1 : stage0;
        {
            u32x4_b pk;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk[e] = (rv[e] & 0x0f0f0f0fu) | ((bv[e] & 0x0f0f0f0fu) << 4);
            if (lane < WS_STAT_ROWS_C) *reinterpret_cast<u32x4_b*>(mstage + lane * 16) = pk;
            if constexpr (h + 1 < NS) load_masks(h + 1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the wave's own writes, read back by other lanes
        float4 r1 = make_float4(0.f, 0.f, 0.f, 0.f), r2 = r1;
        auto quad = [&](auto qc) {
            constexpr int q = decltype(qc)::value;
            __builtin_amdgcn_sched_barrier(0);      // quads in program order: the ring slots are what bounds the registers
            // the loads that run P quads ahead (into the slot quad q + P - 12 left P quads ago)
            load_quad(std::integral_constant<int, h * 12 + q + P>{});
            const int r = q * 4 + (lane >> 4);
            const uint32_t m = ws_out_row(a, (uint32_t)(mw0 + r));
            float4 o = *reinterpret_cast<const float4*>(stage + r * 256 + (((lane & 15) ^ (r & 15)) << 4));
            const uint32_t pkb = *reinterpret_cast<const uint8_t*>(mstage + lane + q * 64);
            const float4 inc = o;          // (the convolution's own share of the gradient: DmlConvDesc.bnr_inc)
            if (HO) {
                float4 t = make_float4(__uint_as_float(old[q][0]), __uint_as_float(old[q][1]), __uint_as_float(old[q][2]),
                                       __uint_as_float(old[q][3]));
                if (resm) {
                    t.x = (pkb & 1u) ? t.x : 0.f; t.y = (pkb & 2u) ? t.y : 0.f; t.z = (pkb & 4u) ? t.z : 0.f; t.w = (pkb & 8u) ? t.w : 0.f;
                }
                o.x += t.x; o.y += t.y; o.z += t.z; o.w += t.w;
            }
            const u32x4_b ov = {__float_as_uint(o.x), __float_as_uint(o.y), __float_as_uint(o.z), __float_as_uint(o.w)};
            // (non-temporal whatever nt_out says: one store instruction on every path -- data gradients with operands write 38-600 MB)
            __builtin_amdgcn_raw_buffer_store_b128(ov, rs_y, m * y_row + coff, 0, 2);
            if (BNR) {
                // (rows beyond M: their accumulators are zero -- the loaders fill such rows with zeros -- and so are the operands)
                const uint32_t bits = (pkb >> 4) | nomask;
                const float4 y4 = make_float4(__uint_as_float(yv[q][0]), __uint_as_float(yv[q][1]), __uint_as_float(yv[q][2]),
                                              __uint_as_float(yv[q][3]));
                const float g0 = (bits & 1u) ? o.x : 0.f, g1 = (bits & 2u) ? o.y : 0.f, g2 = (bits & 4u) ? o.z : 0.f,
                            g3 = (bits & 8u) ? o.w : 0.f;
                // (the sums over this launch's own share where the caller says so -- only an accumulating launch has another share)
                const bool use_inc = HO && a.bnr_inc != 0;
                const float s0 = use_inc ? ((bits & 1u) ? inc.x : 0.f) : g0, s1 = use_inc ? ((bits & 2u) ? inc.y : 0.f) : g1,
                            s2 = use_inc ? ((bits & 4u) ? inc.z : 0.f) : g2, s3 = use_inc ? ((bits & 8u) ? inc.w : 0.f) : g3;
                r1.x += s0; r1.y += s1; r1.z += s2; r1.w += s3;
                r2.x += s0 * (y4.x - mu.x) * is.x; r2.y += s1 * (y4.y - mu.y) * is.y;
                r2.z += s2 * (y4.z - mu.z) * is.z; r2.w += s3 * (y4.w - mu.w) * is.w;
                gmx = max(max(gmx, __float_as_uint(g0) & 0x7fffffffu), max(__float_as_uint(g1) & 0x7fffffffu,
                          max(__float_as_uint(g2) & 0x7fffffffu, __float_as_uint(g3) & 0x7fffffffu)));
                asm volatile("" : "+v"(gmx), "+v"(r1.x), "+v"(r2.x));      // (here, not at the end of the sub-tile with 48 values kept live for it)
            }
        };
        ws_static_for<0, 12>(quad);
        __builtin_amdgcn_sched_barrier(0);
        if (BNR) {
            // the four lane groups (lane >> 4) hold the same four channels: fold, then lanes 0 .. 15 write their channels' pairs
            float v[8] = {r1.x, r2.x, r1.y, r2.y, r1.z, r2.z, r1.w, r2.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                v[e] += __shfl_xor(v[e], 16, 64);
                v[e] += __shfl_xor(v[e], 32, 64);
            }
            const uint32_t po = (lane < 16 && mw0 < a.M && c0 < a.N) ? ((uint32_t)(mw0 / WS_STAT_ROWS_C) * (uint32_t)a.N + (uint32_t)c0) * 8u : OOB;
            const u32x4_b p0 = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
            const u32x4_b p1 = {__float_as_uint(v[4]), __float_as_uint(v[5]), __float_as_uint(v[6]), __float_as_uint(v[7])};
            __builtin_amdgcn_raw_buffer_store_b128(p0, rs_p, po, 0, 0);
            __builtin_amdgcn_raw_buffer_store_b128(p1, rs_p, po + 16u, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // this sub-tile's reads of its staging area and of the masks
        static_assert(NS == 1 || NS == 3, "one or three sub-tiles per wave tile");
        if constexpr (NS == 3 && h == 0) stage_acc(acc3[2], stage0);
    });
    if (BNR && a.bnr_gmax != nullptr) {      // max |g| of the wave tile, for the plane scale of dy (dml_h2_bound_bn_bwd)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmx = max(gmx, (uint32_t)__shfl_xor((int)gmx, o, 64));
        if (lane == 0 && gmx != 0)
            atomicMax(reinterpret_cast<uint32_t*>(a.bnr_gmax) + ((blockIdx.x * 16u + (uint32_t)(mwt / WS_STAT_ROWS_C) + (uint32_t)(nw0 >> 6)) & 1023u), gmx);
    }
}

// The same through 2 KB of LDS that belong to the wave alone (the 144 x 256 configuration leaves 10 KB beside its ring): chunks of
// 16 rows x 32 channels, a store instruction = 8 rows x 128 contiguous bytes (whole cache lines).  No consumer barrier and no held-back
// ring slot: with the staging area inside the ring (conv_epilogue_rows) the loaders prefetch one stage less across the tile boundary
// and the consumers wait for each other -- 8-10 % of a launch with four tiles of eight K steps per CU (profiles/r04_h2_ablations.txt).
// DS operations of one wave execute in order: chunk q + 1 is written right behind the reads of chunk q, whose data the stores then
// wait for with the newer writes still outstanding.  Forward launches only: a version with the data gradients' epilogue operands
// (accumulate, fused BN-backward sums) had four rows of operand loads in flight per lane against the eight of conv_epilogue_rows, and
// those epilogues are bound by exactly that (+2 % on the step's data gradients); both versions in one kernel spill 300 bytes per lane.
template <int NT, int MODE>
__device__ __forceinline__ void conv_epilogue_rows8(f32x4 (&acc)[NT][3], const ConvArgs& a, const int mw0, const int nw0,
                                                    const int lane, char* stage, const bool do_stats = true) {
    static_assert(NT == 4 && MODE == 0, "64-channel wave tile, forward launches");
    const int lr = lane & 15, lq = lane >> 4;
    if (a.stats != nullptr && do_stats) conv_epilogue<float, NT, 3, MODE, false, true>(acc, a, mw0, nw0, lr, lq);
    const int rr = lane >> 3, cc = lane & 7;                     // row side: row within 8, 16-byte chunk within the 128-byte half row
    float* const yb = static_cast<float*>(a.y);
    // chunk q = (fragment j = q >> 1, channel half = q & 1): tiles i = 2 half, 2 half + 1 of fragment j
    auto wr = [&](auto qc) {
        constexpr int q = decltype(qc)::value, j = q >> 1, hf = q & 1;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
            *reinterpret_cast<f32x4*>(stage + lr * 128 + (((lq * 2 + ii) ^ (lr & 7)) << 4)) = acc[2 * hf + ii][j];
    };
    auto rd = [&](float4 (&o)[2]) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int r = k * 8 + rr;
            o[k] = *reinterpret_cast<const float4*>(stage + r * 128 + ((cc ^ (r & 7)) << 4));
        }
        asm volatile("" ::: "memory");
    };
    float4 o[2][2];
    auto chunk = [&](auto qc) {
        constexpr int q = decltype(qc)::value, j = q >> 1, hf = q & 1;
        if constexpr (q + 1 < 6) wr(std::integral_constant<int, (q + 1 < 6 ? q + 1 : 0)>{});      // behind the reads of chunk q (in-order DS)
        const int c0 = nw0 + hf * 32 + cc * 4;
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int m = mw0 + j * 16 + k * 8 + rr;
            if (m < a.M && c0 < a.N)
                st16f(yb + (int64_t)m * a.ldy + c0, o[q & 1][k].x, o[q & 1][k].y, o[q & 1][k].z, o[q & 1][k].w, (a.nt_out & 1) != 0);
        }
        if constexpr (q + 1 < 6) rd(o[(q + 1) & 1]);
    };
    wr(std::integral_constant<int, 0>{});
    rd(o[0]);
    chunk(std::integral_constant<int, 0>{});
    chunk(std::integral_constant<int, 1>{});
    chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{});
    chunk(std::integral_constant<int, 4>{});
    chunk(std::integral_constant<int, 5>{});
}

// timing ablations of the two-plane K loop (tools/build_ablations.sh; never defined in the product build): 1 = no fragment reads in
// the loop, 2 = no flag polls / waits, 4 = the loaders issue no DMA, 8 = no epilogue (nothing is stored), 16 = no MFMAs.  Results are
// garbage, durations are what is measured.
#ifndef DML_WS_ABL
#define DML_WS_ABL 0
#endif
#ifndef DML_WS_SPREAD
#define DML_WS_SPREAD 1                        // fragment reads of the two-plane K loop between the MFMA quads (0: in front of them)
#endif
// (accumulators in the accumulator register file through inline-asm MFMAs were tried in round 5: hipcc splits the 256 registers of a
// two-waves-per-SIMD kernel 128 / 128, the 144 accumulators of the 144 x 64 wave tile do not fit, and the K loop did not get faster:
// 888 -> 944 us on the ASPP 3x3, profiles/r05_h2_kloop_ablations.txt)
#if DML_WS_ABL & 16
#define WS_MFMA_F16(ACCV, A_, B_) asm volatile("" ::"v"(A_), "v"(B_))
#else
#define WS_MFMA_F16(ACCV, A_, B_) ACCV = __builtin_amdgcn_mfma_f32_16x16x32_f16(A_, B_, ACCV, 0, 0, 0)
#endif

typedef unsigned int u32x4_ws __attribute__((ext_vector_type(4)));
constexpr int WS_MT = 9;                       // 16-row fragments per wave tile (144 rows)
constexpr int WS_STAT_ROWS = 48;               // rows per statistics group of this kernel
constexpr int WS_NST = 6, WS_D = 2;            // ring stages; stages in flight per loader behind the published one
#ifndef DML_WS_AHEAD
// activation fragments of the two-plane K loop read this many row groups ahead of their use.  2 (three slots, waits at lgkmcnt 3-6
// instead of 1-2) measured +-0 on the step, 76.64 vs 76.56 ms over five interleaved pairs: the latency of the fragment reads is covered
// at one group already; what they cost is issue slots (profiles/r05_h2_kloop_ablations.txt)
#define DML_WS_AHEAD 1
#endif
#ifndef DML_WS_TAP_INNER
#define DML_WS_TAP_INNER 1                     // K order of the two-plane instantiation: channel groups outermost, taps inside
#endif
#ifndef DML_WS_N_FASTEST
#define DML_WS_N_FASTEST 1                     // tile walk of the two-plane instantiation: column blocks fastest
#endif
#ifndef DML_WS_PLANES_NLD
#define DML_WS_PLANES_NLD 3                    // loader waves of the two-plane instantiation
#endif

// LDS-DMA piece (16 B per lane, 1 KB per wave) from inline asm: m0 = LDS byte address of the piece (wave-uniform), voff per
// lane, soff scalar; offsets beyond the descriptor write zeros.  Not counted by hipcc: completion by wait_vmcnt only.
__device__ __forceinline__ void ws_dma16(const u32x4_ws rsrc, const uint32_t lds_addr, const uint32_t voff, const uint32_t soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32x4_ws ws_make_rsrc(const void* base, const uint32_t bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4_ws r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}
__device__ __forceinline__ uint32_t ws_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void ws_st(uint32_t* p, const uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }

// Tile walk.  One plane: row blocks fastest (the workgroups of an XCD share weight rows).  Two planes: column blocks fastest --
// the 32 workgroups of an XCD then hold 32 / nblk_n neighbouring row blocks x all their column blocks AT THE SAME TIME and the
// activation rows are fetched over the fabric once, not once per column block (row blocks fastest: the column blocks of a row
// block run rounds apart; PMC: 1x1 256 -> 1024 at 48 x 48 read its input 4.2 x).
template <int PL>
__device__ __forceinline__ int ws_tile_m(const int tile, const ConvArgs& a) {
    return (PL == 2 && DML_WS_N_FASTEST != 0) ? tile / a.nblk_n : tile % a.nblk_m;
}
template <int PL>
__device__ __forceinline__ int ws_tile_n(const int tile, const ConvArgs& a) {
    return (PL == 2 && DML_WS_N_FASTEST != 0) ? tile % a.nblk_n : tile / a.nblk_m;
}

// ring stages: six in one-plane (bf16) launches, three on two planes (51 KB stages), TWO in the half-tile configuration (two consumer
// waves, 144 x 128, two workgroups per CU: 2 x 34 KB + flags + staging = 74 KB each)
constexpr int ws_ring_stages(const int PL, const int NCW) { return PL == 1 ? WS_NST : (NCW == 2 ? 2 : 3); }

// loader wave LW of NLD: compile-time piece ownership (no branches in the issue loop)
// PL = 2: every operand tile is two planes (hi, lo fp16 of the scaled fp32 tensor); a stage = [A hi | A lo | B hi | B lo]
template <int MW, int NW, int MODE, int NLD, int LW, int PL, int MT = WS_MT>
__device__ __forceinline__ void conv_ws_loader(const ConvArgs& a, const uint32_t x_bytes, const uint32_t w_bytes, char* smem,
                                               uint32_t* ready, uint32_t* consumed, const int lane, const int ntiles,
                                               const int first_tile) {
    typedef bf16_t T;
    constexpr int NCW = MW * NW, NT = 4;
    constexpr int BM = 16 * MT * MW, BN = 64 * NW;
    constexpr int PA = BM / 16, PB = BN / 16, NP = PL * (PA + PB);
    constexpr int SB = PL * (BM + BN) * BK * 2;
    constexpr int MYP = (NP - LW + NLD - 1) / NLD;
    constexpr int NST = ws_ring_stages(PL, NCW), D = PL == 1 ? WS_D : 1;
    constexpr uint32_t OOB = 0x80000000u;
    static_assert(D * MYP <= 63, "vmcnt is a 6-bit counter");
    // (PL = 2: the descriptors span both planes; the lo plane is reached through the scalar offset)
    const u32x4_ws rs_x = ws_make_rsrc(PL == 1 ? a.x : a.x_planes, PL == 1 ? x_bytes : x_bytes + a.x_plane_bytes),
                   rs_w = ws_make_rsrc(PL == 1 ? a.w : a.w_planes, PL == 1 ? w_bytes : w_bytes + a.w_plane_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((0x78 >> (((lane >> 4) & 3) * 2)) & 3);
    const int sh2 = (MODE != 0 && a.stride == 2) ? 1 : 0;
    const int KT = a.Ktot / BK;
    uint32_t g = 0, pub = 0;               // K steps issued / published as landed
    int base[MYP];
    uint32_t mask[MYP];
    int prev_blk_m = -1;
    for (int tile = first_tile; tile < ntiles; tile += (int)gridDim.x) {
        const int blk_m = ws_tile_m<PL>(tile, a), blk_n = ws_tile_n<PL>(tile, a);
        const int m0 = blk_m * BM, n0 = blk_n * BN;
        // per owned piece and lane: A -- signed byte offset of the row's tap-(0,0) pixel (+ swizzled chunk) and one validity
        // bit per filter tap; B -- byte offset into the tile-major weights at K step 0 (see conv_igemm_dma_kernel).  The A part
        // only depends on the row block: a workgroup whose tiles share it (grid = row blocks: N > 256) sets it up once.
        const bool new_rows = blk_m != prev_blk_m;
        prev_blk_m = blk_m;
#pragma unroll
        for (int q = 0; q < MYP; ++q) {
            const int p = q * NLD + LW;             // piece of the stage: [PL x PA activation pieces | PL x PB weight pieces]
            if (p < PL * PA) {
                if (!new_rows) continue;
                base[q] = 0;
                mask[q] = 0;
                const int m = m0 + (p % PA) * 16 + prow;
                if (m < a.M) {
                    const uint32_t b = fdiv((uint32_t)m, a.div_howo);
                    const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
                    const uint32_t yo = fdiv(rem, a.div_wo);
                    const uint32_t xo = rem - yo * (uint32_t)a.Wo;
                    const int iy = MODE == 0 ? (int)yo * a.stride - a.pad : (int)yo + a.pad;
                    const int ix = MODE == 0 ? (int)xo * a.stride - a.pad_x : (int)xo + a.pad_x;
                    const int by = MODE == 0 ? iy : (iy >> sh2), bx = MODE == 0 ? ix : (ix >> sh2);
                    base[q] = ((((int)b * a.Hi + by) * a.Wi + bx) * a.ldx + lchunk * 8) * 2;
                    uint32_t mk = 0;
                    for (int r = 0, t = 0; r < a.R; ++r)
                        for (int s = 0; s < a.S; ++s, ++t) {
                            bool ok;
                            if (MODE == 0) {
                                const int ys = iy + r * a.dil, xs = ix + s * a.dil;
                                ok = ((unsigned)ys < (unsigned)a.Hi) && ((unsigned)xs < (unsigned)a.Wi);
                            } else {
                                const int ty = iy - r * a.dil, tx = ix - s * a.dil;
                                ok = (sh2 == 0 || (((ty | tx) & 1) == 0)) && ty >= 0 && tx >= 0 && ((ty >> sh2) < a.Hi) &&
                                     ((tx >> sh2) < a.Wi);
                            }
                            mk |= ok ? (1u << t) : 0u;
                        }
                    mask[q] = mk;
                }
            } else {
                const int row = ((p - PL * PA) % PB) * 16 + prow, n = n0 + row;
                const int bchunk = swz_chunk<T>(b_rho<NT>(row & 63), lane & 3);
                base[q] = n < a.N ? (int)((((int64_t)(n >> 6) * KT) * 2048 + (n & 63) * 32 + bchunk * 8) * 2) : (int)OOB;
            }
        }
        // (K order: every workgroup walks K from step 0.  The workgroups of an XCD share one L2 and -- on the m-fastest tile walk -- the
        // same weight rows; letting each start at its own K step, to spread their simultaneous requests over the L2 channels, LOSES:
        // the weight working set of the moment grows from one K step to the whole matrix and falls out of the 4 MB L2 (ASPP 3x3
        // 278 -> 389 us in bf16, 732 -> 870 us on two planes; gpurun_out/r04 h2_krot).)
        // Two planes: K walks 64-channel groups outermost and the filter taps inside a group.  The workgroups of an XCD hold
        // neighbouring row blocks and move through K together, so at any moment they read ONE channel group of one contiguous
        // row range (+- the halo) whatever the tap: ~1 MB that stays in the XCD's 4 MB L2 across the nine taps.  With the taps
        // outermost a tap sweeps all C channels of that range (4.7 MB at C = 256) before the next tap comes back to the same
        // lines, and every tap was a miss (PMC: 3x3 256 -> 256 at 48 x 48 fetched 4.4 x its algorithmic bytes, the decoder's
        // data gradient 12.6 x).  64 channels = the two K steps that share a 128-byte line.
        constexpr bool TAPIN = PL == 2 && DML_WS_TAP_INNER != 0;
        const int ksteps_c = a.C / BK;
        int ir = 0, is = 0, ic0 = 0, cg0 = 0;
        for (int kt = 0; kt < KT; ++kt) {
            if (g >= (uint32_t)NST) {
                // the stage this K step overwrites must have been read by every consumer wave
                const uint32_t need = g - NST + 1;
                bool drained = false;
                for (;;) {
                    uint32_t mn = ws_ld(consumed);
#pragma unroll
                    for (int w = 1; w < NCW; ++w) mn = min(mn, ws_ld(consumed + w));
                    if (mn >= need) break;
                    if (!drained) {
                        // ring full, nothing to issue: let everything in flight land and publish it now instead of D issues
                        // later -- with the shallow ring of the two-plane mode (3 stages) the consumers otherwise only ever
                        // see one stage ahead
                        wait_vmcnt<0>();
                        ws_st(ready + LW, g);
                        pub = g;
                        drained = true;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            }
            const uint32_t sbase = lds0 + (g % NST) * SB;
            const uint32_t tapbit = 1u << (ir * a.S + is);
            const int ktw = TAPIN ? (ir * a.S + is) * ksteps_c + ic0 / BK : kt;      // the K step's tile in the (tap-major) weight copy
            const int soff = (MODE == 0 ? ((ir * a.dil) * a.Wi + is * a.dil) * a.ldx
                                        : -((((ir * a.dil) >> sh2) * a.Wi + ((is * a.dil) >> sh2)) * a.ldx)) * 2 + ic0 * 2;
#pragma unroll
            for (int q = 0; q < MYP; ++q) {
                const int p = q * NLD + LW;
                if (PL == 2 && (DML_WS_ABL & 4)) continue;
                if (p < PL * PA) {
                    const uint32_t voff = (mask[q] & tapbit) ? (uint32_t)(base[q] + soff) : OOB;
                    ws_dma16(rs_x, sbase + p * 1024, voff, p < PA ? 0u : a.x_plane_bytes);
                } else {
                    ws_dma16(rs_w, sbase + p * 1024, (uint32_t)base[q],
                             (uint32_t)ktw * 4096u + ((p - PL * PA) < PB ? 0u : a.w_plane_bytes));
                }
            }
            ic0 += BK;
            if (TAPIN) {
                if (ic0 >= a.C || ic0 >= cg0 + 64) {             // this tap's share of the channel group is done
                    ic0 = cg0;
                    if (++is == a.S) {
                        is = 0;
                        if (++ir == a.R) { ir = 0; cg0 += 64; ic0 = cg0; }
                    }
                }
            } else if (ic0 >= a.C) {
                ic0 = 0;
                if (++is == a.S) { is = 0; ++ir; }
            }
            ++g;
            if (g > (uint32_t)D + pub) {
                wait_vmcnt<D * MYP>();             // at most D stages of this wave's pieces in flight: stage g - 1 - D has landed
                pub = g - D;
                ws_st(ready + LW, pub);
            }
        }
    }
    wait_vmcnt<0>();
    ws_st(ready + LW, g);
}

typedef _Float16 mfma_f16x8 __attribute__((ext_vector_type(8)));

// EPI (two planes, data gradient): 0 no epilogue operand, 1 accumulate / identity-branch gradient, 2 fused BatchNorm-backward sums,
// 3 both -- ONE epilogue per instantiation (all of them behind run-time branches in one kernel: 144 accumulators live at a four-way
// fork, 400-700 bytes of scratch per lane)
template <int MW, int NW, int MODE, int NLD, int PL, int MT, int EPI>
__device__ __forceinline__ void conv_ws_body(const ConvArgs& a, const uint32_t x_bytes, const uint32_t w_bytes) {
    typedef bf16_t T;
    static_assert(MW * NW == 4 || (PL == 2 && MW == 1 && NW == 2 && NLD == 2),
                  "four consumer waves, one per SIMD -- or the half-tile configuration: two consumer + two loader waves, two workgroups per CU");
    static_assert(MT % 3 == 0 && MT >= 3, "48-row sub-tiles (statistics groups, row epilogue)");
    constexpr int NCW = MW * NW, NT = 4, NST = ws_ring_stages(PL, NCW);
    constexpr int BM = 16 * MT * MW, BN = 64 * NW;
    constexpr int SB = PL * (BM + BN) * BK * 2;
    // two planes, 144 x 256: 2 KB of private staging per consumer wave behind the flags (conv_epilogue_rows8); the 288 x 128
    // configuration has no room for it and stages in the slot of the tile's last K stage (conv_epilogue_rows)
    // (forward only: nearly every data gradient of a plan carries epilogue operands -- accumulate, fused BN-backward sums -- whose
    // loads want the deeper row groups of conv_epilogue_rows, and both epilogues in one kernel spill 300 bytes per lane)
    constexpr bool PRIV_STAGE = PL == 2 && MW == 1 && MODE == 0;
    // two planes, 192 x 64 (MW = 4, MT = 3: the 64-channel layers; 48 x 64 wave tiles): its ring is 3 x 32 KB, so the row epilogue's
    // 12 KB per wave sit BESIDE the ring in both directions -- no held-back slot, no consumer barrier (conv_epilogue_rows)
    constexpr bool PRIV_ROWS = PL == 2 && MW == 4;
    constexpr int ROWS_STAGE = PRIV_ROWS ? WS_STAT_ROWS * 256 : 0;
    // (two planes, data gradient: 768 bytes per consumer wave for the ReLU masks of a 48-row sub-tile, conv_epilogue_rows)
    constexpr int MASK_STAGE = (PL == 2 && !PRIV_STAGE) ? WS_STAT_ROWS * 16 : 0;
    __shared__ __attribute__((aligned(1024))) char smem[NST * SB + 64 + (PRIV_STAGE ? NCW * 2048 : NCW * (MASK_STAGE + ROWS_STAGE))];
    static_assert(sizeof(smem) <= 160 * 1024, "LDS of one CU");
    uint32_t* const ready = reinterpret_cast<uint32_t*>(smem + NST * SB);          // [NLD] stages landed, per loader
    uint32_t* const consumed = ready + 4;                                          // [4] stages whose reads were issued

    const int tid = threadIdx.x, lane = tid & 63;
    int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 16) reinterpret_cast<uint32_t*>(smem + NST * SB)[tid] = 0;
    if constexpr (NCW == 2) {
        __syncthreads();
        // Half-tile configuration: workgroups b and b + grid / 2 share a CU, and the four waves of a workgroup sit on the CU's four
        // SIMDs in an order that rotates from workgroup to workgroup (tools/probe_wg_placement.hip, profiles/r06_wg_placement.txt).
        // Roles by SIMD, so that the CU's four consumer waves own a SIMD each: the first workgroup's consumers are its waves on SIMDs
        // 0 / 1, the second's those on SIMDs 2 / 3.  (Were two waves of a workgroup ever on one SIMD, roles by wave index.)
        uint32_t hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        const int simd = (int)((hw >> 4) & 3u);
        uint32_t* const sid = reinterpret_cast<uint32_t*>(smem + NST * SB) + 12;      // [4] (words 0 .. 11: ready / consumed / edone)
        if (lane == 0) sid[wave] = 1u << simd;
        __syncthreads();
        const bool perm = (sid[0] | sid[1] | sid[2] | sid[3]) == 0xfu;
        const int slot = (blockIdx.x >= (gridDim.x + 1) / 2) ? 1 : 0;
        if (perm) wave = __builtin_amdgcn_readfirstlane(((simd >> 1) == slot ? 0 : NCW) + (simd & 1));
        // the second workgroup of a CU starts late, so that its epilogues fall under the first one's K loops and vice versa: left to
        // themselves the two start together and do the same thing at the same time
        if (slot == 1 && a.half_stagger > 0) {
            const uint64_t t0 = wall_clock64();
            while (wall_clock64() - t0 < (uint64_t)a.half_stagger) __builtin_amdgcn_s_sleep(16);
        }
    }
    __syncthreads();

    const int ntiles = a.nblk_m * a.nblk_n;
    const int first_tile = xcd_remap(blockIdx.x, gridDim.x);      // this workgroup walks first_tile, + grid, + 2 grid, ...
    const int KT = a.Ktot / BK;

    if (wave >= NCW) {
        const int lw = wave - NCW;
        if (lw == 0) conv_ws_loader<MW, NW, MODE, NLD, 0, PL, MT>(a, x_bytes, w_bytes, smem, ready, consumed, lane, ntiles, first_tile);
        if (NLD > 1 && lw == 1) conv_ws_loader<MW, NW, MODE, NLD, (NLD > 1 ? 1 : 0), PL, MT>(a, x_bytes, w_bytes, smem, ready, consumed, lane, ntiles, first_tile);
        if (NLD > 2 && lw == 2) conv_ws_loader<MW, NW, MODE, NLD, (NLD > 2 ? 2 : 0), PL, MT>(a, x_bytes, w_bytes, smem, ready, consumed, lane, ntiles, first_tile);
        if (NLD > 3 && lw == 3) conv_ws_loader<MW, NW, MODE, NLD, (NLD > 3 ? 3 : 0), PL, MT>(a, x_bytes, w_bytes, smem, ready, consumed, lane, ntiles, first_tile);
        return;
    }

    // ---------------------------------------------------------------------- consumer
    __builtin_amdgcn_s_setprio(3);      // the MFMA waves ahead of the loader wave that shares their SIMD (+0.15 % on the step)
    const int wm = wave / NW, wn = wave % NW;
    const int lr = lane & 15, lq = lane >> 4;
    constexpr bool priv = PRIV_STAGE || PRIV_ROWS;      // the epilogue stages beside the ring: a tile's last stage is released like any other
    uint32_t g = 0, rflag = 0, tiles_done = 0;
    uint32_t* const edone = consumed + 4;                  // [4] tiles whose last fragment reads were issued, per consumer wave
    auto read_ready = [&]() -> uint32_t {
        uint32_t v = ws_ld(ready);
#pragma unroll
        for (int w = 1; w < NLD; ++w) v = min(v, ws_ld(ready + w));
        return v;
    };
    auto wait_ready = [&](const uint32_t need) {
        while (rflag < need) {
            __builtin_amdgcn_s_sleep(1);
            rflag = read_ready();
        }
        asm volatile("" ::: "memory");
    };
    constexpr int A_PLANE = BM * BK * 2, B_PLANE = BN * BK * 2;      // bytes from the hi plane to the lo plane inside a stage
    constexpr bool EPI_OPS = PL == 2 && MODE == 1 && !PRIV_STAGE && DML_WS_EPI_OPS != 0;
    static_assert(EPI == 0 || EPI_OPS, "epilogue operands: two-plane data gradients");
    constexpr bool hold2 = EPI != 0 && !priv;

    for (int tile = first_tile; tile < ntiles; tile += (int)gridDim.x) {
        const int blk_m = ws_tile_m<PL>(tile, a), blk_n = ws_tile_n<PL>(tile, a);
        // fragment byte offsets inside a stage (hi plane; the lo plane is BM / BN rows further).  Recomputed per tile from a lane
        // index the optimiser cannot see through: hoisted out of the tile loop these 13 registers stay live across the epilogue,
        // where the 144 accumulator registers leave no room for them (spills, and a scratch reload waits vmcnt(0) -- for every
        // store issued before it).
        int lr_k = lr, lq_k = lq;
        asm volatile("" : "+v"(lr_k), "+v"(lq_k));
        int a_off[MT], b_off[NT];
        // (the chunk swizzle of a 64-byte bf16 / fp16 row looks at row bits 2-3 only: fragment j sits j KB behind fragment 0 -- written
        // that way, the nine offsets are ONE register and immediates)
        static_assert(BK * 2 * 16 == 1024, "16 rows of 64 bytes per fragment");
        const int a_off0 = (wm * (16 * MT) + lr_k) * (BK * 2) + swz_chunk<T>(lr_k & 15, lq_k) * 16;
#pragma unroll
        for (int j = 0; j < MT; ++j) a_off[j] = a_off0 + j * 1024;
#pragma unroll
        for (int i = 0; i < NT; ++i) b_off[i] = PL * BM * BK * 2 + (wn * 64 + b_row<NT>(i, lr_k)) * (BK * 2) + swz_chunk<T>(lr_k, lq_k) * 16;
        // accumulators by 48-row sub-tile: ACC(i, j) = acc3[j / 3][i][j % 3], so that the epilogue takes a sub-tile by reference
        f32x4 acc3[MT / 3][NT][3];
#define ACC(i, j) acc3[(j) / 3][i][(j) % 3]
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) ACC(i, j) = (f32x4){0.f, 0.f, 0.f, 0.f};

        if constexpr (PL == 1) {
            mfma_bf16x8 bfA[NT], bfB[NT], af[MT], alA, alB;       // af[MT - 1] unused: the last A fragment alternates alA / alB
            wait_ready(g + 1);                                      // first K step of the tile: exposed once per tile
            {
                const char* sb = smem + (g % NST) * SB;
#pragma unroll
                for (int i = 0; i < NT; ++i) bfA[i] = *reinterpret_cast<const mfma_bf16x8*>(sb + b_off[i]);
#pragma unroll
                for (int j = 0; j < MT - 1; ++j) af[j] = *reinterpret_cast<const mfma_bf16x8*>(sb + a_off[j]);
                alA = *reinterpret_cast<const mfma_bf16x8*>(sb + a_off[MT - 1]);
            }
            // one K step: MFMAs of step g from (bc, af, alc); the fragments of step g + 1 into (bn, af, aln) as registers free up
            auto step = [&](mfma_bf16x8 (&bc)[NT], mfma_bf16x8 (&bn)[NT], mfma_bf16x8& alc, mfma_bf16x8& aln, const bool has_next) {
                asm volatile("" ::: "memory");
                ws_st(consumed + wave, g + 1);         // every read of stage g has been issued (DS runs a wave's operations in order)
                const char* sn = smem + ((g + 1) % NST) * SB;
                if (has_next) {
                    wait_ready(g + 2);
#pragma unroll
                    for (int i = 0; i < NT; ++i) bn[i] = *reinterpret_cast<const mfma_bf16x8*>(sn + b_off[i]);
                    aln = *reinterpret_cast<const mfma_bf16x8*>(sn + a_off[MT - 1]);
                }
#pragma unroll
                for (int j = 0; j < MT - 1; ++j) {
#pragma unroll
                    for (int i = 0; i < NT; ++i) ACC(i, j) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[i], af[j], ACC(i, j), 0, 0, 0);
                    if (has_next) af[j] = *reinterpret_cast<const mfma_bf16x8*>(sn + a_off[j]);
                    if (j == (MT - 1) / 2) rflag = read_ready();      // the next step's poll, answered under the MFMAs
                }
#pragma unroll
                for (int i = 0; i < NT; ++i) ACC(i, MT - 1) = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[i], alc, ACC(i, MT - 1), 0, 0, 0);
                ++g;
            };
            int kt = 0;
            for (; kt + 1 < KT; kt += 2) {
                step(bfA, bfB, alA, alB, true);
                step(bfB, bfA, alB, alA, kt + 2 < KT);
            }
            if (kt < KT) step(bfA, bfB, alA, alB, false);
        } else {
            // two fp16 planes per operand: x s = xh + xl, w t = wh + wl; per 16 x 16 x 32 block the three products wh xh, wh xl,
            // wl xh (the dropped wl xl is below 2^-22 of the block) -- 108 MFMAs per K step against 26 fragment reads.
            // Every fragment is in flight one row group (12 MFMAs, ~200 cycles) before its first use: the activation pair of
            // group j + 1 is read ahead of group j's MFMAs (two register slots; 9 groups per step, so the slot of group 0
            // alternates from step to step: P), the hi weight fragments of step k + 1 under groups 5.. of step k (two sets,
            // alternating), the lo weight fragments -- used LAST in every group -- right after step k's final MFMAs, under the
            // first eight of step k + 1.  __builtin_amdgcn_sched_barrier pins that order: left to itself hipcc sinks each
            // read to just before its use and waits lgkmcnt(0) there, nine exposed LDS latencies per step (measured: 1.45 us per
            // step against 0.72 of MFMA issue).
            // read-ahead distance of the activation fragments in row groups, and their slots.  Two groups (~24 MFMAs) on the 144-row
            // wave tiles; the 48-row ones (three groups per step) keep one: two ahead would read the NEXT stage from group 1 on, before
            // the poll of that stage has been answered
            constexpr int AH = (DML_WS_AHEAD == 2 && MT >= 9) ? 2 : 1, NS_A = AH + 1;
            static_assert(AH == 1 || MT % NS_A == 0, "slot of group j = j % 3 in every step");
            mfma_f16x8 bhA[NT], bhB[NT], bl[NT], ah[NS_A], al[NS_A];
            uint32_t pl[NLD];
#define WS_FRAG(p) (*reinterpret_cast<const mfma_f16x8*>(p))
            wait_ready(g + 1);                                      // first K step of the tile: exposed once per tile
            {
                const char* sb = smem + (g % NST) * SB;
#pragma unroll
                for (int i = 0; i < NT; ++i) {
                    bhA[i] = WS_FRAG(sb + b_off[i]);
                    bl[i] = WS_FRAG(sb + b_off[i] + B_PLANE);
                }
                ah[0] = WS_FRAG(sb + a_off[0]);
                al[0] = WS_FRAG(sb + a_off[0] + A_PLANE);
                if constexpr (AH == 2) {
                    ah[1] = WS_FRAG(sb + a_off[1]);
                    al[1] = WS_FRAG(sb + a_off[1] + A_PLANE);
                }
            }
            auto step = [&](auto pc, mfma_f16x8 (&bc)[NT], mfma_f16x8 (&bn)[NT], const bool has_next, const bool rel) {
                constexpr int P = decltype(pc)::value;
                const char* sb = smem + (g % NST) * SB;
                // (last step of the tile: the "next" reads fall on this stage again and are never used)
                const char* sn = has_next ? smem + ((g + 1) % NST) * SB : sb;
#pragma unroll
                for (int j = 0; j < MT; ++j) {
                    // (positions inside the step: row group 1 / 4 / 5.. of the nine of a 144-row wave tile, 0 / 1 / 2 of a 48-row one)
                    constexpr int JP = MT >= 9 ? 1 : 0, JW = MT >= 9 ? 4 : 1, JB = MT >= 9 ? 5 : MT - 1;
#if DML_WS_SPREAD
                    // The non-MFMA instructions of a row group SPREAD between its three MFMA quads instead of clustered in front of
                    // them: a wave issues in order, and a cluster of five or six of them (two fragment reads, waits, hazard nops)
                    // outlasts the 16 cycles of the MFMA before it -- the ablations put 25-30 % of the K loop on the fragment reads
                    // although the LDS itself is busy a fifth of the time (profiles/r05_h2_kloop_ablations.txt).
                    // fragment slots: this group's and the one of the group read ahead (AH = 1: two slots, alternating from step to
                    // step with P because MT is odd; AH = 2: three slots, slot = j % 3 in every step)
                    const int CUR = AH == 2 ? j % NS_A : (j + P) & 1, NXT = AH == 2 ? (j + AH) % NS_A : (j + 1 + P) & 1;
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) WS_MFMA_F16(ACC(i, j), bc[i], ah[CUR]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + AH < MT) {
                        if (!(DML_WS_ABL & 1)) ah[NXT] = WS_FRAG(sb + a_off[(j + AH) % MT]);
                    } else {
                        asm volatile("" ::: "memory");
                        if (!(DML_WS_ABL & 1)) ah[NXT] = WS_FRAG(sn + a_off[(j + AH) % MT]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) WS_MFMA_F16(ACC(i, j), bc[i], al[CUR]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j + AH < MT) {
                        if (!(DML_WS_ABL & 1)) al[NXT] = WS_FRAG(sb + a_off[(j + AH) % MT] + A_PLANE);
                    } else {
                        asm volatile("" ::: "memory");
                        // every read of stage g has been issued (the tile's LAST stage is announced after the epilogue, which
                        // stages the output rows in its slot)
                        if (j + 1 == MT && (rel || priv)) ws_st(consumed + wave, g + 1);
                        if (!(DML_WS_ABL & 1)) al[NXT] = WS_FRAG(sn + a_off[(j + AH) % MT] + A_PLANE);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) WS_MFMA_F16(ACC(i, j), bl[i], ah[CUR]);
                    __builtin_amdgcn_sched_barrier(0);
                    if (j == JP && !(DML_WS_ABL & 2)) {     // the next step's poll, answered under the MFMAs
#pragma unroll
                        for (int w = 0; w < NLD; ++w) pl[w] = ws_ld(ready + w);
                    }
                    if (j == JW && !(DML_WS_ABL & 2)) {
                        rflag = pl[0];
#pragma unroll
                        for (int w = 1; w < NLD; ++w) rflag = min(rflag, pl[w]);
                        if (has_next) wait_ready(g + 2);
                    }
                    if (!(DML_WS_ABL & 1)) {
                        if (MT >= 9) {                      // the next step's hi weight fragments, one per row group
                            if (j >= JB && j < JB + NT) bn[j - JB] = WS_FRAG(sn + b_off[j - JB]);
                        } else if (j == JB) {
#pragma unroll
                            for (int i = 0; i < NT; ++i) bn[i] = WS_FRAG(sn + b_off[i]);
                        }
                    }
#else
                    if (j + 1 < MT) {
                        if (!(DML_WS_ABL & 1)) {
                            ah[(j + 1 + P) & 1] = WS_FRAG(sb + a_off[j + 1]);
                            al[(j + 1 + P) & 1] = WS_FRAG(sb + a_off[j + 1] + A_PLANE);
                        }
                    } else {
                        asm volatile("" ::: "memory");
                        // every read of stage g has been issued (the tile's LAST stage is announced after the epilogue, which
                        // stages the output rows in its slot)
                        if (rel || priv) ws_st(consumed + wave, g + 1);
                        if (!(DML_WS_ABL & 1)) {
                            ah[(MT + P) & 1] = WS_FRAG(sn + a_off[0]);
                            al[(MT + P) & 1] = WS_FRAG(sn + a_off[0] + A_PLANE);
                        }
                    }
                    if (j == JP && !(DML_WS_ABL & 2)) {     // the next step's poll, answered under the MFMAs
#pragma unroll
                        for (int w = 0; w < NLD; ++w) pl[w] = ws_ld(ready + w);
                    }
                    if (j == JW && !(DML_WS_ABL & 2)) {
                        rflag = pl[0];
#pragma unroll
                        for (int w = 1; w < NLD; ++w) rflag = min(rflag, pl[w]);
                        if (has_next) wait_ready(g + 2);
                    }
                    if (j == JB && !(DML_WS_ABL & 1)) {
#pragma unroll
                        for (int i = 0; i < NT; ++i) bn[i] = WS_FRAG(sn + b_off[i]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) ACC(i, j) = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc[i], ah[(j + P) & 1], ACC(i, j), 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) ACC(i, j) = __builtin_amdgcn_mfma_f32_16x16x32_f16(bc[i], al[(j + P) & 1], ACC(i, j), 0, 0, 0);
#pragma unroll
                    for (int i = 0; i < NT; ++i) ACC(i, j) = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[i], ah[(j + P) & 1], ACC(i, j), 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#endif
                }
                if (!(DML_WS_ABL & 1)) {
#pragma unroll
                    for (int i = 0; i < NT; ++i) bl[i] = WS_FRAG(sn + b_off[i] + B_PLANE);
                }
                __builtin_amdgcn_sched_barrier(0);
                ++g;
            };
            // (a data gradient with epilogue operands holds back the slots of its last TWO K stages: conv_epilogue_rows_ops)
            int kt = 0;
            for (; kt + 1 < KT; kt += 2) {
                step(std::integral_constant<int, 0>{}, bhA, bhB, true, !(hold2 && kt + 2 == KT));
                step(std::integral_constant<int, 1>{}, bhB, bhA, kt + 2 < KT, kt + 2 < KT && !(hold2 && kt + 3 == KT));
            }
            if (kt < KT) step(std::integral_constant<int, 0>{}, bhA, bhB, false, false);
        }
        // cut the accumulators' live ranges (see conv_igemm_kernel), then the shared epilogue per 48-row group
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(ACC(i, j)));
        if constexpr (PL == 2) {
            const float un = a.x_unscale[0] * a.w_unscale[0];      // exact powers of two
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j) ACC(i, j) *= un;
        }
        // (three explicit copies: hipcc does not unroll a loop around the inlined epilogue, and a run-time index into acc
        // sends all 36 accumulator fragments through scratch memory -- 37 MB written and read back per launch, +25 us)
        char* rows_stage = nullptr;
        if (PRIV_STAGE) rows_stage = smem + NST * SB + 64 + wave * 2048;
        if (PRIV_ROWS) rows_stage = smem + NST * SB + 64 + NCW * MASK_STAGE + wave * ROWS_STAGE;
        if (PL == 2 && !priv) {
            // the slot of the tile's last K step becomes the staging area of conv_epilogue_rows (12 KB per wave): every consumer
            // wave must have issued its last fragment reads first
            ++tiles_done;
            asm volatile("" ::: "memory");
            ws_st(edone + wave, tiles_done);
            for (;;) {
                uint32_t mn = ws_ld(edone);
#pragma unroll
                for (int w = 1; w < NCW; ++w) mn = min(mn, ws_ld(edone + w));
                if (mn >= tiles_done) break;
                __builtin_amdgcn_s_sleep(1);
            }
            asm volatile("" ::: "memory");
            rows_stage = smem + ((g - 1) % NST) * SB + wave * (WS_STAT_ROWS * 256);
        }
        // One copy of the epilogue code, run three times on acc[.][0..2] with the accumulators rotated down by a sub-tile in
        // between (2 x 96 register moves): three inlined copies -- what a compile-time sub-tile index costs; a run-time index into
        // acc would send all of it through scratch memory -- made these kernels 10-19 K instructions, more than the instruction
        // cache two CUs share.
        // forward, 144-row wave tiles: the BatchNorm statistics of the whole wave tile at once (ws_tile_stats)
        constexpr bool TILE_STATS = PL == 2 && MODE == 0 && MT == 9;
        if constexpr (TILE_STATS) {
            if (a.stats != nullptr && !(DML_WS_ABL & 8)) ws_tile_stats<NT, MT / 3>(acc3, a, blk_m * BM + wm * (16 * MT), blk_n * BN + wn * 64, lr, lq);
        }
        if constexpr (EPI != 0 && !(DML_WS_ABL & 8))
            conv_epilogue_rows_ops<NT, MT / 3, (EPI & 1) != 0, (EPI & 2) != 0>(
                acc3, a, blk_m * BM + wm * (16 * MT), blk_n * BN + wn * 64, lane, rows_stage,
                priv ? rows_stage : smem + ((g - 2) % NST) * SB + wave * (WS_STAT_ROWS * 256), smem + NST * SB + 64 + wave * MASK_STAGE);
#pragma clang loop unroll(disable)
        for (int h = 0; h < ((EPI != 0 || (DML_WS_ABL & 8)) ? 0 : MT / 3); ++h) {      // (ablation 8: no epilogue at all)
            const int mw0 = blk_m * BM + wm * (16 * MT) + h * WS_STAT_ROWS, nw0 = blk_n * BN + wn * 64;
            if constexpr (PL == 1) conv_epilogue<bf16_t, NT, 3, MODE, false>(acc3[0], a, mw0, nw0, lr, lq);
            else if (MODE == 0 && (a.bias != nullptr || a.post_scale != nullptr)) {
                conv_epilogue<float, NT, 3, MODE, false>(acc3[0], a, mw0, nw0, lr, lq);      // (inference epilogue, bias: scattered stores)
            } else if (PRIV_STAGE) {
                if constexpr (PRIV_STAGE) conv_epilogue_rows8<NT, MODE>(acc3[0], a, mw0, nw0, lane, rows_stage, !TILE_STATS);
            } else {
                conv_epilogue_rows<NT, MODE, !EPI_OPS>(acc3[0], a, mw0, nw0, lane, rows_stage, smem + NST * SB + 64 + wave * MASK_STAGE,
                                                       !TILE_STATS);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // its reads of the staging area, before the next group's writes
            }
            if constexpr (MT == 9) {
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) {
                        acc3[0][i][j] = acc3[1][i][j];
                        acc3[1][i][j] = acc3[2][i][j];
                        // (keeps the loop a loop: the optimiser must not see through the rotation and re-specialise the three trips)
                        asm volatile("" : "+v"(acc3[0][i][j]), "+v"(acc3[1][i][j]));
                    }
            } else {
                static_assert(MT == 9 || MT == 3, "accumulator rotation of the sub-tile loop");
            }
        }
#undef ACC
        if (PL == 2 && !priv) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ws_st(consumed + wave, g);                  // the tile's last stage: its slot is free again
        }
    }
}

template <int MW, int NW, int MODE, int NLD, int PL = 1, int MT = WS_MT, int EPI = 0>
__global__ __launch_bounds__((MW * NW + NLD) * 64) void conv_ws_kernel(const ConvArgs a, const uint32_t x_bytes, const uint32_t w_bytes) {
    conv_ws_body<MW, NW, MODE, NLD, PL, MT, EPI>(a, x_bytes, w_bytes);
}
// the half-tile configuration (144 x 128, two consumer + two loader waves): at most 256 registers, so that two workgroups share a CU
template <int MODE, int EPI>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void conv_ws_half_kernel(const ConvArgs a, const uint32_t x_bytes,
                                                                                                const uint32_t w_bytes) {
    conv_ws_body<1, 2, MODE, 2, 2, WS_MT, EPI>(a, x_bytes, w_bytes);
}

// may this launch run on conv_ws_kernel?  (shared by launch_conv and dml_conv_stat_rows)
static bool conv_ws_eligible(const ConvArgs& a, const int64_t xb, const int64_t wb, const int mode = 1) {
    // bf16 plans: OPT-IN (DML_CONV_WS=1).  Per launch it wins where K is long (rule below), but in the train step the persistent
    // 150 KB-of-LDS workgroups keep the side stream's weight-gradient workgroups off the CUs they hold and static tile lists
    // cannot rebalance around them: whole step 391.5 / 392.2 images/s with it, 395.3 / 394.9 without (two interleaved pairs,
    // profiles/r04_ab_ws.txt).  The two-plane fp32 mode (conv_ws_planes_eligible) always runs on it.
    // (forward-only for the long-K launches, where nothing runs beside the main stream, measured +0.1...0.4 % on the step -- and moved
    // the bf16 plan's rounding (48-row statistics groups) enough to redraw the chaotic 30-step trajectory of
    // tests/test_gpu_training_equivalence.py past its bar; not worth a different default)
    static const int ws_on = getenv("DML_CONV_WS") ? atoi(getenv("DML_CONV_WS")) : 0;
    if ((!ws_on && a.ws_min_tiles <= 0) || !a.w_tiled || (a.C % BK) != 0 || a.R * a.S > 32 || (a.N % 128) != 0) return false;
    if (xb >= (1ll << 31) || wb >= (1ll << 31)) return false;
    // long K loops only: a consumer wave runs its tile's epilogue itself, with nothing of the same workgroup to cover it, and
    // below ~32 K steps per tile that costs more than the K loop gains (tools/bench_ws.py, profiles/r04_bench_ws_2.txt: K = 256
    // with four tiles per workgroup 42.5 vs 31.6 us, K = 512 120 vs 89; K = 1024 28.6 vs 32.0, K = 2304 52.9 vs 60.5, K = 4608
    // 158 vs 207, K = 18432 284 vs 372).  ws_min_tiles > 0 (tests) lifts the rule.
    if (a.ws_min_tiles <= 0 && a.R * a.S * a.C < 1024) return false;
    // a persistent workgroup per CU: the tile list must fill most of the chip
    const int bn = (a.N % 256) == 0 ? 256 : 128, bm = (a.N % 256) == 0 ? 144 : 288;
    const int64_t ntiles = ((int64_t)a.M + bm - 1) / bm * (a.N / bn);
    return ntiles >= (a.ws_min_tiles > 0 ? a.ws_min_tiles : 192);
}
// two planes, forward: 48 x 256 tiles instead of 144 x 256 for launches that leave most of the chip empty (launch_conv; also decides
// the rows per statistics partial: dml_conv_stat_rows)
static bool ws_planes_short(const ConvArgs& a, const int mode) {
    static const int short_on = getenv("DML_WS_SHORT") ? atoi(getenv("DML_WS_SHORT")) : 1;
    return mode == 0 && short_on != 0 && (a.N % 256) == 0 && ((a.M + 143) / 144) * (a.N / 256) * 2 <= 256;
}
// rows of the GEMM per BatchNorm statistics partial of a two-plane FORWARD launch: the whole 144-row wave tile (ws_tile_stats) except on
// the 48-row wave tiles (64 output channels, short tiles)
static int ws_planes_stat_rows(const ConvArgs& a, const int mode) {
    if (mode != 0) return 48;
    return (a.N == 64 || ws_planes_short(a, mode)) ? 48 : 144;
}
// the same kernel on two fp16 planes per operand (fp32 tensors, f32_split == 2): every shape it can address -- the alternative
// is the three-term split kernel at a third of its rate
static bool conv_ws_planes_eligible(const ConvArgs& a, const int mode) {
    if (mode == 0 && a.accum) return false;      // (an accumulating FORWARD launch: no plan issues one; the forward epilogue has no such path)
    if (a.f32_split != 2 || !a.x_planes || !a.w_planes || !a.x_unscale || !a.w_unscale) return false;
    // (N = 320: the decoder's data gradient.  N = 64 -- layer1's 3x3 and the 256 -> 64 1x1 -- runs the 288 x 128 configuration with
    // half of every tile empty: zero weight rows through the descriptor's range check, no stores; still ahead of the three-term
    // kernel those layers took before: whole step +0.4 %, two interleaved pairs)
    if ((a.C % BK) != 0 || a.R * a.S > 32 || (a.N % 64) != 0) return false;
    if (mode == 1 && a.Ktot < 2 * BK) return false;      // (conv_epilogue_rows_ops stages in the slots of a tile's last two K stages)
    // whole 16-byte vectors of the fp32 output and of every epilogue operand (conv_epilogue_rows)
    if ((a.ldy & 3) != 0 || (a.post_res != nullptr && (a.post_ldres & 3) != 0) ||
        ((reinterpret_cast<uintptr_t>(a.y) | reinterpret_cast<uintptr_t>(a.post_res) | reinterpret_cast<uintptr_t>(a.bias) |
          reinterpret_cast<uintptr_t>(a.post_scale) | reinterpret_cast<uintptr_t>(a.post_shift) |
          reinterpret_cast<uintptr_t>(a.post_mean)) & 15) != 0)
        return false;
    const int64_t xb = (((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2, wb = (int64_t)a.N * a.Ktot * 2;
    return xb + a.x_plane_bytes < (1ll << 31) && wb + a.w_plane_bytes < (1ll << 31);
}

// ------------------------------------------------------------------------------------------------
// weight-gradient kernel: dw[n][kc] += sum_m dy[m][n] * x[src(m, tap(kc))][c(kc)]
// tile 128 (n) x 128 (kc), K loop over 32-pixel slabs, split over the pixel dimension.
// Both operands are "K-major" in memory (pixel rows, channel contiguous): the bf16 fragments are
// read with the gfx950 LDS transpose read ds_read_b64_tr_b16, the fp32 ones need no transpose.
// ------------------------------------------------------------------------------------------------
struct WgradArgs {
    const void* x;
    const void* dy;
    float* dw;
    int B, Hi, Wi, C, ldx;
    int Ho, Wo, N, ldy;
    int R, S, stride, dil, pad;
    int M, Ktot;
    int nblk_n, nblk_k;
    int slab_tiles;   // K tiles (of 32 pixels) per split
    float* ws;        // optional split-K workspace [splits][N][Ktot]: plain stores instead of atomics
    FastDiv div_wo, div_howo, div_c;
    // f32_split == 2: fp16 hi / lo planes of both operands (DmlWgradDesc::x_planes ...)
    const void* x_planes;
    const void* dy_planes;
    const float* x_unscale;
    const float* dy_unscale;
    uint32_t x_plane_bytes, dy_plane_bytes;
};

template <typename T> struct WgLds;
template <> struct WgLds<float> { static constexpr int PITCH = 128 + 16; };    // floats; banks (kg*16 + i)
template <> struct WgLds<bf16_t> { static constexpr int PITCH = 128 + 8; };    // bf16

__device__ __forceinline__ bf16x4 lds_tr16_b64(const bf16_t* p) {
    // 16 lanes x 4 bf16 block transposed in hardware (ds_read_b64_tr_b16)
    typedef short v4i16 __attribute__((ext_vector_type(4)));
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)(p));
}

template <typename T>
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_kernel(const WgradArgs a) {
    constexpr int TILE = 128;
    constexpr int VEC = Elem<T>::VEC;
    constexpr int CV = TILE / VEC;                // 16-byte vectors per tile row (16 bf16 / 32 f32)
    constexpr int RPP = NTHREADS / CV;            // rows per pass (16 / 8)
    constexpr int LD = BK / RPP;                  // loads per thread per operand (2 / 4)
    constexpr int PITCH = WgLds<T>::PITCH;
    constexpr int MT = 4, NT = 4;                 // wave tile 64 (n) x 64 (kc)

    __shared__ __attribute__((aligned(16))) T smem[2 * 2 * BK * PITCH];
    auto Ys = [&](int buf) -> T* { return smem + buf * 2 * BK * PITCH; };
    auto Xs = [&](int buf) -> T* { return smem + buf * 2 * BK * PITCH + BK * PITCH; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    // 1-D grid, XCD-aware: every (n, kc) tile of one pixel slab runs on the same XCD, so the dy / x slabs are
    // fetched into that XCD's L2 once instead of once per XCD (PMC: 363 MB fetched per launch before, see DESIGN)
    const int ntile = a.nblk_n * a.nblk_k;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, tile_id = logical - split * ntile;
    const int blk_n = tile_id % a.nblk_n, blk_k = tile_id / a.nblk_n;
    const int n0 = blk_n * TILE, kc0 = blk_k * TILE;

    const T* __restrict__ X = static_cast<const T*>(a.x);
    const T* __restrict__ DY = static_cast<const T*>(a.dy);

    const int vcol = tid % CV, prow = tid / CV;
    // this thread's fixed filter tap / channel for the X operand
    const int kc = kc0 + vcol * VEC;
    const bool kc_ok = kc < a.Ktot;
    const uint32_t tap = fdiv((uint32_t)(kc_ok ? kc : 0), a.div_c);
    const int xc = (kc_ok ? kc : 0) - (int)tap * a.C;
    const int tr = (int)tap / a.S, ts = (int)tap - tr * a.S;
    const int dyo = tr * a.dil - a.pad, dxo = ts * a.dil - a.pad;
    const int yn = n0 + vcol * VEC;
    const bool yn_ok = yn < a.N;

    const int tile_beg = split * a.slab_tiles;
    const int tiles_total = (a.M + BK - 1) / BK;
    const int tile_end = min(tiles_total, tile_beg + a.slab_tiles);

    const bool lin1x1 = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0;
    uint4 y_reg[LD], x_reg[LD];
    auto load_tiles = [&](int t) {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            const int m = t * BK + prow + j * RPP;
            uint4 yv = make_uint4(0, 0, 0, 0), xv = make_uint4(0, 0, 0, 0);
            if (m < a.M) {
                if (yn_ok) yv = *reinterpret_cast<const uint4*>(DY + (int64_t)m * a.ldy + yn);
                if (kc_ok && lin1x1) {
                    xv = *reinterpret_cast<const uint4*>(X + (int64_t)m * a.ldx + xc);      // 1x1 stride 1: source pixel = m
                } else if (kc_ok) {
                    const uint32_t b = fdiv((uint32_t)m, a.div_howo);
                    const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
                    const uint32_t yo = fdiv(rem, a.div_wo);
                    const uint32_t xo = rem - yo * (uint32_t)a.Wo;
                    const int ys = (int)yo * a.stride + dyo, xs = (int)xo * a.stride + dxo;
                    if ((unsigned)ys < (unsigned)a.Hi && (unsigned)xs < (unsigned)a.Wi)
                        xv = *reinterpret_cast<const uint4*>(
                            X + ((int64_t)((int)b * a.Hi + ys) * a.Wi + xs) * a.ldx + xc);
                }
            }
            y_reg[j] = yv;
            x_reg[j] = xv;
        }
    };
    auto store_tiles = [&](int buf) {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            const int row = prow + j * RPP;
            int off;
            if constexpr (sizeof(T) == 2) {
                // bf16: 16-column sub-tiles [col/16][k'][16] (1056-byte pitch), k rows stored with bits 2/3 swapped:
                // conflict-free for ds_read_b64_tr_b16 (2 x 32 lanes, 64 banks) and for these 16-byte writes
                const int prow_ = (row & 0x13) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3);
                off = (vcol >> 1) * 528 + prow_ * 16 + (vcol & 1) * 8;
            } else {
                off = row * PITCH + vcol * VEC;
            }
            *reinterpret_cast<uint4*>(Ys(buf) + off) = y_reg[j];
            *reinterpret_cast<uint4*>(Xs(buf) + off) = x_reg[j];
        }
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (tile_beg < tile_end) {
        load_tiles(tile_beg);
        store_tiles(0);
    }
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    int cur = 0;
    for (int t = tile_beg; t < tile_end; ++t) {
        const bool has_next = t + 1 < tile_end;
        if (has_next) load_tiles(t + 1);
        const T* ys = Ys(cur) + (sizeof(T) == 2 ? wn * 4 * 528 : wn * 64);
        const T* xs = Xs(cur) + (sizeof(T) == 2 ? wk * 4 * 528 : wk * 64);
        if constexpr (sizeof(T) == 2) {
            // A[i = n][k = pixel]: lanes of a 16-group address the 4 x 16 block (rows k0..k0+3,
            // cols 16*i..) in 8-byte pieces: lane L -> row L/4, cols 4*(L%4)..; the hardware
            // returns to lane i the 4 k-values of column i.
            mfma_bf16x8 af[NT], bfr[MT];
            // k rows 8*lq + (lr>>2) (+4 for the second read) at their swapped positions inside a sub-tile
            const int p_lo = ((lr >> 2) | ((lq & 1) << 2) | ((lq >> 1) << 4)) * 16 + (lr & 3) * 4;
            const int p_hi = p_lo + 8 * 16;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const bf16x4 lo = lds_tr16_b64(ys + i * 528 + p_lo);
                const bf16x4 hi = lds_tr16_b64(ys + i * 528 + p_hi);
                bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[i] = __builtin_bit_cast(mfma_bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const bf16x4 lo = lds_tr16_b64(xs + j * 528 + p_lo);
                const bf16x4 hi = lds_tr16_b64(xs + j * 528 + p_hi);
                bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[j] = __builtin_bit_cast(mfma_bf16x8, v);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        } else {
#pragma unroll
            for (int kk = 0; kk < BK / 4; ++kk) {
                float af[NT], bfr[MT];
#pragma unroll
                for (int i = 0; i < NT; ++i) af[i] = ys[(kk * 4 + lq) * PITCH + i * 16 + lr];
#pragma unroll
                for (int j = 0; j < MT; ++j) bfr[j] = xs[(kk * 4 + lq) * PITCH + j * 16 + lr];
#pragma unroll
                for (int i = 0; i < NT; ++i)
#pragma unroll
                    for (int j = 0; j < MT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfr[j], acc[i][j], 0, 0, 0);
            }
        }
        if (has_next) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }

    // acc[i][j][q] = dw[n = n0 + wn*64 + i*16 + lq*4 + q][kc = kc0 + wk*64 + j*16 + lr]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + i * 16 + lq * 4 + q;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int k = kc0 + wk * 64 + j * 16 + lr;
                if (k < a.Ktot) {
                    if (a.ws != nullptr)
                        a.ws[((int64_t)split * a.N + n) * a.Ktot + k] = acc[i][j][q];
                    else
                        atomicAdd(a.dw + (int64_t)n * a.Ktot + k, acc[i][j][q]);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// fp32 weight gradient with the products on the bf16 matrix cores (three-term split, see conv_igemm_x3_kernel): the
// 128 x 128 tile / split-K scheme of conv_wgrad_kernel<float>, the fp32 operand tiles split into (hi, mid, lo) bf16 planes
// on their way from the staging registers into LDS -- in conv_wgrad_kernel<bf16_t>'s sub-tiled image, so that the
// fragments come out of ds_read_b64_tr_b16 -- and six MFMA groups per K step.  One LDS buffer (51 KB), two barriers per
// step, the next tile's split + store interleaved with the MFMAs, global loads a whole step ahead in registers.
// ------------------------------------------------------------------------------------------------
template <bool LIN>      // LIN: 1x1, stride 1, no padding -- pixel row m of x is output row m
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2))) void conv_wgrad_x3_kernel(
    const WgradArgs a, const uint32_t x_bytes, const uint32_t dy_bytes) {
    constexpr int TILE = 128, CV = TILE / 4, RPP = NTHREADS / CV, LD = BK / RPP;      // 32 float4 per row, 8 rows per pass, 4 loads
    constexpr int SUB = 528, PL = (TILE / 16) * SUB;                                   // bf16 elements per operand plane
    constexpr int MT = 4, NT = 4;
    constexpr uint32_t OOB = 0x80000000u;
    __shared__ __attribute__((aligned(16))) bf16_t smem[6 * PL];
    bf16_t* const Ys = smem;                     // planes hi, mid, lo of dy
    bf16_t* const Xs = smem + 3 * PL;            // planes of x

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int ntile = a.nblk_n * a.nblk_k;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, tile_id = logical - split * ntile;
    const int blk_n = tile_id % a.nblk_n, blk_k = tile_id / a.nblk_n;
    const int n0 = blk_n * TILE, kc0 = blk_k * TILE;
    // out-of-range rows / columns / taps read through the descriptors' range check: zeros, no branch, no access
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)dy_bytes, 0x00020000);
    const int vcol = tid % CV, prow = tid / CV;
    const int kc = kc0 + vcol * 4;
    const bool kc_ok = kc < a.Ktot;
    const uint32_t tap = fdiv((uint32_t)(kc_ok ? kc : 0), a.div_c);
    const int xc = (kc_ok ? kc : 0) - (int)tap * a.C;
    const int tr = (int)tap / a.S, ts = (int)tap - tr * a.S;
    const int dyo = tr * a.dil - a.pad, dxo = ts * a.dil - a.pad;
    const int yn = n0 + vcol * 4;
    const bool yn_ok = yn < a.N;
    const int tile_beg = split * a.slab_tiles;
    const int tiles_total = (a.M + BK - 1) / BK;
    const int tile_end = min(tiles_total, tile_beg + a.slab_tiles);

    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t y_reg[LD], x_reg[LD];
    auto load_row = [&](int t, int j) {
        const int m = t * BK + prow + j * RPP;
        const bool mv = t < tile_end && m < a.M;
        uint32_t xo_b;
        bool xv = mv && kc_ok;
        if (LIN) {
            xo_b = ((uint32_t)m * (uint32_t)a.ldx + (uint32_t)xc) * 4u;
        } else {
            const uint32_t b = fdiv((uint32_t)m, a.div_howo);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
            const uint32_t yo = fdiv(rem, a.div_wo);
            const uint32_t xo = rem - yo * (uint32_t)a.Wo;
            const int ys = (int)yo * a.stride + dyo, xs = (int)xo * a.stride + dxo;
            xv = xv && (unsigned)ys < (unsigned)a.Hi && (unsigned)xs < (unsigned)a.Wi;
            xo_b = ((uint32_t)(((int)b * a.Hi + ys) * a.Wi + xs) * (uint32_t)a.ldx + (uint32_t)xc) * 4u;
        }
        const uint32_t yo_b = ((uint32_t)m * (uint32_t)a.ldy + (uint32_t)yn) * 4u;
        y_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)((mv && yn_ok) ? yo_b : OOB), 0, 0);
        x_reg[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)(xv ? xo_b : OOB), 0, 0);
    };
    auto split_store = [&](const u32x4_t v, bf16_t* plane0, int off) {
        const float x[4] = {__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3])};
        uint32_t h[2], m[2], l[2];
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            h[e] = pack_bf16x2(x[2 * e], x[2 * e + 1]);
            const float r0 = x[2 * e] - __uint_as_float(h[e] << 16), r1 = x[2 * e + 1] - __uint_as_float(h[e] & 0xffff0000u);
            m[e] = pack_bf16x2(r0, r1);
            const float s0 = r0 - __uint_as_float(m[e] << 16), s1 = r1 - __uint_as_float(m[e] & 0xffff0000u);
            l[e] = pack_bf16x2(s0, s1);
        }
        *reinterpret_cast<uint2*>(plane0 + off) = make_uint2(h[0], h[1]);
        *reinterpret_cast<uint2*>(plane0 + PL + off) = make_uint2(m[0], m[1]);
        *reinterpret_cast<uint2*>(plane0 + 2 * PL + off) = make_uint2(l[0], l[1]);
    };
    // conv_wgrad_kernel<bf16_t>'s image: 16-column sub-tiles [col / 16][k'][16], k rows with bits 2 / 3 swapped; this
    // thread's four columns are a quarter of a sub-tile row
    auto lds_off = [&](int j) {
        const int row = prow + j * RPP;
        const int prow_ = (row & 0x13) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3);
        return (vcol >> 2) * SUB + prow_ * 16 + (vcol & 3) * 4;
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < LD; ++j) load_row(tile_beg, j);
#pragma unroll
    for (int j = 0; j < LD; ++j) {
        split_store(y_reg[j], Ys, lds_off(j));
        split_store(x_reg[j], Xs, lds_off(j));
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < LD; ++j) load_row(tile_beg + 1, j);
    const int lr = lane & 15, lq = lane >> 4;
    const int p_lo = ((lr >> 2) | ((lq & 1) << 2) | ((lq >> 1) << 4)) * 16 + (lr & 3) * 4;
    const int p_hi = p_lo + 8 * 16;
    const bf16_t* ys = Ys + wn * 4 * SUB;
    const bf16_t* xs = Xs + wk * 4 * SUB;
    for (int t = tile_beg; t < tile_end; ++t) {
        mfma_bf16x8 af[3][NT], bfr[3][MT];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                const bf16x4 lo = lds_tr16_b64(ys + p * PL + i * SUB + p_lo);
                const bf16x4 hi = lds_tr16_b64(ys + p * PL + i * SUB + p_hi);
                bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                af[p][i] = __builtin_bit_cast(mfma_bf16x8, v);
            }
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const bf16x4 lo = lds_tr16_b64(xs + p * PL + j * SUB + p_lo);
                const bf16x4 hi = lds_tr16_b64(xs + p * PL + j * SUB + p_hi);
                bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[p][j] = __builtin_bit_cast(mfma_bf16x8, v);
            }
        }
        __syncthreads();
        auto mm = [&](int py, int px) {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < MT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[py][i], bfr[px][j], acc[i][j], 0, 0, 0);
        };
        // each staging register is refilled (tile t + 2) as soon as its split has been stored: the load has the rest of this
        // step and the next step's fragment reads to land
        mm(2, 0); split_store(y_reg[0], Ys, lds_off(0)); split_store(x_reg[0], Xs, lds_off(0)); load_row(t + 2, 0);
        mm(0, 2); split_store(y_reg[1], Ys, lds_off(1)); split_store(x_reg[1], Xs, lds_off(1)); load_row(t + 2, 1);
        mm(1, 1); split_store(y_reg[2], Ys, lds_off(2)); split_store(x_reg[2], Xs, lds_off(2)); load_row(t + 2, 2);
        mm(1, 0); split_store(y_reg[3], Ys, lds_off(3)); split_store(x_reg[3], Xs, lds_off(3)); load_row(t + 2, 3);
        mm(0, 1);
        mm(0, 0);
#pragma unroll
        for (int q = 0; q < 64; ++q) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, LIN ? 3 : 4, 0);
            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            if (q % 8 == 7) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
        __syncthreads();
    }
    // acc[i][j][q] = dw[n = n0 + wn*64 + i*16 + lq*4 + q][kc = kc0 + wk*64 + j*16 + lr]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + i * 16 + lq * 4 + q;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int k = kc0 + wk * 64 + j * 16 + lr;
                if (k < a.Ktot) {
                    if (a.ws != nullptr)
                        a.ws[((int64_t)split * a.N + n) * a.Ktot + k] = acc[i][j][q];
                    else
                        atomicAdd(a.dw + (int64_t)n * a.Ktot + k, acc[i][j][q]);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// fp32 weight gradient with the products on the fp16 matrix cores: both operands as two fp16 planes of the scaled tensors
// (dml_h2_split; DmlWgradDesc.x_planes / dy_planes).  conv_wgrad_kernel<bf16_t>'s 128 x 128 tile, sub-tiled LDS image and
// ds_read_b64_tr_b16 fragments, with four operand planes per stage and three MFMA groups per 32-pixel K step
// (dy_l x_h, dy_h x_l, dy_h x_h): 48 MFMAs per wave against 16 bytes x 8 loads per thread -- no VALU split in the loop, the
// planes were written once by the producer of the tensor.  Double-buffered LDS (68 KB: two workgroups per CU), loads of step
// t + 1 in registers while step t computes.  Same split-K / workspace scheme as the other weight-gradient kernels.
// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------
// Weight gradient on fp16 planes, wave-specialised (the structure of conv_ws_kernel): three loader waves bring a K step -- 32
// pixels of dy [32][128] and of the gathered x [32][256], hi and lo plane each, 48 KB -- into a 3-stage LDS ring with LDS-DMA
// (whole 256 / 512-byte pixel rows per piece: the operands are pixel-major in memory), four consumer waves (2 x 2, wave tile
// 64 n x 128 kc = 32 accumulator fragments) read transposed fragments (ds_read_b64_tr_b16 on the row-major image, 16-byte chunk
// index XOR 2 (row & 7): conflict-free) and issue 96 MFMAs per K step.  Persistent workgroups walk the (output tile, pixel slab)
// list; slabs go to the workspace and wgrad_reduce_kernel folds them as for the other kernels.
// ------------------------------------------------------------------------------------------------
constexpr int WGW_NST = 3, WGW_NLD = 3;
constexpr int WGW_TN = 128, WGW_TK = 256;
constexpr int WGW_YROW = WGW_TN * 2, WGW_XROW = WGW_TK * 2;                  // bytes per pixel row and plane
constexpr int WGW_YPL = 32 * WGW_YROW, WGW_XPL = 32 * WGW_XROW;              // bytes per plane and stage
constexpr int WGW_SB = 2 * WGW_YPL + 2 * WGW_XPL;                            // [dy hi | dy lo | x hi | x lo] = 48 KB

template <bool LIN, int LW>
__device__ __forceinline__ void wgrad_ws_loader(const WgradArgs& a, const uint32_t x_bytes, const uint32_t dy_bytes, char* smem,
                                                uint32_t* ready, uint32_t* consumed, const int lane, const int nitems, const int ntile) {
    constexpr int NLD = WGW_NLD, NST = WGW_NST, NG = 24, MYG = NG / NLD;         // row groups: 8 of dy (4 rows), 16 of x (2 rows)
    constexpr uint32_t OOB = 0x80000000u;
    const u32x4_ws rs_x = ws_make_rsrc(a.x_planes, x_bytes + a.x_plane_bytes), rs_y = ws_make_rsrc(a.dy_planes, dy_bytes + a.dy_plane_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int tiles_total = (a.M + BK - 1) / BK;
    uint32_t g = 0, pub = 0;
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += (int)gridDim.x) {      // (neighbouring items -- the tiles of one pixel slab -- on one XCD: they share dy and x rows in its L2)
        const int split = item / ntile, tile_id = item - split * ntile;
        const int blk_n = tile_id % a.nblk_n, blk_k = tile_id / a.nblk_n;
        const int n0 = blk_n * WGW_TN, kc0 = blk_k * WGW_TK;
        const int tile_beg = split * a.slab_tiles, tile_end = min(tiles_total, tile_beg + a.slab_tiles);
        const int m_end = min(a.M, tile_end * BK);
        int row[MYG], colb[MYG], dyo[MYG], dxo[MYG];
        bool cok[MYG];
#pragma unroll
        for (int q = 0; q < MYG; ++q) {
            const int gi = q * NLD + LW;
            if (gi < 8) {
                row[q] = gi * 4 + (lane >> 4);
                const int gch = (lane & 15) ^ (2 * (row[q] & 7));
                const int n = n0 + gch * 8;
                cok[q] = n < a.N;
                colb[q] = n * 2;
                dyo[q] = dxo[q] = 0;
            } else {
                row[q] = (gi - 8) * 2 + (lane >> 5);
                const int gch = (lane & 31) ^ (2 * (row[q] & 7));
                const int kc = kc0 + gch * 8;
                cok[q] = kc < a.Ktot;
                const uint32_t tap = fdiv((uint32_t)(cok[q] ? kc : 0), a.div_c);
                const int xc = (cok[q] ? kc : 0) - (int)tap * a.C;
                const int tr = (int)tap / a.S, ts = (int)tap - tr * a.S;
                dyo[q] = tr * a.dil - a.pad;
                dxo[q] = ts * a.dil - a.pad;
                colb[q] = xc * 2;
            }
        }
        for (int t = tile_beg; t < tile_end; ++t) {
            if (g >= (uint32_t)NST) {
                const uint32_t need = g - NST + 1;
                bool drained = false;
                for (;;) {
                    uint32_t mn = ws_ld(consumed);
#pragma unroll
                    for (int w = 1; w < 4; ++w) mn = min(mn, ws_ld(consumed + w));
                    if (mn >= need) break;
                    if (!drained) {
                        wait_vmcnt<0>();
                        ws_st(ready + LW, g);
                        pub = g;
                        drained = true;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            }
            const uint32_t sbase = lds0 + (g % NST) * WGW_SB;
#pragma unroll
            for (int q = 0; q < MYG; ++q) {
                const int gi = q * NLD + LW;
                const int m = t * BK + row[q];
                bool ok = cok[q] && m < m_end;
                uint32_t voff;
                if (gi < 8) {
                    voff = (uint32_t)m * (uint32_t)(a.ldy * 2) + (uint32_t)colb[q];
                    ws_dma16(rs_y, sbase + gi * 1024, ok ? voff : OOB, 0u);
                    ws_dma16(rs_y, sbase + WGW_YPL + gi * 1024, ok ? voff : OOB, a.dy_plane_bytes);
                } else {
                    if (LIN) {
                        voff = (uint32_t)m * (uint32_t)(a.ldx * 2) + (uint32_t)colb[q];
                    } else {
                        const uint32_t mm = ok ? (uint32_t)m : 0u;
                        const uint32_t b = fdiv(mm, a.div_howo);
                        const uint32_t rem = mm - b * (uint32_t)(a.Ho * a.Wo);
                        const uint32_t yo = fdiv(rem, a.div_wo);
                        const uint32_t xo = rem - yo * (uint32_t)a.Wo;
                        const int ys = (int)yo * a.stride + dyo[q], xs = (int)xo * a.stride + dxo[q];
                        ok = ok && (unsigned)ys < (unsigned)a.Hi && (unsigned)xs < (unsigned)a.Wi;
                        voff = (uint32_t)(((int)b * a.Hi + ys) * a.Wi + xs) * (uint32_t)(a.ldx * 2) + (uint32_t)colb[q];
                    }
                    const int xi = gi - 8;
                    ws_dma16(rs_x, sbase + 2 * WGW_YPL + xi * 1024, ok ? voff : OOB, 0u);
                    ws_dma16(rs_x, sbase + 2 * WGW_YPL + WGW_XPL + xi * 1024, ok ? voff : OOB, a.x_plane_bytes);
                }
            }
            ++g;
            if (g > 1u + pub) {
                wait_vmcnt<2 * MYG>();             // one stage of this wave's pieces in flight behind the one it publishes
                pub = g - 1;
                ws_st(ready + LW, pub);
            }
        }
    }
    wait_vmcnt<0>();
    ws_st(ready + LW, g);
}

template <bool LIN>
__global__ __launch_bounds__((4 + WGW_NLD) * 64) void conv_wgrad_ws_kernel(const WgradArgs a, const uint32_t x_bytes, const uint32_t dy_bytes,
                                                                          const int nitems) {
    constexpr int NST = WGW_NST, NLD = WGW_NLD, NT = 4, MT = 8;
    __shared__ __attribute__((aligned(1024))) char smem[NST * WGW_SB + 64];
    uint32_t* const ready = reinterpret_cast<uint32_t*>(smem + NST * WGW_SB);
    uint32_t* const consumed = ready + 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 16) reinterpret_cast<uint32_t*>(smem + NST * WGW_SB)[tid] = 0;
    __syncthreads();
    const int ntile = a.nblk_n * a.nblk_k;
    if (wave >= 4) {
        const int lw = wave - 4;
        if (lw == 0) wgrad_ws_loader<LIN, 0>(a, x_bytes, dy_bytes, smem, ready, consumed, lane, nitems, ntile);
        if (lw == 1) wgrad_ws_loader<LIN, 1>(a, x_bytes, dy_bytes, smem, ready, consumed, lane, nitems, ntile);
        if (lw == 2) wgrad_ws_loader<LIN, 2>(a, x_bytes, dy_bytes, smem, ready, consumed, lane, nitems, ntile);
        return;
    }
    // ---------------------------------------------------------------------- consumer
    __builtin_amdgcn_s_setprio(3);
    const int wn = wave >> 1, wk = wave & 1;
    const int lr = lane & 15, lq = lane >> 4;
    const int tiles_total = (a.M + BK - 1) / BK;
    uint32_t g = 0, rflag = 0;
    auto wait_ready = [&](const uint32_t need) {
        while (rflag < need) {
            __builtin_amdgcn_s_sleep(1);
            uint32_t v = ws_ld(ready);
#pragma unroll
            for (int w = 1; w < NLD; ++w) v = min(v, ws_ld(ready + w));
            rflag = v;
        }
        asm volatile("" ::: "memory");
    };
    // transposed fragment = two ds_read_b64_tr_b16 (pixel rows r, r + 8 of the lane's row group) of four channels
    const int row_lo = (lr >> 2) | ((lq & 1) << 2) | ((lq >> 1) << 4);
    const int sw = 2 * (row_lo & 7), sub = ((lr & 3) & 1) * 8;
    int y_off[NT], x_off[MT];
#pragma unroll
    for (int i = 0; i < NT; ++i) y_off[i] = row_lo * WGW_YROW + (((wn * 8 + i * 2 + ((lr & 3) >> 1)) ^ sw) << 4) + sub;
#pragma unroll
    for (int j = 0; j < MT; ++j) x_off[j] = 2 * WGW_YPL + row_lo * WGW_XROW + (((wk * 16 + j * 2 + ((lr & 3) >> 1)) ^ sw) << 4) + sub;
    auto frag = [&](const char* p, const int rowb) -> mfma_f16x8 {
        const bf16x4 lo = lds_tr16_b64(reinterpret_cast<const bf16_t*>(p));
        const bf16x4 hi = lds_tr16_b64(reinterpret_cast<const bf16_t*>(p + 8 * rowb));
        const bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(mfma_f16x8, v);
    };
    for (int item = xcd_remap(blockIdx.x, gridDim.x); item < nitems; item += (int)gridDim.x) {      // (neighbouring items -- the tiles of one pixel slab -- on one XCD: they share dy and x rows in its L2)
        const int split = item / ntile, tile_id = item - split * ntile;
        const int blk_n = tile_id % a.nblk_n, blk_k = tile_id / a.nblk_n;
        const int n0 = blk_n * WGW_TN, kc0 = blk_k * WGW_TK;
        const int tile_beg = split * a.slab_tiles, tile_end = min(tiles_total, tile_beg + a.slab_tiles);
        const int KT = tile_end - tile_beg;
        f32x4 acc[NT][MT];
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
        mfma_f16x8 yh[NT], yhn[NT], yl[NT], xh[2], xl[2];
        uint32_t pl[NLD];
        if (KT > 0) {
            wait_ready(g + 1);
            const char* sb = smem + (g % NST) * WGW_SB;
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                yh[i] = frag(sb + y_off[i], WGW_YROW);
                yl[i] = frag(sb + y_off[i] + WGW_YPL, WGW_YROW);
            }
            xh[0] = frag(sb + x_off[0], WGW_XROW);
            xl[0] = frag(sb + x_off[0] + WGW_XPL, WGW_XROW);
        }
        for (int kt = 0; kt < KT; ++kt) {
            const bool has_next = kt + 1 < KT;
            const char* sb = smem + (g % NST) * WGW_SB;
            const char* sn = has_next ? smem + ((g + 1) % NST) * WGW_SB : sb;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                // (round 5: the transposed fragment reads of the next column group sit BETWEEN this group's three MFMA quads, as in
                // conv_ws_kernel: a cluster of reads in front of twelve MFMAs outlasts the MFMA before it)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[j & 1], yh[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < MT) xh[(j + 1) & 1] = frag(sb + x_off[j + 1], WGW_XROW);
                else xh[0] = frag(sn + x_off[0], WGW_XROW);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xl[j & 1], yh[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < MT) {
                    xl[(j + 1) & 1] = frag(sb + x_off[j + 1] + WGW_XPL, WGW_XROW);
                } else {
                    asm volatile("" ::: "memory");
                    ws_st(consumed + wave, g + 1);      // every read of stage g has been issued
                    xl[0] = frag(sn + x_off[0] + WGW_XPL, WGW_XROW);
                }
                if (j == 1) {
#pragma unroll
                    for (int w = 0; w < NLD; ++w) pl[w] = ws_ld(ready + w);
                }
                if (j == 3) {
                    rflag = pl[0];
#pragma unroll
                    for (int w = 1; w < NLD; ++w) rflag = min(rflag, pl[w]);
                    if (has_next) wait_ready(g + 2);
                }
                if (j >= 4 && j < 4 + NT) yhn[j - 4] = frag(sn + y_off[j - 4], WGW_YROW);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < NT; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(xh[j & 1], yl[i], acc[i][j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) {
                yl[i] = frag(sn + y_off[i] + WGW_YPL, WGW_YROW);
                yh[i] = yhn[i];
            }
            __builtin_amdgcn_sched_barrier(0);
            ++g;
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
        const float un = a.x_unscale[0] * a.dy_unscale[0];
        // the x fragment is the MFMA's row operand: acc[i][j][q] = dw[n = n0 + wn*64 + i*16 + lr][kc = kc0 + wk*128 + j*16 + lq*4 + q],
        // four consecutive kc per lane -> 16-byte slab stores (Ktot % 8 == 0)
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int n = n0 + wn * 64 + i * 16 + lr;
            if (n >= a.N) continue;
            float* wrow = a.ws + ((int64_t)split * a.N + n) * a.Ktot;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int k = kc0 + wk * 128 + j * 16 + lq * 4;
                if (k < a.Ktot) *reinterpret_cast<f32x4*>(wrow + k) = acc[i][j] * un;
            }
        }
    }
}

template <bool LIN>      // LIN: 1x1, stride 1, no padding -- pixel row m of x is output row m
__global__ __launch_bounds__(NTHREADS) __attribute__((amdgpu_waves_per_eu(2))) void conv_wgrad_h2_kernel(
    const WgradArgs a, const uint32_t x_bytes, const uint32_t dy_bytes) {
    constexpr int TILE = 128, CV = TILE / 8, RPP = NTHREADS / CV, LD = BK / RPP;       // 16 vectors per row, 16 rows per pass, 2 loads
    constexpr int SUB = 528, PLN = (TILE / 16) * SUB;                                  // fp16 elements per operand plane and stage
    constexpr int MT = 4, NT = 4;
    constexpr uint32_t OOB = 0x80000000u;
    typedef _Float16 f16_t;
    __shared__ __attribute__((aligned(16))) f16_t smem[2 * 4 * PLN];                   // [buf][dy hi, dy lo, x hi, x lo]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 1, wk = wave & 1;
    const int ntile = a.nblk_n * a.nblk_k;
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, tile_id = logical - split * ntile;
    const int blk_n = tile_id % a.nblk_n, blk_k = tile_id / a.nblk_n;
    const int n0 = blk_n * TILE, kc0 = blk_k * TILE;
    // descriptors span both planes; the lo plane is reached through the scalar offset
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x_planes), 0, (int)(x_bytes + a.x_plane_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy_planes), 0, (int)(dy_bytes + a.dy_plane_bytes), 0x00020000);
    const int vcol = tid % CV, prow = tid / CV;
    const int kc = kc0 + vcol * 8;
    const bool kc_ok = kc < a.Ktot;
    const uint32_t tap = fdiv((uint32_t)(kc_ok ? kc : 0), a.div_c);
    const int xc = (kc_ok ? kc : 0) - (int)tap * a.C;
    const int tr = (int)tap / a.S, ts = (int)tap - tr * a.S;
    const int dyo = tr * a.dil - a.pad, dxo = ts * a.dil - a.pad;
    const int yn = n0 + vcol * 8;
    const bool yn_ok = yn < a.N;
    const int tile_beg = split * a.slab_tiles;
    const int tiles_total = (a.M + BK - 1) / BK;
    const int tile_end = min(tiles_total, tile_beg + a.slab_tiles);

    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    u32x4_t yh_r[LD], yl_r[LD], xh_r[LD], xl_r[LD];
    auto load_row = [&](int t, int j) {
        const int m = t * BK + prow + j * RPP;
        const bool mv = t < tile_end && m < a.M;
        uint32_t xo_b;
        bool xv = mv && kc_ok;
        if (LIN) {
            xo_b = ((uint32_t)m * (uint32_t)a.ldx + (uint32_t)xc) * 2u;
        } else {
            const uint32_t b = fdiv((uint32_t)m, a.div_howo);
            const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
            const uint32_t yo = fdiv(rem, a.div_wo);
            const uint32_t xo = rem - yo * (uint32_t)a.Wo;
            const int ys = (int)yo * a.stride + dyo, xs = (int)xo * a.stride + dxo;
            xv = xv && (unsigned)ys < (unsigned)a.Hi && (unsigned)xs < (unsigned)a.Wi;
            xo_b = ((uint32_t)(((int)b * a.Hi + ys) * a.Wi + xs) * (uint32_t)a.ldx + (uint32_t)xc) * 2u;
        }
        const uint32_t yo_b = ((uint32_t)m * (uint32_t)a.ldy + (uint32_t)yn) * 2u;
        const int yoff = (int)((mv && yn_ok) ? yo_b : OOB), xoff = (int)(xv ? xo_b : OOB);
        yh_r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, yoff, 0, 0);
        yl_r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_y, yoff, (int)a.dy_plane_bytes, 0);
        xh_r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff, 0, 0);
        xl_r[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_x, xoff, (int)a.x_plane_bytes, 0);
    };
    // conv_wgrad_kernel<bf16_t>'s image: 16-column sub-tiles [col / 16][k'][16], k rows with bits 2 / 3 swapped
    auto store_tiles = [&](int buf) {
        f16_t* base = smem + buf * 4 * PLN;
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            const int row = prow + j * RPP;
            const int prow_ = (row & 0x13) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3);
            const int off = (vcol >> 1) * SUB + prow_ * 16 + (vcol & 1) * 8;
            *reinterpret_cast<u32x4_t*>(base + off) = yh_r[j];
            *reinterpret_cast<u32x4_t*>(base + PLN + off) = yl_r[j];
            *reinterpret_cast<u32x4_t*>(base + 2 * PLN + off) = xh_r[j];
            *reinterpret_cast<u32x4_t*>(base + 3 * PLN + off) = xl_r[j];
        }
    };
    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (tile_beg < tile_end) {
#pragma unroll
        for (int j = 0; j < LD; ++j) load_row(tile_beg, j);
        store_tiles(0);
    }
    __syncthreads();
    const int lr = lane & 15, lq = lane >> 4;
    const int p_lo = ((lr >> 2) | ((lq & 1) << 2) | ((lq >> 1) << 4)) * 16 + (lr & 3) * 4;
    const int p_hi = p_lo + 8 * 16;
    auto frag = [&](const f16_t* p) -> mfma_f16x8 {
        const bf16x4 lo = lds_tr16_b64(reinterpret_cast<const bf16_t*>(p + p_lo));
        const bf16x4 hi = lds_tr16_b64(reinterpret_cast<const bf16_t*>(p + p_hi));
        const bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(mfma_f16x8, v);
    };
    int cur = 0;
    for (int t = tile_beg; t < tile_end; ++t) {
        const bool has_next = t + 1 < tile_end;
        if (has_next) {
#pragma unroll
            for (int j = 0; j < LD; ++j) load_row(t + 1, j);
        }
        const f16_t* sb = smem + cur * 4 * PLN;
        mfma_f16x8 yh[NT], yl[NT], xh[MT], xl[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            yh[i] = frag(sb + (wn * 4 + i) * SUB);
            yl[i] = frag(sb + PLN + (wn * 4 + i) * SUB);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            xh[j] = frag(sb + 2 * PLN + (wk * 4 + j) * SUB);
            xl[j] = frag(sb + 3 * PLN + (wk * 4 + j) * SUB);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yl[i], xh[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh[i], xl[j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(yh[i], xh[j], acc[i][j], 0, 0, 0);
        if (has_next) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    const float un = a.x_unscale[0] * a.dy_unscale[0];      // exact powers of two
    // acc[i][j][q] = dw[n = n0 + wn*64 + i*16 + lq*4 + q][kc = kc0 + wk*64 + j*16 + lr]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = n0 + wn * 64 + i * 16 + lq * 4 + q;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int k = kc0 + wk * 64 + j * 16 + lr;
                if (k < a.Ktot) {
                    if (a.ws != nullptr)
                        a.ws[((int64_t)split * a.N + n) * a.Ktot + k] = acc[i][j][q] * un;
                    else
                        atomicAdd(a.dw + (int64_t)n * a.Ktot + k, acc[i][j][q] * un);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// bf16 weight gradient for N <= 64 output channels (layer1's 3x3 and 1x1 convolutions, the stem, the 48- and
// 16-channel head convolutions): tile 64 (n) x 256 (kc), 4 waves side by side along kc (wave tile 64 x 64).  In the
// 128 x 128 kernel half of every MFMA row block is padding for these layers and two of the four waves idle; here all
// four compute.  Same LDS image (16-column sub-tiles read with ds_read_b64_tr_b16), same split-K / workspace scheme.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NTHREADS) void conv_wgrad_n64_kernel(const WgradArgs a) {
    typedef bf16_t T;
    constexpr int TN = 64, TK = 256, VEC = 8;
    constexpr int SUB = 528;                       // elements per 16-column sub-tile (32 k rows x 16 + pad)
    constexpr int YS = (TN / 16) * SUB, XS = (TK / 16) * SUB;
    constexpr int NT = 4, MT = 4;

    __shared__ __attribute__((aligned(16))) T smem[2 * (YS + XS)];
    auto Ys = [&](int buf) -> T* { return smem + buf * (YS + XS); };
    auto Xs = [&](int buf) -> T* { return smem + buf * (YS + XS) + YS; };

    const int tid = threadIdx.x, lane = tid & 63, wk = tid >> 6;
    const int ntile = a.nblk_k;                    // one n tile
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    const int split = logical / ntile, blk_k = logical - split * ntile;
    const int kc0 = blk_k * TK;

    const T* __restrict__ X = static_cast<const T*>(a.x);
    const T* __restrict__ DY = static_cast<const T*>(a.dy);

    // dy: 32 rows x 8 vectors = one 16-byte load per thread; x: 32 rows x 32 vectors = four per thread
    const int yv = tid & 7, yrow = tid >> 3;
    const int yn = yv * VEC;
    const bool yn_ok = yn < a.N;
    const int xv = tid & 31, xrow0 = tid >> 5;      // rows xrow0 + 8 j
    const int kc = kc0 + xv * VEC;
    const bool kc_ok = kc < a.Ktot;
    const uint32_t tap = fdiv((uint32_t)(kc_ok ? kc : 0), a.div_c);
    const int xc = (kc_ok ? kc : 0) - (int)tap * a.C;
    const int tr = (int)tap / a.S, ts = (int)tap - tr * a.S;
    const int dyo = tr * a.dil - a.pad, dxo = ts * a.dil - a.pad;

    const int tile_beg = split * a.slab_tiles;
    const int tiles_total = (a.M + BK - 1) / BK;
    const int tile_end = min(tiles_total, tile_beg + a.slab_tiles);
    const bool lin1x1 = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0;

    uint4 y_reg, x_reg[4];
    auto load_tiles = [&](int t) {
        {
            const int m = t * BK + yrow;
            y_reg = make_uint4(0, 0, 0, 0);
            if (m < a.M && yn_ok) y_reg = *reinterpret_cast<const uint4*>(DY + (int64_t)m * a.ldy + yn);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int m = t * BK + xrow0 + j * 8;
            uint4 xv4 = make_uint4(0, 0, 0, 0);
            if (m < a.M && kc_ok) {
                if (lin1x1) {
                    xv4 = *reinterpret_cast<const uint4*>(X + (int64_t)m * a.ldx + xc);
                } else {
                    const uint32_t b = fdiv((uint32_t)m, a.div_howo);
                    const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
                    const uint32_t yo = fdiv(rem, a.div_wo);
                    const uint32_t xo = rem - yo * (uint32_t)a.Wo;
                    const int ys = (int)yo * a.stride + dyo, xs = (int)xo * a.stride + dxo;
                    if ((unsigned)ys < (unsigned)a.Hi && (unsigned)xs < (unsigned)a.Wi)
                        xv4 = *reinterpret_cast<const uint4*>(X + ((int64_t)((int)b * a.Hi + ys) * a.Wi + xs) * a.ldx + xc);
                }
            }
            x_reg[j] = xv4;
        }
    };
    // sub-tile image [col/16][k'][16], k rows stored with bits 2/3 swapped (conflict-free for the transpose reads and
    // for these 16-byte writes) -- as in conv_wgrad_kernel
    auto row_pos = [](int row) { return (row & 0x13) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3); };
    auto store_tiles = [&](int buf) {
        *reinterpret_cast<uint4*>(Ys(buf) + (yv >> 1) * SUB + row_pos(yrow) * 16 + (yv & 1) * 8) = y_reg;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<uint4*>(Xs(buf) + (xv >> 1) * SUB + row_pos(xrow0 + j * 8) * 16 + (xv & 1) * 8) = x_reg[j];
    };

    f32x4 acc[NT][MT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if (tile_beg < tile_end) {
        load_tiles(tile_beg);
        store_tiles(0);
    }
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    const int p_lo = ((lr >> 2) | ((lq & 1) << 2) | ((lq >> 1) << 4)) * 16 + (lr & 3) * 4;
    const int p_hi = p_lo + 8 * 16;
    int cur = 0;
    for (int t = tile_beg; t < tile_end; ++t) {
        const bool has_next = t + 1 < tile_end;
        if (has_next) load_tiles(t + 1);
        const T* ys = Ys(cur);
        const T* xs = Xs(cur) + wk * 4 * SUB;
        mfma_bf16x8 af[NT], bfr[MT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const bf16x4 lo = lds_tr16_b64(ys + i * SUB + p_lo);
            const bf16x4 hi = lds_tr16_b64(ys + i * SUB + p_hi);
            bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            af[i] = __builtin_bit_cast(mfma_bf16x8, v);
        }
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const bf16x4 lo = lds_tr16_b64(xs + j * SUB + p_lo);
            const bf16x4 hi = lds_tr16_b64(xs + j * SUB + p_hi);
            bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            bfr[j] = __builtin_bit_cast(mfma_bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
        if (has_next) store_tiles(cur ^ 1);
        __syncthreads();
        cur ^= 1;
    }
    // acc[i][j][q] = dw[n = i*16 + lq*4 + q][kc = kc0 + wk*64 + j*16 + lr]
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int n = i * 16 + lq * 4 + q;
            if (n >= a.N) continue;
#pragma unroll
            for (int j = 0; j < MT; ++j) {
                const int k = kc0 + wk * 64 + j * 16 + lr;
                if (k < a.Ktot) {
                    if (a.ws != nullptr)
                        a.ws[((int64_t)split * a.N + n) * a.Ktot + k] = acc[i][j][q];
                    else
                        atomicAdd(a.dw + (int64_t)n * a.Ktot + k, acc[i][j][q]);
                }
            }
        }
}

// ------------------------------------------------------------------------------------------------
// bf16 weight gradient, large tile: 256 (n) x 256 (kc) per 512-thread workgroup, 8 waves as 2 (n) x 4 (kc), wave
// tile 128 x 64 = 32 MFMAs per 32-pixel K step.  Why: the 128 x 128 kernel moves 16 KB through the CU's vector
// memory pipeline (64 B/clk) and 48 KB through LDS per 64 MFMAs -- both as busy as the matrix cores, so none of
// them gets past ~50 %; this tile halves both per MFMA.  One workgroup per CU; split-K over pixel slabs sized so
// that the grid is a multiple of the CU count.  Same LDS image as conv_wgrad_kernel (16-column sub-tiles read with
// ds_read_b64_tr_b16); the x fragment is the MFMA row operand here, so a lane ends up with 4 consecutive kc of
// one n and the split-K slab is written with 16-byte stores.  Operand tiles are written to LDS one K step ahead,
// right after the barrier (one register set, loads for step t+2 re-issued immediately).
// Addressing: buffer loads, 32-bit offsets advanced incrementally; tile edges and padding are out-of-range offsets.
// ------------------------------------------------------------------------------------------------
constexpr int WB_THREADS = 512;
constexpr int WB_SUB = 528;                 // elements per 16-column sub-tile (32 k rows x 16 + pad)
constexpr int WB_OP = 16 * WB_SUB;          // one operand, one stage (256 columns)

// DEPTH = 2: two K steps per barrier -- two register sets and 2 x 2 LDS buffers, so that a load has two K steps to land
// (twice the bytes in flight per CU: 64 KB; the one-workgroup-per-CU kernel is bound by what its 512 threads keep
// outstanding) and the barrier is paid once per 64 MFMAs of a wave.
// `logical`: this workgroup's index among the job's nblk_n * nblk_k * splits workgroups (tiles of one pixel slab adjacent)
template <int DEPTH>
__device__ __forceinline__ void wgrad_big_body(const WgradArgs& a, const uint32_t x_bytes, const uint32_t dy_bytes,
                                               const int logical, bf16_t* smem) {
    typedef bf16_t T;
    constexpr int TILE = 256, VEC = 8, RPP = 16, LD = 2;
    constexpr int NT = 8, MT = 4;                 // wave tile: 8 n-tiles x 4 kc-tiles of 16
    constexpr uint32_t OOB = 0x80000000u;

    auto Ys = [&](int buf, int h) -> T* { return smem + (buf * DEPTH + h) * 2 * WB_OP; };
    auto Xs = [&](int buf, int h) -> T* { return smem + (buf * DEPTH + h) * 2 * WB_OP + WB_OP; };

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wn = wave >> 2, wk = wave & 3;
    const int ntile = a.nblk_n * a.nblk_k;
    const int split = logical / ntile, tile_id = logical - split * ntile;
    const int blk_n = tile_id % a.nblk_n, blk_k = tile_id / a.nblk_n;
    const int n0 = blk_n * TILE, kc0 = blk_k * TILE;

    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.x), 0, (int)x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_y = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(a.dy), 0, (int)dy_bytes, 0x00020000);

    const int vcol = tid & 31, prow = tid >> 5;
    const int kc = kc0 + vcol * VEC;
    const bool kc_ok = kc < a.Ktot;
    const uint32_t tap = fdiv((uint32_t)(kc_ok ? kc : 0), a.div_c);
    const int xc = (kc_ok ? kc : 0) - (int)tap * a.C;
    const int tr = (int)tap / a.S, ts = (int)tap - tr * a.S;
    const int dyo = tr * a.dil - a.pad, dxo = ts * a.dil - a.pad;
    const int yn = n0 + vcol * VEC;
    const bool yn_ok = yn < a.N;

    const int tile_beg = split * a.slab_tiles;
    const int tiles_total = (a.M + BK - 1) / BK;
    const int tile_end = min(tiles_total, tile_beg + a.slab_tiles);
    const bool lin1x1 = a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0;

    // incremental per-row state: rows m = t*32 + prow + 16 j.  Rows past M need no test: their offsets fall
    // beyond the descriptors (dy: m*ldy >= M*ldy; x: image index >= B).
    uint32_t yoff[LD], xoff[LD];
    int px_b[LD], px_y[LD], px_x[LD];             // b*Hi, yo, xo of the general gather
#pragma unroll
    for (int j = 0; j < LD; ++j) {
        const int m = tile_beg * BK + prow + j * RPP;
        yoff[j] = yn_ok ? (uint32_t)((m * a.ldy + yn) * 2) : OOB;
        xoff[j] = kc_ok ? (uint32_t)((m * a.ldx + xc) * 2) : OOB;          // 1x1 stride 1: source pixel = m
        const uint32_t b = fdiv((uint32_t)m, a.div_howo);
        const uint32_t rem = (uint32_t)m - b * (uint32_t)(a.Ho * a.Wo);
        const uint32_t yo = fdiv(rem, a.div_wo);
        px_b[j] = (int)b * a.Hi;
        px_y[j] = (int)yo;
        px_x[j] = (int)(rem - yo * (uint32_t)a.Wo);
    }
    const uint32_t ystep = (uint32_t)(BK * a.ldy * 2), xstep = (uint32_t)(BK * a.ldx * 2);

    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    uint4 y_reg[DEPTH][LD], x_reg[DEPTH][LD];
    // loads the NEXT tile in sequence into register set h (call once per K step, in tile order)
    auto load_next = [&](const int h) {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            const u32x4_t yv = __builtin_amdgcn_raw_buffer_load_b128(rs_y, (int)yoff[j], 0, 0);
            yoff[j] += ystep;
            uint32_t vo;
            if (lin1x1) {
                vo = xoff[j];
                xoff[j] += xstep;
            } else {
                const int ys = px_y[j] * a.stride + dyo, xs = px_x[j] * a.stride + dxo;
                const bool ok = kc_ok && ((unsigned)ys < (unsigned)a.Hi) && ((unsigned)xs < (unsigned)a.Wi);
                const uint32_t pix = __umul24((uint32_t)(px_b[j] + ys), (uint32_t)a.Wi) + (uint32_t)xs;
                vo = ok ? (__umul24(pix, (uint32_t)(a.ldx * 2)) + (uint32_t)(xc * 2)) : OOB;
                px_x[j] += BK;
                while (px_x[j] >= a.Wo) {
                    px_x[j] -= a.Wo;
                    if (++px_y[j] == a.Ho) {
                        px_y[j] = 0;
                        px_b[j] += a.Hi;
                    }
                }
            }
            const u32x4_t xv = __builtin_amdgcn_raw_buffer_load_b128(rs_x, (int)vo, 0, 0);
            y_reg[h][j] = make_uint4(yv.x, yv.y, yv.z, yv.w);
            x_reg[h][j] = make_uint4(xv.x, xv.y, xv.z, xv.w);
        }
    };
    auto store_tiles = [&](int buf, const int h) {
#pragma unroll
        for (int j = 0; j < LD; ++j) {
            const int row = prow + j * RPP;
            // 16-column sub-tiles [col/16][k'][16] (1056-byte pitch), k rows stored with bits 2/3 swapped:
            // conflict-free for ds_read_b64_tr_b16 and for these 16-byte writes
            const int prow_ = (row & 0x13) | (((row >> 3) & 1) << 2) | (((row >> 2) & 1) << 3);
            const int off = (vcol >> 1) * WB_SUB + prow_ * 16 + (vcol & 1) * 8;
            *reinterpret_cast<uint4*>(Ys(buf, h) + off) = y_reg[h][j];
            *reinterpret_cast<uint4*>(Xs(buf, h) + off) = x_reg[h][j];
        }
    };

    f32x4 acc[MT][NT];
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) acc[j][i] = (f32x4){0.f, 0.f, 0.f, 0.f};

#pragma unroll
    for (int h = 0; h < DEPTH; ++h)
        if (tile_beg + h < tile_end) load_next(h);
#pragma unroll
    for (int h = 0; h < DEPTH; ++h)
        if (tile_beg + h < tile_end) store_tiles(0, h);
#pragma unroll
    for (int h = 0; h < DEPTH; ++h)
        if (tile_beg + DEPTH + h < tile_end) load_next(h);
    __syncthreads();

    const int lr = lane & 15, lq = lane >> 4;
    // k rows 8*lq + (lr>>2) (+8 for the second read) at their swapped positions inside a sub-tile
    const int p_lo = ((lr >> 2) | ((lq & 1) << 2) | ((lq >> 1) << 4)) * 16 + (lr & 3) * 4;
    const int p_hi = p_lo + 8 * 16;
    int cur = 0;
    for (int t = tile_beg; t < tile_end; t += DEPTH) {
#pragma unroll
      for (int h = 0; h < DEPTH; ++h) {
        if (DEPTH > 1 && t + h >= tile_end) break;      // workgroup-uniform
        const T* ys = Ys(cur, h) + wn * 8 * WB_SUB;
        const T* xs = Xs(cur, h) + wk * 4 * WB_SUB;
        mfma_bf16x8 yf[NT], xf[MT];
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const bf16x4 lo = lds_tr16_b64(xs + j * WB_SUB + p_lo);
            const bf16x4 hi = lds_tr16_b64(xs + j * WB_SUB + p_hi);
            bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            xf[j] = __builtin_bit_cast(mfma_bf16x8, v);
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const bf16x4 lo = lds_tr16_b64(ys + i * WB_SUB + p_lo);
            const bf16x4 hi = lds_tr16_b64(ys + i * WB_SUB + p_hi);
            bf16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            yf[i] = __builtin_bit_cast(mfma_bf16x8, v);
        }
        if (t + h + DEPTH < tile_end) {
            store_tiles(cur ^ 1, h);                            // tile t+h+DEPTH (in registers since the previous round)
            if (t + h + 2 * DEPTH < tile_end) load_next(h);     // tile t+h+2*DEPTH
        }
#pragma unroll
        for (int j = 0; j < MT; ++j)
#pragma unroll
            for (int i = 0; i < NT; ++i)
                acc[j][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[j], yf[i], acc[j][i], 0, 0, 0);
      }
        __syncthreads();
        cur ^= 1;
    }
#pragma unroll
    for (int j = 0; j < MT; ++j)
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[j][i]));

    // acc[j][i][q] = dw[n = n0 + wn*128 + i*16 + lr][kc = kc0 + wk*64 + j*16 + lq*4 + q]
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int n = n0 + wn * 128 + i * 16 + lr;
        if (n >= a.N) continue;
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int k = kc0 + wk * 64 + j * 16 + lq * 4;
            if (k >= a.Ktot) continue;            // Ktot is a multiple of 8: whole vectors
            if (a.ws != nullptr) {
                *reinterpret_cast<float4*>(a.ws + ((int64_t)split * a.N + n) * a.Ktot + k) =
                    make_float4(acc[j][i][0], acc[j][i][1], acc[j][i][2], acc[j][i][3]);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) atomicAdd(a.dw + (int64_t)n * a.Ktot + k + q, acc[j][i][q]);
            }
        }
    }
}

template <int DEPTH>
__global__ __launch_bounds__(WB_THREADS) void conv_wgrad_big_kernel(const WgradArgs a, const uint32_t x_bytes,
                                                                    const uint32_t dy_bytes) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * DEPTH * 2 * WB_OP];
    wgrad_big_body<DEPTH>(a, x_bytes, dy_bytes, xcd_remap(blockIdx.x, gridDim.x), smem);
}

// Several weight gradients in ONE launch.  A single 48 x 48 layer has 4-9 output tiles of 256 x 256, so filling 256 CUs
// takes ~28 pixel slabs per tile and every slab is a 256 KB fp32 partial that is written and read again by the fold
// kernel: 64 MB + 64 MB per layer whatever its size (PMC: 11.3 GB per step for a 0.24 GB result).  Weight gradients
// only feed the optimizer, so the plan launches the jobs of consecutive layers together: the same 256 workgroups then
// cover several layers with a few slabs each, and the slab traffic (and the launch count) drops by the group size.
constexpr int WG_MAX_JOBS = 12;
struct WgradGroup {
    int njobs;
    int start[WG_MAX_JOBS + 1];            // first workgroup of each job in the launch
    uint32_t xb[WG_MAX_JOBS], yb[WG_MAX_JOBS];
    WgradArgs job[WG_MAX_JOBS];
};
template <int DEPTH>
__global__ __launch_bounds__(WB_THREADS) void conv_wgrad_group_kernel(const WgradGroup g) {
    __shared__ __attribute__((aligned(16))) bf16_t smem[2 * DEPTH * 2 * WB_OP];
    const int logical = xcd_remap(blockIdx.x, gridDim.x);
    int j = 0;
    while (j + 1 < g.njobs && logical >= g.start[j + 1]) ++j;
    wgrad_big_body<DEPTH>(g.job[j], g.xb[j], g.yb[j], logical - g.start[j], smem);
}

// dw[n][rs][c < Cm] += sum_s ws[s * slab_stride][n][rs][c]   (fp32 atomics on the L2 are ~10x more expensive per byte than this)
// A small weight tensor with hundreds of splits stays parallel AND deterministic in two launches: with fold_only, blockIdx.y
// walks chunks of 16 splits and leaves each chunk's sum in the chunk's first slab (every thread reads and writes only its own
// element); the second launch then adds the chunk sums (slab_stride = 16) in order.
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(float* __restrict__ ws, float* __restrict__ dw, int splits, int64_t NRS,
                                                           int Cm, int Cp, int slab_stride, int fold_only) {
    const int64_t total = NRS * Cm;
    const int64_t plane = NRS * Cp;
    const int s0 = fold_only ? blockIdx.y * 16 : 0, s1 = fold_only ? min(splits, s0 + 16) : splits;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t src = i;
        if (Cm != Cp) {
            const int64_t t = i / Cm;
            src = t * Cp + (i - t * Cm);
        }
        float acc = 0.f;
        #pragma unroll 8
        for (int sidx = s0; sidx < s1; ++sidx) acc += ws[(int64_t)sidx * slab_stride * plane + src];
        if (fold_only) ws[(int64_t)s0 * plane + src] = acc;
        else dw[i] += acc;
    }
}

struct ReduceGroup {
    int njobs;
    const float* ws[WG_MAX_JOBS];
    float* dw[WG_MAX_JOBS];
    int splits[WG_MAX_JOBS], Cm[WG_MAX_JOBS], Cp[WG_MAX_JOBS];
    int64_t NRS[WG_MAX_JOBS];
};
// blockIdx.y = job: dw_j[n][rs][c < Cm] += sum_s ws_j[s][n][rs][c]
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const ReduceGroup g) {
    const int j = blockIdx.y;
    const float* __restrict__ ws = g.ws[j];
    float* __restrict__ dw = g.dw[j];
    const int Cm = g.Cm[j], Cp = g.Cp[j], splits = g.splits[j];
    const int64_t total = g.NRS[j] * Cm, plane = g.NRS[j] * Cp;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t src = i;
        if (Cm != Cp) {
            const int64_t t = i / Cm;
            src = t * Cp + (i - t * Cm);
        }
        float acc = 0.f;
#pragma unroll 4
        for (int sidx = 0; sidx < splits; ++sidx) acc += ws[sidx * plane + src];
        dw[i] += acc;
    }
}

// shared by dml_conv_wgrad (single job) and dml_conv_wgrad_group: may this job run on the 256 x 256-tile kernel?
static bool wgrad_big_eligible(const DmlWgradDesc* d) {
    if (!d || !d->x || !d->dy || !d->dw || d->dtype != DML_BF16) return false;
    if (d->C % 8 || d->ldx % 8 || d->N % 8 || d->ldy % 8) return false;
    const int64_t M = (int64_t)d->B * d->Ho * d->Wo;
    const int64_t Ktot = (int64_t)d->R * d->S * d->C;
    const int64_t tiles = (M + BK - 1) / BK;
    const int cm = d->Cm > 0 ? d->Cm : d->C;
    if (cm > d->C) return false;
    const int64_t plane = (int64_t)d->N * Ktot;
    const int64_t xb64 = (((int64_t)(d->B * d->Hi) * d->Wi - 1) * d->ldx + d->C) * 2;
    const int64_t yb64 = ((M - 1) * d->ldy + d->N) * 2;
    return plane >= 65536 && d->N % 256 == 0 && (d->N / 256) * ((Ktot + 255) / 256) >= 4 && tiles >= 16 &&
           xb64 < (1ll << 31) && yb64 < (1ll << 31) && (int64_t)(d->B + 1) * d->Hi * d->Wi < (1 << 24) && M < (1 << 24) &&
           d->ldx < (1 << 22) && d->ldy < (1 << 22);
}

static void fill_wgrad_args(WgradArgs& a, const DmlWgradDesc* d) {
    a.x = d->x; a.dy = d->dy; a.dw = d->dw;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.N = d->N; a.ldy = d->ldy;
    a.R = d->R; a.S = d->S; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.Ktot = d->R * d->S * d->C;
    a.div_wo = make_fastdiv((uint32_t)d->Wo);
    a.div_howo = make_fastdiv((uint32_t)(d->Ho * d->Wo));
    a.div_c = make_fastdiv((uint32_t)d->C);
}

template <typename T, int MODE>
int launch_conv(const ConvArgs& base, hipStream_t st) {
    ConvArgs a = base;
    const int64_t xb64 = (((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * (int64_t)sizeof(T);
    const int64_t wb64 = (int64_t)a.N * a.Ktot * (int64_t)sizeof(T);
    const bool small = xb64 < (1ll << 31) && wb64 < (1ll << 31);
    a.x_bytes = small ? (uint32_t)xb64 : 0u;
    a.w_bytes = small ? (uint32_t)wb64 : 0u;
    const bool aligned = (a.C % BK) == 0 && a.R * a.S <= 32 && small;
    a.nblk_m = (a.M + 127) / 128;
    // non-temporal epilogue stores for outputs that do not fit the L2 anyway (st16); smaller ones are better left there for
    // their consumer (1x1 1024 -> 256 at 48 x 48, 19 MB: 45.3 us plain, 46.9 non-temporal).  DML_CONV_NT: 0 never, 1 this
    // rule (default), 2 always
    constexpr int nt_mode = 1;
    // the partial statistics (64 B per wave instruction) follow only where they are many: beside non-temporal outputs,
    // 19 MB of plain partial stores bring the stalls back (192 x 192, 64 -> 256: 158 us, 126 with both non-temporal; all
    // plain 141), while 4.7 MB do better plain (48 x 48, 256 -> 1024: 41.1 against 45.4 us)
    const bool nt_y = nt_mode == 2 || (nt_mode == 1 && (int64_t)a.M * a.N * (int64_t)sizeof(T) >= (32ll << 20));
    const bool nt_p = nt_y && (int64_t)((a.M + 63) / 64) * a.N * 8 >= (8ll << 20);
    a.nt_out = (nt_y ? 1 : 0) | (nt_p ? 2 : 0);
    if constexpr (sizeof(T) == 2) {
        // LDS-DMA kernel: bf16, K tiles inside one tap, tensors addressable with a 31-bit byte offset.  With a 3-stage
        // ring (48 KB of LDS) and the 168-register budget three workgroups fit a CU like the register-staged kernel, and
        // it is the default for every eligible shape: whole train step 362.4 vs 356.4 images/s (three interleaved A/B
        // runs; a 4-stage ring on the largest grids only: 360.2).  In isolation the two kernels are within +-5 % of each
        // other (tools/bench_conv.py); the register-staged kernel stays for fp32, unaligned channel counts (stem, final
        // conv) and N <= 32.
        static const bool use_v1 = getenv("DML_CONV_V1") != nullptr;        // tuning / test switch: register-staged kernel
        const int64_t xb = ((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx * 2 + (int64_t)a.C * 2;
        const int64_t wb = (int64_t)a.N * a.Ktot * 2;
        if (!use_v1 && aligned && a.N > 32 && xb < (1ll << 31) && wb < (1ll << 31)) {
            // wave-specialised kernel (conv_ws_kernel): one persistent workgroup per CU on 144-row tiles, tile-major weights,
            // N a multiple of 128, enough tiles to fill the chip.  Same-box microbenchmarks against the ring kernel below
            // (profiles/r04_ws_probe_3.txt vs r04_ws_probe_1_shipped.txt): layer3 3x3 67.5 -> 43.6 us, 1x1 1024 -> 256
            // 37.6 -> 24.2, 1x1 256 -> 1024 38.4 -> 31.0, ASPP 3x3 385 -> 325-357, decoder 3x3 886 -> 825-870.
            if (conv_ws_eligible(a, xb, wb, MODE)) {
                constexpr int CUS = 256, NLD = 3;
                const bool wide = (a.N % 256) == 0;
                const int bm = wide ? 144 : 288;
                a.nblk_m = (a.M + bm - 1) / bm;
                a.nblk_n = a.N / (wide ? 256 : 128);
                const int ntiles = a.nblk_m * a.nblk_n;
                const int grid = ntiles < CUS ? ntiles : CUS;
                if (wide)
                    hipLaunchKernelGGL((conv_ws_kernel<1, 4, MODE, NLD>), dim3(grid), dim3((4 + NLD) * 64), 0, st, a, (uint32_t)xb,
                                       (uint32_t)wb);
                else
                    hipLaunchKernelGGL((conv_ws_kernel<2, 2, MODE, NLD>), dim3(grid), dim3((4 + NLD) * 64), 0, st, a, (uint32_t)xb,
                                       (uint32_t)wb);
                DML_LAUNCH_CHECK();
                return 0;
            }
            // grids that leave most of the chip idle (batch-1 inference, the anomaly model's down-scaled inputs): halve the
            // tile width to double the workgroup count; per-FLOP the 128 x 64 tile is 15-25 % slower, so only below 200
            // workgroups (1024 x 2048 bs 1: 235 -> 250 images/s, 5-scale open-set evaluation 92 -> 97.5 frames/s; the
            // training grids are all larger)
            const bool narrow = a.nblk_m * ((a.N + 127) / 128) < 200;
            // tiles of the partially filled last round, split along K (DmlConvDesc::tail_*): when the remainder is at most
            // half a round of the CUs, q = CUs / remainder parts per tile give every CU the same share
            a.tail_full = 0;
            a.tail_q = 1;
            if (a.tail_ws != nullptr && a.N > 64 && !narrow) {
                constexpr int CUS = 256;
                const int ntiles = a.nblk_m * ((a.N + 127) / 128), full = ntiles / CUS * CUS, rem = ntiles - full;
                if (full >= CUS && full <= 6 * CUS && rem > 0 && rem <= CUS / 2) {
                    int q = CUS / rem;
                    if (q > 8) q = 8;
                    const int KTall = a.Ktot / BK;
                    while (q > 1 && KTall / q < 6) --q;
                    if (q > 1 && (int64_t)rem * q * 128 * 128 <= base.tail_ws_elems_ && rem <= base.tail_cnt_len_) {
                        a.tail_full = full;
                        a.tail_q = q;
                    }
                }
            }
            // 256-row tiles pay where the K loop is long (3x3 convolutions over >= 256 channels: layer4 / ASPP / decoder),
            // with the K-split tail balancing their few tiles: in isolation ASPP 3x3 389 -> 343 us (1014 TFLOP/s), layer4
            // 3x3 202 -> 176 us, decoder 3x3 1003 -> 922 us.  On the short-K 1x1 layers three
            // workgroups of four waves hide latency better than two of eight (whole step -1.8 % with 256-row tiles
            // everywhere).  DML_CONV_BM256: 0 = never, 1 = this rule (default), 2 = every eligible layer,
            // >= 64 = the K threshold of the rule
            // In the plan (serial profile, r02 v3 -> v5): ASPP forward 1.14 -> 0.98 ms, ASPP data gradients 1.12 -> 1.01,
            // decoder forward 0.95 -> 0.86, layer4 3x3 -0.05; but layer3's 3x3 (K = 2304, only 288 tiles of 256 rows) LOSES
            // 0.24 ms over its 44 launches -- one 72-step tile per CU at two waves per SIMD -- hence the tile-count clause.
            static const int bm256 = getenv("DML_CONV_BM256") ? atoi(getenv("DML_CONV_BM256")) : 1;
            const int tiles256 = ((a.M + 255) / 256) * (a.N / 128);
            const bool long_k = bm256 >= 64 ? a.Ktot >= bm256 : (a.Ktot >= 4608 || (a.Ktot >= 2304 && tiles256 >= 512));
            // The 256-row tile runs as 4 waves on 128 x 64 WAVE tiles (conv_igemm_dma_kernel<..., 256, 128>: 32 MFMAs per wave and
            // K step against 12 KB of fragment reads, half the barriers per FLOP); the 8-wave variant on 64 x 64 wave tiles it
            // replaced in round 3 (decoder 3x3 data gradient 1.064 -> 0.957 ms, whole step 385.3 -> 388.3 images/s) is gone.
            if ((bm256 == 2 || (bm256 != 0 && long_k)) && a.N >= 128 && !narrow && a.N % 128 == 0) {
                // 256-row tiles: whole tiles on the first multiple of 256 workgroups, the remainder split along K
                constexpr int CUS = 256;
                a.nblk_m = (a.M + 255) / 256;
                a.nblk_n = a.N / 128;
                const int ntiles = a.nblk_m * a.nblk_n, full = ntiles / CUS * CUS, rem = ntiles - full;
                a.tail_full = 0;
                a.tail_q = 1;
                if (base.tail_ws != nullptr && full >= CUS && rem > 0 && rem <= CUS / 2) {
                    int q = CUS / rem;
                    if (q > 8) q = 8;
                    const int KTall = a.Ktot / BK;
                    while (q > 1 && KTall / q < 6) --q;
                    if (q > 1 && (int64_t)rem * q * 256 * 128 <= base.tail_ws_elems_ && rem <= base.tail_cnt_len_) {
                        a.tail_full = full;
                        a.tail_q = q;
                    }
                }
                const int grid = a.tail_q > 1 ? a.tail_full + (ntiles - a.tail_full) * a.tail_q : ntiles;
                hipLaunchKernelGGL((conv_igemm_dma_kernel<128, MODE, 3, 2, 256, 128>), dim3(grid), dim3(256), 0, st, a, (uint32_t)xb,
                                   (uint32_t)wb);
                DML_LAUNCH_CHECK();
                return 0;
            }
            if (a.tail_q > 1) {
                a.nblk_n = (a.N + 127) / 128;
                const int ntiles = a.nblk_m * a.nblk_n;
                hipLaunchKernelGGL((conv_igemm_dma_kernel<128, MODE, 3>), dim3(a.tail_full + (ntiles - a.tail_full) * a.tail_q),
                                   dim3(NTHREADS), 0, st, a, (uint32_t)xb, (uint32_t)wb);
            } else if (a.N > 64 && !narrow) {
                a.nblk_n = (a.N + 127) / 128;
                hipLaunchKernelGGL((conv_igemm_dma_kernel<128, MODE, 3>), dim3(a.nblk_m * a.nblk_n), dim3(NTHREADS), 0, st, a,
                                   (uint32_t)xb, (uint32_t)wb);
            } else {
                a.nblk_n = (a.N + 63) / 64;
                hipLaunchKernelGGL((conv_igemm_dma_kernel<64, MODE, 3>), dim3(a.nblk_m * a.nblk_n), dim3(NTHREADS), 0, st, a,
                                   (uint32_t)xb, (uint32_t)wb);
            }
            DML_LAUNCH_CHECK();
            return 0;
        }
    }
    if (a.w_tiled) return DML_EUNSUPPORTED;      // tile-major weights: only the LDS-DMA kernels above read them
    if constexpr (sizeof(T) == 4 && MODE != 2) {
        if (conv_ws_planes_eligible(a, MODE)) {
            // fp32 tensors, products of two fp16 planes per operand on the matrix cores (DmlConvDesc.x_planes ...)
            constexpr int CUS = 256, NLD = DML_WS_PLANES_NLD;
            const bool wide = (a.N % 256) == 0;
            // 64 output channels (layer1's 3x3, the 256 -> 64 1x1, the data gradients of the 64 -> 256 1x1): 192 x 64 tiles on four
            // 48 x 64 wave tiles.  On the 288 x 128 configuration half of every tile -- MFMAs and DMA pieces -- was empty
            // (r04: 0.16-0.17 of the class's own bound).  Same box, old -> new: 3x3 64 -> 64 at 192 x 192 forward 347 -> 286 us, 1x1
            // 256 -> 64 244 -> 185, 1x1 128 -> 64 190 -> 132; whole step +0.8 % (profiles/r05_ab_n64.txt)
            const bool n64 = a.N == 64;
            // Forward launches that leave most of the chip empty on 144-row tiles (batch-1 inference at 1024 x 2048: 57 row blocks on
            // layer3 / layer4 / ASPP; the small maps of the tests): 48 x 256 tiles -- three times the workgroups, each with a third
            // of the MFMAs per K step behind the same weight pieces.  (Round 6; 1024 x 2048 batch 1: see profiles/r06_infer_short_tiles.txt)
            const bool shortm = ws_planes_short(a, MODE);
            const int bm = wide ? (shortm ? 48 : 144) : (n64 ? 192 : 288);
            a.nblk_m = (a.M + bm - 1) / bm;
            a.nblk_n = wide ? a.N / 256 : (n64 ? 1 : (a.N + 127) / 128);      // (a last 128-wide block may be half empty: zero rows, no stores)
            const int ntiles = a.nblk_m * a.nblk_n;
            const int grid = ntiles < CUS ? ntiles : CUS;
            const uint32_t xpb = (uint32_t)((((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2), wpb = (uint32_t)((int64_t)a.N * a.Ktot * 2);
            // Half-tile configuration (round 6): short K loops with several tiles per workgroup -- the layer3 1x1
            // launches whose K loops (HBM nearly idle) and epilogues (matrix cores idle) alternate.  144 x 128 tiles, two consumer
            // and two loader waves, 74 KB of LDS: TWO workgroups per CU, one's epilogue under the other's K loop.
            static const int half_on = getenv("DML_WS_HALF") ? atoi(getenv("DML_WS_HALF")) : 0;
            static const int half_kt = getenv("DML_WS_HALF_KT") ? atoi(getenv("DML_WS_HALF_KT")) : 32;
            static const int half_stag = getenv("DML_WS_HALF_STAGGER") ? atoi(getenv("DML_WS_HALF_STAGGER")) : 0;
            const int half_tiles = ((a.M + 143) / 144) * (a.N / 128);
            const bool half = half_on != 0 && !n64 && (a.N % 128) == 0 && a.Ktot / BK <= half_kt && a.Ktot / BK >= 2 &&
                              half_tiles >= (half_on == 2 ? 1 : 4) * CUS;
            if (half) {
                a.nblk_m = (a.M + 143) / 144;
                a.nblk_n = a.N / 128;
                a.half_stagger = half_stag;
            }
            auto go = [&](auto epi_c) {
                constexpr int EPI = decltype(epi_c)::value;
                if (half)
                    hipLaunchKernelGGL((conv_ws_half_kernel<MODE, EPI>), dim3(half_tiles < 2 * CUS ? half_tiles : 2 * CUS), dim3(256), 0, st, a, xpb, wpb);
                else if (shortm) {
                    if constexpr (MODE == 0 && EPI == 0)
                        hipLaunchKernelGGL((conv_ws_kernel<1, 4, 0, NLD, 2, 3, 0>), dim3(grid), dim3((4 + NLD) * 64), 0, st, a, xpb, wpb);
                } else if (n64)
                    hipLaunchKernelGGL((conv_ws_kernel<4, 1, MODE, NLD, 2, 3, EPI>), dim3(grid), dim3((4 + NLD) * 64), 0, st, a, xpb, wpb);
                else if (wide)
                    hipLaunchKernelGGL((conv_ws_kernel<1, 4, MODE, NLD, 2, WS_MT, EPI>), dim3(grid), dim3((4 + NLD) * 64), 0, st, a, xpb, wpb);
                else
                    hipLaunchKernelGGL((conv_ws_kernel<2, 2, MODE, NLD, 2, WS_MT, EPI>), dim3(grid), dim3((4 + NLD) * 64), 0, st, a, xpb, wpb);
            };
            // data gradients: one instantiation per set of epilogue operands (conv_epilogue_rows_ops)
            const int epi = (MODE == 1 && DML_WS_EPI_OPS != 0)
                                ? ((a.accum != 0 || a.res_dz != nullptr) ? 1 : 0) | (a.bnr_partials != nullptr ? 2 : 0) : 0;
            if constexpr (MODE == 1 && DML_WS_EPI_OPS != 0) {
                if (epi == 3) go(std::integral_constant<int, 3>{});
                else if (epi == 2) go(std::integral_constant<int, 2>{});
                else if (epi == 1) go(std::integral_constant<int, 1>{});
                else go(std::integral_constant<int, 0>{});
            } else {
                go(std::integral_constant<int, 0>{});
            }
            DML_LAUNCH_CHECK();
            return 0;
        }
        if (a.f32_split && aligned && a.N > 32) {      // (f32_split == 2 launches the planes kernel did not take: three-term split)
            if (a.N > 64) {
                a.nblk_n = (a.N + 127) / 128;
                hipLaunchKernelGGL((conv_igemm_x3_kernel<128, MODE>), dim3(a.nblk_m * a.nblk_n), dim3(NTHREADS), 0, st, a);
            } else {
                a.nblk_n = (a.N + 63) / 64;
                hipLaunchKernelGGL((conv_igemm_x3_kernel<64, MODE>), dim3(a.nblk_m * a.nblk_n), dim3(NTHREADS), 0, st, a);
            }
            DML_LAUNCH_CHECK();
            return 0;
        }
    }
    if constexpr (MODE == 2) {
        return DML_EUNSUPPORTED;      // acc32: LDS-DMA kernels only (bf16, C % 32 == 0, N > 32)
    } else {
        auto go = [&](auto bn_tag, auto al_tag) {
            constexpr int BN = decltype(bn_tag)::value;
            constexpr bool AL = decltype(al_tag)::value;
            a.nblk_n = (a.N + BN - 1) / BN;
            const int grid = a.nblk_m * a.nblk_n;
            hipLaunchKernelGGL((conv_igemm_kernel<T, BN, AL, MODE>), dim3(grid), dim3(NTHREADS), 0, st, a);
        };
        using I128 = std::integral_constant<int, 128>;
        using I64 = std::integral_constant<int, 64>;
        using I32 = std::integral_constant<int, 32>;
        const int bn = a.N > 64 ? 128 : (a.N > 32 ? 64 : 32);
        if (aligned) {
            if (bn == 128) go(I128{}, std::true_type{});
            else if (bn == 64) go(I64{}, std::true_type{});
            else go(I32{}, std::true_type{});
        } else {
            if (bn == 128) go(I128{}, std::false_type{});
            else if (bn == 64) go(I64{}, std::false_type{});
            else go(I32{}, std::false_type{});
        }
        DML_LAUNCH_CHECK();
        return 0;
    }
}

}  // namespace

// Rows of the GEMM covered by one statistics partial (DmlConvDesc.stats / bnr_partials) of THIS launch: 48 where the
// wave-specialised kernel takes it, DML_STAT_ROWS otherwise.  The caller sizes the partial buffers with it and passes it on to
// dml_bn_finalize_rows / dml_bn_moments_rows (the BN-backward finalize only needs the group count).
extern "C" int dml_conv_stat_rows(const DmlConvDesc* d) {
    if (!d) return DML_STAT_ROWS;
    ConvArgs a;
    a.w_tiled = d->w_tiled; a.C = d->C; a.R = d->R; a.S = d->S; a.N = d->N; a.ws_min_tiles = d->ws_min_tiles;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.ldx = d->ldx;
    a.M = d->B * d->Ho * d->Wo;
    a.Ktot = d->R * d->S * d->C;
    a.y = d->y; a.ldy = d->ldy; a.bias = d->bias; a.accum = d->accum;
    a.post_scale = d->post_scale; a.post_shift = d->post_shift; a.post_mean = d->post_mean; a.post_res = d->post_res;
    a.post_ldres = d->post_ldres;
    if (d->dtype == DML_F32) {
        a.f32_split = d->f32_split; a.x_planes = d->x_planes; a.w_planes = d->w_planes;
        a.x_unscale = d->x_unscale; a.w_unscale = d->w_unscale;
        a.x_plane_bytes = (uint32_t)(d->x_plane_stride * 2); a.w_plane_bytes = (uint32_t)(d->w_plane_stride * 2);
        return (d->x_plane_stride < (1ll << 30) && d->w_plane_stride < (1ll << 30) && conv_ws_planes_eligible(a, d->mode))
                   ? ws_planes_stat_rows(a, d->mode) : DML_STAT_ROWS;
    }
    if (d->dtype != DML_BF16 || !d->w_tiled) return DML_STAT_ROWS;
    const int64_t xb = ((int64_t)(d->B * d->Hi) * d->Wi - 1) * d->ldx * 2 + (int64_t)d->C * 2;
    const int64_t wb = (int64_t)d->N * d->R * d->S * d->C * 2;
    static const bool use_v1 = getenv("DML_CONV_V1") != nullptr;
    return (!use_v1 && conv_ws_eligible(a, xb, wb, d->mode)) ? WS_STAT_ROWS : DML_STAT_ROWS;
}

extern "C" int dml_conv_igemm(const DmlConvDesc* d, void* stream) {
    if (!d || !d->x || !d->w || !d->y) return DML_EINVAL;
    if (d->dtype != DML_F32 && d->dtype != DML_BF16) return DML_EINVAL;
    const int vec = d->dtype == DML_BF16 ? 8 : 4;
    if (d->C % vec || d->ldx % vec) return DML_EALIGN;
    if (d->mode != 0 && d->mode != 1) return DML_EINVAL;
    if (d->mode == 1 && d->stride != 1 && d->stride != 2) return DML_EUNSUPPORTED;
    if (d->B <= 0 || d->Ho <= 0 || d->Wo <= 0 || d->N <= 0 || d->R <= 0 || d->S <= 0) return DML_EINVAL;
    if ((int64_t)d->B * d->Ho * d->Wo >= (1ll << 31)) return DML_EINVAL;
    if (d->stats && d->bias) return DML_EINVAL;
    ConvArgs a;
    a.x = d->x; a.w = d->w; a.y = d->y; a.bias = d->bias; a.stats = d->stats; a.dbg = nullptr;
    a.bnr_y = nullptr; a.bnr_mask = nullptr; a.bnr_mean = nullptr; a.bnr_invstd = nullptr; a.bnr_partials = nullptr;
    a.bnr_gmax = nullptr;
    a.bnr_ldy = 0; a.bnr_relu = 0;
    a.post_scale = nullptr; a.post_shift = nullptr; a.post_mean = nullptr; a.post_res = nullptr; a.post_ldres = 0;
    a.post_relu = 0;
    a.tail_ws = nullptr; a.tail_cnt = nullptr; a.tail_full = 0; a.tail_q = 1; a.tail_ws_elems_ = 0; a.tail_cnt_len_ = 0;
    a.res_dz = nullptr; a.res_mask = nullptr; a.res_ld = 0; a.nt_out = 0; a.half_stagger = 0;
    a.acc32 = nullptr; a.acc32_ld = 0; a.f32_split = d->dtype == DML_F32 ? d->f32_split : 0;
    if (a.f32_split < 0 || a.f32_split > 2) return DML_EINVAL;
    a.x_planes = nullptr; a.w_planes = nullptr; a.x_unscale = nullptr; a.w_unscale = nullptr; a.x_plane_bytes = a.w_plane_bytes = 0;
    if (a.f32_split == 2 && d->x_planes && d->w_planes) {
        if (!d->x_unscale || !d->w_unscale || d->x_plane_stride <= 0 || d->w_plane_stride <= 0 ||
            d->x_plane_stride >= (1ll << 30) || d->w_plane_stride >= (1ll << 30))
            return DML_EINVAL;
        if ((reinterpret_cast<uintptr_t>(d->x_planes) & 15) || (reinterpret_cast<uintptr_t>(d->w_planes) & 15) ||
            (d->x_plane_stride & 7) || (d->w_plane_stride & 7))
            return DML_EALIGN;
        a.x_planes = d->x_planes; a.w_planes = d->w_planes; a.x_unscale = d->x_unscale; a.w_unscale = d->w_unscale;
        a.x_plane_bytes = (uint32_t)(d->x_plane_stride * 2); a.w_plane_bytes = (uint32_t)(d->w_plane_stride * 2);
    }
    a.w_tiled = 0;
    a.ws_min_tiles = d->ws_min_tiles;
    if (d->w_tiled) {
        if (d->dtype != DML_BF16 || d->C % BK || d->N % 64) return DML_EUNSUPPORTED;
        a.w_tiled = 1;
    }
    if (d->acc32) {
        // fp32 staging of a gradient with several producers: the 16-byte-vector bf16 path of the data gradient only
        if (d->mode != 1 || d->dtype != DML_BF16 || d->y_f32 || d->accum || d->res_dz || d->bnr_partials) return DML_EINVAL;
        if (d->N % 8 || d->ldy % 8 || d->acc32_ld % 4 || (reinterpret_cast<uintptr_t>(d->y) & 15) ||
            (reinterpret_cast<uintptr_t>(d->acc32) & 15) || d->N <= 32)
            return DML_EALIGN;
        a.acc32 = d->acc32; a.acc32_ld = d->acc32_ld;
    }
    if (d->res_dz) {
        // masked residual gradient in the epilogue: the 16-byte-vector bf16 path of the data gradient only
        // (fp32: the two-plane kernel's row epilogue, 4-channel mask bytes; checked against the launch below)
        if (d->mode != 1 || d->y_f32 || d->accum || !d->res_mask) return DML_EINVAL;
        if (d->dtype == DML_BF16) {
            if (d->N % 8 || d->ldy % 8 || d->res_ld % 8 || (reinterpret_cast<uintptr_t>(d->y) & 15) ||
                (reinterpret_cast<uintptr_t>(d->res_dz) & 15) || d->N <= 32)
                return DML_EALIGN;
            if ((d->N & 63) == 0 && (reinterpret_cast<uintptr_t>(d->res_mask) & 7)) return DML_EALIGN;      // 8-byte mask rows
        } else if (d->N % 64 || d->res_ld % 4 || ((reinterpret_cast<uintptr_t>(d->res_dz) | reinterpret_cast<uintptr_t>(d->res_mask)) & 15)) {
            return DML_EALIGN;      // (16 mask bytes per pixel row and 64 channels are one load)
        }
        a.res_dz = d->res_dz; a.res_mask = d->res_mask; a.res_ld = d->res_ld;
    }
    if (d->tail_ws && d->tail_counters && d->tail_ws_elems > 0 && d->tail_counters_len > 0) {
        if (reinterpret_cast<uintptr_t>(d->tail_ws) & 15) return DML_EALIGN;
        a.tail_ws = d->tail_ws; a.tail_cnt = d->tail_counters;
        a.tail_ws_elems_ = d->tail_ws_elems; a.tail_cnt_len_ = d->tail_counters_len;
    }
    if (d->post_scale) {
        if (d->mode != 0 || !d->post_shift || !d->post_mean || d->stats || d->bias || d->accum || d->y_f32) return DML_EINVAL;
        if (d->post_res && (d->post_ldres % vec || (reinterpret_cast<uintptr_t>(d->post_res) & 15))) return DML_EALIGN;
        a.post_scale = d->post_scale; a.post_shift = d->post_shift; a.post_mean = d->post_mean; a.post_res = d->post_res;
        a.post_ldres = d->post_ldres; a.post_relu = d->post_relu;
    }
    a.bnr_gmax = nullptr;
    bool bnr_inc_req = false;
    if (d->bnr_partials) {
        // fused BN-backward reduce: data-gradient mode; bf16 result stored as 16-byte vectors, 8-channel mask bytes -- or fp32 on the
        // two-plane kernel (checked below, once the launch is described), 4-channel mask bytes
        if (d->mode != 1 || d->y_f32 || !d->bnr_y || !d->bnr_mean || !d->bnr_invstd || (d->bnr_relu && !d->bnr_mask))
            return DML_EINVAL;
        if (d->dtype == DML_BF16) {
            if (d->N % 8 || d->ldy % 8 || d->bnr_ldy % 8 || (reinterpret_cast<uintptr_t>(d->y) & 15) ||
                (reinterpret_cast<uintptr_t>(d->bnr_y) & 15) || (reinterpret_cast<uintptr_t>(d->bnr_partials) & 15) || d->N <= 32)
                return DML_EUNSUPPORTED;
            if (d->bnr_relu && (d->N & 63) == 0 && (reinterpret_cast<uintptr_t>(d->bnr_mask) & 7)) return DML_EALIGN;
        } else {
            if (d->N % 64 || d->bnr_ldy % 4) return DML_EUNSUPPORTED;
            if (((reinterpret_cast<uintptr_t>(d->bnr_y) | reinterpret_cast<uintptr_t>(d->bnr_partials) |
                  reinterpret_cast<uintptr_t>(d->bnr_mean) | reinterpret_cast<uintptr_t>(d->bnr_invstd) |
                  (d->bnr_relu ? reinterpret_cast<uintptr_t>(d->bnr_mask) : 0)) & 15) != 0)
                return DML_EALIGN;
            a.bnr_gmax = d->bnr_gmax;
        }
        a.bnr_y = d->bnr_y; a.bnr_mask = d->bnr_mask; a.bnr_mean = d->bnr_mean; a.bnr_invstd = d->bnr_invstd;
        a.bnr_partials = d->bnr_partials; a.bnr_ldy = d->bnr_ldy; a.bnr_relu = d->bnr_relu;
        bnr_inc_req = d->bnr_inc != 0;
    }
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.N = d->N; a.ldy = d->ldy;
    a.R = d->R; a.S = d->S; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.pad_x = d->pad; a.sub_grid = 0; a.sub_y = a.sub_x = 0; a.bnr_inc = 0;
    a.M = d->B * d->Ho * d->Wo;
    a.Ktot = d->R * d->S * d->C;
    a.y_f32 = d->y_f32; a.accum = d->accum;
    a.bnr_inc = 0;
    if (bnr_inc_req) {
        // (only the two-plane row epilogue with both operand sets knows the increment: an accumulating fp32 launch)
        if (!d->accum || d->dtype != DML_F32 || d->res_dz) return DML_EUNSUPPORTED;
        a.bnr_inc = 1;
    }
    if (d->pad_w_set) a.pad_x = d->pad_w;
    if (d->sub_grid) {
        // a parity class of a stride-2 data gradient: stride-1 geometry on dY's grid, the two-plane kernel's row epilogues only
        if (d->mode != 1 || d->stride != 1 || d->dil != 1 || d->Hi != d->Ho || d->Wi != d->Wo || (unsigned)d->sub_y > 1u ||
            (unsigned)d->sub_x > 1u || d->dtype != DML_F32)
            return DML_EINVAL;
        if ((int64_t)4 * d->B * d->Ho * d->Wo * (int64_t)d->ldy * 4 >= (1ll << 31)) return DML_EUNSUPPORTED;
        a.sub_grid = 1; a.sub_y = d->sub_y; a.sub_x = d->sub_x;
    }
    a.nblk_n = a.nblk_m = 0;
    a.div_wo = make_fastdiv((uint32_t)d->Wo);
    a.div_howo = make_fastdiv((uint32_t)(d->Ho * d->Wo));
    a.div_c = make_fastdiv((uint32_t)d->C);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if ((a.sub_grid || a.pad_x != a.pad) && (d->dtype == DML_BF16 || a.acc32)) return DML_EUNSUPPORTED;
    if (a.acc32) return launch_conv<bf16_t, 2>(a, st);
    if (d->dtype == DML_BF16)
        return d->mode == 0 ? launch_conv<bf16_t, 0>(a, st) : launch_conv<bf16_t, 1>(a, st);
    // fp32: only the two-plane kernel writes the BN-backward sums / adds the masked identity-branch gradient / maps a sub-grid
    if ((a.bnr_partials || a.res_dz || a.sub_grid || a.pad_x != a.pad) && !conv_ws_planes_eligible(a, d->mode)) return DML_EUNSUPPORTED;
    // x == x_planes: the caller keeps this operand as fp16 planes ONLY (no fp32 tensor behind `x`).  Every other fp32 kernel would
    // read the planes as floats: refuse instead of falling back
    if (d->f32_split == 2 && d->x_planes && d->x == d->x_planes && !conv_ws_planes_eligible(a, d->mode)) return DML_EUNSUPPORTED;
    return d->mode == 0 ? launch_conv<float, 0>(a, st) : launch_conv<float, 1>(a, st);
}

// tuning builds only (make tuning -> libdmlnet_hip_tuning.so; tools/bench_conv.py abl / phases / dmaphases): the 128 x 128
// forward kernels with parts removed or with s_memtime stamps.  Not part of the ABI, not in the product library.
#ifdef DML_TUNING
extern "C" int dml_debug_conv_ablate(const DmlConvDesc* d, int abl, float* dbg, const float* aux0, const float* aux1, void* stream) {
    ConvArgs a;
    a.x = d->x; a.w = d->w; a.y = d->y; a.bias = nullptr; a.stats = d->stats;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.N = d->N; a.ldy = d->ldy;
    a.R = d->R; a.S = d->S; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad; a.pad_x = d->pad; a.sub_grid = 0; a.sub_y = a.sub_x = 0; a.bnr_inc = 0;
    a.M = d->B * d->Ho * d->Wo; a.Ktot = d->R * d->S * d->C; a.y_f32 = 0; a.accum = 0;
    a.nblk_m = (a.M + 127) / 128; a.nblk_n = (a.N + 127) / 128;
    a.div_wo = make_fastdiv((uint32_t)d->Wo); a.div_howo = make_fastdiv((uint32_t)(d->Ho * d->Wo));
    a.div_c = make_fastdiv((uint32_t)d->C);
    a.dbg = dbg;
    a.bnr_y = nullptr; a.bnr_mask = nullptr; a.bnr_mean = nullptr; a.bnr_invstd = nullptr; a.bnr_partials = nullptr;
    a.bnr_gmax = nullptr;
    a.bnr_ldy = 0; a.bnr_relu = 0;
    a.post_scale = nullptr; a.post_shift = nullptr; a.post_mean = nullptr; a.post_res = nullptr; a.post_ldres = 0;
    a.post_relu = 0;
    a.tail_ws = nullptr; a.tail_cnt = nullptr; a.tail_full = 0; a.tail_q = 1; a.tail_ws_elems_ = 0; a.tail_cnt_len_ = 0;
    a.res_dz = nullptr; a.res_mask = nullptr; a.res_ld = 0; a.nt_out = 0; a.half_stagger = 0; a.acc32 = nullptr; a.acc32_ld = 0; a.f32_split = 0;
    a.w_tiled = 0; a.ws_min_tiles = 0;
    a.x_planes = nullptr; a.w_planes = nullptr; a.x_unscale = nullptr; a.w_unscale = nullptr; a.x_plane_bytes = a.w_plane_bytes = 0;
    a.x_bytes = (uint32_t)((((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2);
    a.w_bytes = (uint32_t)((int64_t)a.N * a.Ktot * 2);
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 grid(a.nblk_m * a.nblk_n);
    if (abl == 0) hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, true, 0, 0>), grid, dim3(NTHREADS), 0, st, a);
    else if (abl == 1) hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, true, 0, 1>), grid, dim3(NTHREADS), 0, st, a);
    else if (abl == 3) hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, true, 0, 3>), grid, dim3(NTHREADS), 0, st, a);
    else if (abl == 5) {          // the LDS-DMA 128 x 128 kernel with phase stamps; dbg = 8 floats per wave
        hipLaunchKernelGGL((conv_igemm_dma_kernel<128, 0, 3, 3, 128, 64, true>), grid, dim3(NTHREADS), 0, st, a, a.x_bytes, a.w_bytes);
    }
    else if (abl == 4) {
        a.dbg = nullptr;
        a.bnr_mean = aux0; a.bnr_invstd = aux1;       // per INPUT channel: the probe's scale / shift
        if (!a.bnr_mean || !a.bnr_invstd) return DML_EINVAL;
        hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, true, 0, 4>), grid, dim3(NTHREADS), 0, st, a);
    }
    else hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, 128, true, 0, 2>), grid, dim3(NTHREADS), 0, st, a);
    DML_LAUNCH_CHECK();
    return 0;
}

#endif

extern "C" int dml_conv_wgrad(const DmlWgradDesc* d, void* stream) {
    if (!d || !d->x || !d->dy || !d->dw) return DML_EINVAL;
    if (d->dtype != DML_F32 && d->dtype != DML_BF16) return DML_EINVAL;
    const int vec = d->dtype == DML_BF16 ? 8 : 4;
    if (d->C % vec || d->ldx % vec || d->N % vec || d->ldy % vec) return DML_EALIGN;
    if ((int64_t)d->B * d->Ho * d->Wo >= (1ll << 31)) return DML_EINVAL;
    WgradArgs a;
    a.x = d->x; a.dy = d->dy; a.dw = d->dw;
    a.B = d->B; a.Hi = d->Hi; a.Wi = d->Wi; a.C = d->C; a.ldx = d->ldx;
    a.Ho = d->Ho; a.Wo = d->Wo; a.N = d->N; a.ldy = d->ldy;
    a.R = d->R; a.S = d->S; a.stride = d->stride; a.dil = d->dil; a.pad = d->pad;
    a.M = d->B * d->Ho * d->Wo;
    a.Ktot = d->R * d->S * d->C;
    a.nblk_n = (a.N + 127) / 128;
    a.nblk_k = (a.Ktot + 127) / 128;
    a.div_wo = make_fastdiv((uint32_t)d->Wo);
    a.div_howo = make_fastdiv((uint32_t)(d->Ho * d->Wo));
    a.div_c = make_fastdiv((uint32_t)d->C);
    a.x_planes = nullptr; a.dy_planes = nullptr; a.x_unscale = nullptr; a.dy_unscale = nullptr; a.x_plane_bytes = a.dy_plane_bytes = 0;
    bool planes = false;
    if (d->dtype == DML_F32 && d->f32_split == 2 && d->x_planes && d->dy_planes) {
        if (!d->x_unscale || !d->dy_unscale || d->x_plane_stride <= 0 || d->dy_plane_stride <= 0) return DML_EINVAL;
        if ((reinterpret_cast<uintptr_t>(d->x_planes) & 15) || (reinterpret_cast<uintptr_t>(d->dy_planes) & 15) ||
            (d->x_plane_stride & 7) || (d->dy_plane_stride & 7))
            return DML_EALIGN;
        const int64_t xpb = (((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2, ypb = (((int64_t)a.M - 1) * a.ldy + a.N) * 2;
        // fp16 vectors of 8: C, N, pitches multiples of 8; both planes within 31-bit offsets
        planes = d->C % 8 == 0 && d->N % 8 == 0 && d->ldx % 8 == 0 && d->ldy % 8 == 0 && xpb + d->x_plane_stride * 2 < 0x7fffffffll &&
                 ypb + d->dy_plane_stride * 2 < 0x7fffffffll;
        if (planes) {
            a.x_planes = d->x_planes; a.dy_planes = d->dy_planes; a.x_unscale = d->x_unscale; a.dy_unscale = d->dy_unscale;
            a.x_plane_bytes = (uint32_t)(d->x_plane_stride * 2); a.dy_plane_bytes = (uint32_t)(d->dy_plane_stride * 2);
        }
    }
    // x == x_planes / dy == dy_planes: that operand exists as fp16 planes ONLY -- the fp32 kernels below must not read it as floats
    if (!planes && ((d->x_planes && d->x == d->x_planes) || (d->dy_planes && d->dy == d->dy_planes))) return DML_EUNSUPPORTED;
    const int tiles = (a.M + BK - 1) / BK;
    const int cm = d->Cm > 0 ? d->Cm : d->C;
    if (cm > d->C) return DML_EINVAL;
    if (cm != d->C && !d->ws) return DML_EINVAL;      // dropping padded channels needs the workspace path
    const int64_t plane = (int64_t)a.N * a.Ktot;
    // With a workspace every weight gradient is a fixed-order sum of slabs: the train step is bitwise reproducible.  (Small
    // weight tensors -- below 64 K elements -- used to take fp32 atomics instead, the reduce pass being latency-bound there;
    // DML_WGRAD_ATOMICS=1 restores that.)  Without a workspace: atomics.
    const bool use_ws = d->ws != nullptr;
    int splitk = d->splitk;
    if (splitk <= 0) {
        const int base = a.nblk_n * a.nblk_k;
        if (use_ws) {
            // measured (tools/bench_conv.py): >= 512 workgroups and ~48 K tiles per workgroup
            splitk = (512 + base - 1) / base;
            if (tiles / 48 > splitk) splitk = tiles / 48;
        } else {
            splitk = (1024 + base - 1) / base;
        }
        const int cap = (tiles + 7) / 8;
        if (splitk > cap) splitk = cap;
        if (splitk < 1) splitk = 1;
    }
    if (use_ws && (int64_t)splitk * plane > d->ws_elems) splitk = (int)(d->ws_elems / plane);
    if (use_ws && splitk < 1) return DML_EINVAL;
    hipStream_t st = static_cast<hipStream_t>(stream);
    // large-tile kernel: bf16, whole 256-channel output tiles, everything addressable with 31-bit byte offsets and
    // 24-bit pixel indices.  Measured (tools/bench_conv.py wgrad, incl. the reduce pass, vs the 128 x 128 kernel):
    // decoder 3x3 320->256 @192^2 970 vs 727 TFLOP/s, 3x3 512->512 d2 879 vs 669, ASPP 3x3 2048->256 857 vs 693,
    // layer3 3x3 256->256 569 vs 500.  With 4 output tiles (the 1x1 256<->1024 layers) filling 256 CUs would take 64
    // splits and the slab traffic eats the gain, so those run as 128 workgroups (32 splits): standalone that is a
    // few % slower than the small-tile kernel, but the weight gradients share the chip with the main stream and what
    // counts there is CU-time -- whole step +0.75 % (3 interleaved A/B runs of bench.py).
    const int64_t xb64 = (((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2;
    const int64_t yb64 = (((int64_t)a.M - 1) * a.ldy + a.N) * 2;
    if (d->dtype == DML_BF16 && use_ws && a.N % 256 == 0 && (a.N / 256) * ((a.Ktot + 255) / 256) >= 4 &&
        tiles >= 16 &&
        xb64 < (1ll << 31) && yb64 < (1ll << 31) && (int64_t)(a.B + 1) * a.Hi * a.Wi < (1 << 24) && a.M < (1 << 24) &&
        a.ldx < (1 << 22) && a.ldy < (1 << 22)) {
        a.nblk_n = a.N / 256;
        a.nblk_k = (a.Ktot + 255) / 256;
        const int base = a.nblk_n * a.nblk_k;
        int sk = d->splitk;
        if (sk <= 0) {
            // one workgroup per CU: pick the split count that fills whole rounds of 256 (128) workgroups best,
            // fewest splits on ties (less slab traffic); at least 8 K steps per workgroup
            int smax = tiles / 8;
            if (smax > 256) smax = 256;
            if ((int64_t)smax * plane > d->ws_elems) smax = (int)(d->ws_elems / plane);
            if (smax < 1) smax = 1;
            double best = -1.0;
            sk = 1;
            for (int c = 1; c <= smax; ++c) {
                const int slab = (tiles + c - 1) / c;
                const int real = (tiles + slab - 1) / slab;
                const int blocks = base * real;
                const int rnd = base < 6 ? 128 : 256;
                const double eff = (double)blocks / (double)(((blocks + rnd - 1) / rnd) * rnd);
                if (eff > best + 1e-9) { best = eff; sk = real; }
            }
        }
        if ((int64_t)sk * plane > d->ws_elems) sk = (int)(d->ws_elems / plane);
        if (sk < 1) return DML_EINVAL;
        if (sk > tiles) sk = tiles;
        a.ws = d->ws;
        a.slab_tiles = (tiles + sk - 1) / sk;
        sk = (tiles + a.slab_tiles - 1) / a.slab_tiles;
        hipLaunchKernelGGL(conv_wgrad_big_kernel<WGRAD_DEPTH>, dim3(base * sk), dim3(WB_THREADS), 0, st, a, (uint32_t)xb64, (uint32_t)yb64);
        const int64_t nrs = (int64_t)a.N * a.R * a.S;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(nrs * cm, 256), 1), dim3(256), 0, st, d->ws, d->dw, sk, nrs,
                           cm, d->C, 1, 0);
        DML_LAUNCH_CHECK();
        return 0;
    }
    if (planes && use_ws && a.N % WGW_TN == 0 && tiles >= 16 && (reinterpret_cast<uintptr_t>(d->ws) & 15) == 0) {
        // wave-specialised kernel on planes: 128 x 256 output tiles, persistent workgroups over (tile, slab) items.  Against
        // conv_wgrad_h2_kernel in isolation (same box, tools/bench_h2.py wgrad): 3x3 256 -> 256 167-176 -> 149 us, 1x1 1024 <-> 256
        // 76-83 -> 69-71, ASPP 3x3 1070-1147 -> 939, 3x3 512 -> 512 563 -> 473, decoder 3x3 2614-2724 -> 2532; in the train step,
        // where the weight gradients share the chip with the main stream, the step is unchanged (183.6 vs 183.2 images/s, three
        // interleaved pairs): DESIGN.md section 5, round 4
        a.nblk_n = a.N / WGW_TN;
        a.nblk_k = (a.Ktot + WGW_TK - 1) / WGW_TK;
        const int base = a.nblk_n * a.nblk_k;
        int sk = d->splitk;
        if (sk <= 0) {
            // the split count with the smallest modelled time: rounds of 256 workgroups x (K steps of an item x 1.45 us + 8 us for
            // its 128 KB slab store and the restart of the K loop) + the fold pass over the slabs (written and read once, ~4 TB/s);
            // >= 16 K steps per item.  (Round 4 maximised the filling of the rounds alone and took 71 slabs of 16 steps for 3x3
            // 256 -> 256 -- five rounds, a third of each an epilogue, 335 MB of slabs -- where 14 slabs of 83 steps fill one round.)
            int smax = tiles / 16;
            if (smax > 256) smax = 256;
            if ((int64_t)smax * plane > d->ws_elems) smax = (int)(d->ws_elems / plane);
            if (smax < 1) smax = 1;
            double best = 1e30;
            sk = 1;
            for (int c = 1; c <= smax; ++c) {
                const int slab = (tiles + c - 1) / c;
                const int real = (tiles + slab - 1) / slab;
                const int items = base * real;
                const double t = (double)((items + 255) / 256) * (slab * 1.45 + 8.0) + (double)real * (double)plane * 8.0 / 4.0e6;
                if (t < best - 1e-9) { best = t; sk = real; }
            }
        }
        if ((int64_t)sk * plane > d->ws_elems) sk = (int)(d->ws_elems / plane);
        if (sk < 1) return DML_EINVAL;
        if (sk > tiles) sk = tiles;
        a.ws = d->ws;
        a.slab_tiles = (tiles + sk - 1) / sk;
        sk = (tiles + a.slab_tiles - 1) / a.slab_tiles;
        const int nitems = base * sk;
        const uint32_t xpb = (uint32_t)((((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2), ypb = (uint32_t)((((int64_t)a.M - 1) * a.ldy + a.N) * 2);
        // (one item per workgroup instead of persistent workgroups, with and without a high-priority main stream: no change of the
        // overlapped step, 83.4-83.9 ms in every combination, serial 86.1 -- profiles/r05_ab_wgrad_nonpersist_priority.txt)
        const dim3 wgrid(nitems < 256 ? nitems : 256), wblock((4 + WGW_NLD) * 64);
        if (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0)
            hipLaunchKernelGGL(conv_wgrad_ws_kernel<true>, wgrid, wblock, 0, st, a, xpb, ypb, nitems);
        else
            hipLaunchKernelGGL(conv_wgrad_ws_kernel<false>, wgrid, wblock, 0, st, a, xpb, ypb, nitems);
        const int64_t nrs = (int64_t)a.N * a.R * a.S;
        // (one-wave blocks: they start beside the persistent data-gradient workgroups of the other stream instead of waiting for one
        // to retire -- 28.6 us per launch in the overlapped trace against 9.8 alone, 122 launches per step; see dml_bn_bwd_apply)
        static const int rt = getenv("DML_WGRAD_REDUCE_THREADS") ? atoi(getenv("DML_WGRAD_REDUCE_THREADS")) : 64;
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(nrs * cm, rt, 256 * 8 * (256 / rt)), 1), dim3(rt), 0, st, d->ws, d->dw, sk, nrs,
                           cm, d->C, 1, 0);
        DML_LAUNCH_CHECK();
        return 0;
    }
    a.ws = use_ws ? d->ws : nullptr;
    // N <= 64 (bf16): 64 x 256 tiles instead of half-empty 128 x 128 ones
    // (measured, tools/bench_conv.py wgrad at 192 x 192: 3x3 64 -> 64 164 -> 160 us; the 1x1 layers are bound by the
    // operand loads, not by the idle MFMA rows, and LOSE 20 % with the wider x tile -- they stay on the 128 x 128 kernel)
    const bool n64 = d->dtype == DML_BF16 && a.N <= 64 && a.R * a.S > 1;
    if (n64) {
        a.nblk_n = 1;
        a.nblk_k = (a.Ktot + 255) / 256;
        if (d->splitk <= 0) {
            const int base = a.nblk_k;
            splitk = use_ws ? (512 + base - 1) / base : (1024 + base - 1) / base;
            if (use_ws && tiles / 48 > splitk) splitk = tiles / 48;
            const int cap = (tiles + 7) / 8;
            if (splitk > cap) splitk = cap;
            if (splitk < 1) splitk = 1;
            if (use_ws && (int64_t)splitk * plane > d->ws_elems) splitk = (int)(d->ws_elems / plane);
        }
    }
    if (splitk > tiles) splitk = tiles > 0 ? tiles : 1;
    a.slab_tiles = (tiles + splitk - 1) / splitk;
    splitk = (tiles + a.slab_tiles - 1) / a.slab_tiles;
    dim3 grid(a.nblk_n * a.nblk_k * splitk);
    // split-product kernel: operands through buffer descriptors (32-bit byte offsets)
    const int64_t x3_xb = (((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 4, x3_yb = (((int64_t)a.M - 1) * a.ldy + a.N) * 4;
    const bool x3_ok = x3_xb < (int64_t)0x7fffffff && x3_yb < (int64_t)0x7fffffff;
    if (planes) {
        const uint32_t xpb = (uint32_t)((((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2), ypb = (uint32_t)((((int64_t)a.M - 1) * a.ldy + a.N) * 2);
        if (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0)
            hipLaunchKernelGGL(conv_wgrad_h2_kernel<true>, grid, dim3(NTHREADS), 0, st, a, xpb, ypb);
        else
            hipLaunchKernelGGL(conv_wgrad_h2_kernel<false>, grid, dim3(NTHREADS), 0, st, a, xpb, ypb);
    } else if (n64)
        hipLaunchKernelGGL(conv_wgrad_n64_kernel, grid, dim3(NTHREADS), 0, st, a);
    else if (d->dtype == DML_BF16)
        hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(NTHREADS), 0, st, a);
    else if (d->f32_split && x3_ok) {
        if (a.R == 1 && a.S == 1 && a.stride == 1 && a.pad == 0)
            hipLaunchKernelGGL(conv_wgrad_x3_kernel<true>, grid, dim3(NTHREADS), 0, st, a, (uint32_t)x3_xb, (uint32_t)x3_yb);
        else
            hipLaunchKernelGGL(conv_wgrad_x3_kernel<false>, grid, dim3(NTHREADS), 0, st, a, (uint32_t)x3_xb, (uint32_t)x3_yb);
    } else
        hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(NTHREADS), 0, st, a);
    if (use_ws) {
        const int64_t nrs = (int64_t)a.N * a.R * a.S;
        // small tensors with very many splits: fold chunks of 16 splits first (two launches, see wgrad_reduce_kernel)
        const int ychunks = (nrs * cm < 65536 && splitk > 64) ? (splitk + 15) / 16 : 1;
        if (ychunks > 1) {
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(nrs * cm, 256), ychunks), dim3(256), 0, st, d->ws, d->dw, splitk,
                               nrs, cm, d->C, 1, 1);
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(nrs * cm, 256)), dim3(256), 0, st, d->ws, d->dw, ychunks, nrs, cm,
                               d->C, 16, 0);
        } else {
            hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(grid_for(nrs * cm, 256)), dim3(256), 0, st, d->ws, d->dw, splitk, nrs, cm,
                               d->C, 1, 0);
        }
    }
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_conv_wgrad_group_eligible(const DmlWgradDesc* d) { return wgrad_big_eligible(d) ? 1 : 0; }

// Weight gradients of several layers in one launch (conv_wgrad_group_kernel) + one fold launch.  Every job must pass
// dml_conv_wgrad_group_eligible; the descriptors' own ws / ws_elems / splitk fields are ignored: the shared workspace
// `ws` is partitioned here.  Splits are chosen so that the launch is one round of <= 256 workgroups with about the
// same number of pixel tiles per workgroup in every job.
extern "C" int dml_conv_wgrad_group(const DmlWgradDesc* const* descs, int n, float* ws, int64_t ws_elems, void* stream) {
    if (!descs || n <= 0 || n > WG_MAX_JOBS || !ws) return DML_EINVAL;
    WgradGroup g;
    ReduceGroup r;
    int base[WG_MAX_JOBS], tiles[WG_MAX_JOBS], sk[WG_MAX_JOBS];
    int64_t plane[WG_MAX_JOBS];
    double work = 0.0;
    int tbase = 0;
    for (int j = 0; j < n; ++j) {
        const DmlWgradDesc* d = descs[j];
        if (!wgrad_big_eligible(d)) return DML_EUNSUPPORTED;
        WgradArgs& a = g.job[j];
        fill_wgrad_args(a, d);
        a.nblk_n = a.N / 256;
        a.nblk_k = (a.Ktot + 255) / 256;
        base[j] = a.nblk_n * a.nblk_k;
        tiles[j] = (a.M + BK - 1) / BK;
        plane[j] = (int64_t)a.N * a.Ktot;
        work += (double)base[j] * tiles[j];
        tbase += base[j];
        g.xb[j] = (uint32_t)((((int64_t)(a.B * a.Hi) * a.Wi - 1) * a.ldx + a.C) * 2);
        g.yb[j] = (uint32_t)((((int64_t)a.M - 1) * a.ldy + a.N) * 2);
    }
    const int budget = tbase <= 256 ? 256 : ((tbase + 255) / 256) * 256;
    int blocks = 0;
    for (int j = 0; j < n; ++j) {
        int s = (int)((double)tiles[j] * budget / work);             // equal pixel tiles per workgroup
        const int cap = tiles[j] / 8 > 0 ? tiles[j] / 8 : 1;          // at least 8 K steps per workgroup
        if (s > cap) s = cap;
        if (s < 1) s = 1;
        sk[j] = s;
        blocks += base[j] * s;
    }
    // hand the remaining workgroups of the round to the jobs whose workgroups are the longest
    for (;;) {
        int best = -1;
        double longest = 0.0;
        for (int j = 0; j < n; ++j) {
            const double per = (double)tiles[j] / sk[j];
            if (blocks + base[j] <= budget && sk[j] < tiles[j] / 8 && per > longest) { longest = per; best = j; }
        }
        if (best < 0) break;
        ++sk[best];
        blocks += base[best];
    }
    int64_t off = 0;
    g.njobs = r.njobs = n;
    g.start[0] = 0;
    int64_t max_items = 0;
    for (int j = 0; j < n; ++j) {
        WgradArgs& a = g.job[j];
        a.slab_tiles = (tiles[j] + sk[j] - 1) / sk[j];
        sk[j] = (tiles[j] + a.slab_tiles - 1) / a.slab_tiles;
        if (off + (int64_t)sk[j] * plane[j] > ws_elems) return DML_EINVAL;      // workspace too small for this group
        a.ws = ws + off;
        g.start[j + 1] = g.start[j] + base[j] * sk[j];
        const DmlWgradDesc* d = descs[j];
        const int cm = d->Cm > 0 ? d->Cm : d->C;
        r.ws[j] = a.ws; r.dw[j] = d->dw; r.splits[j] = sk[j]; r.Cm[j] = cm; r.Cp[j] = d->C;
        r.NRS[j] = (int64_t)a.N * a.R * a.S;
        if (r.NRS[j] * cm > max_items) max_items = r.NRS[j] * cm;
        off += (int64_t)sk[j] * plane[j];
    }
    hipStream_t st = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(conv_wgrad_group_kernel<WGRAD_DEPTH>, dim3(g.start[n]), dim3(WB_THREADS), 0, st, g);
    hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(grid_for(max_items, 256, 1024), n), dim3(256), 0, st, r);
    DML_LAUNCH_CHECK();
    return 0;
}
