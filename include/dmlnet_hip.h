/*
 * dmlnet_hip.h -- C ABI of libdmlnet_hip.so, the MI355X (gfx950) kernels under the DMLNet hot path.
 *
 * The reference (Jun-CEN/Open-World-Semantic-Segmentation) is 100 % Python and has no FFI: every
 * entry point below replaces a stock ATen/cuDNN op that the reference reaches through the call
 * site cited next to it (paths relative to /root/reference/DeepLabV3Plus-Pytorch unless noted).
 * The Python package `open-world-semantic-segmentation_amd/network` binds these with ctypes and
 * re-exposes the reference's own module API (network.deeplabv3plus_embedding_resnet101, ...).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless stated;
 *   - activations are NHWC, channel pitch ("ld", in elements) given explicitly so that producers
 *     can write straight into channel slices of a concat buffer (replaces torch.cat,
 *     network/utils.py:32,360);
 *   - conv weights are K-R-S-C ("OHWI"): w[n][r][s][c];
 *   - `dtype` selects the storage type of activations / compute weights: DML_F32 or DML_BF16;
 *     accumulation, batch-norm statistics, the distance head, the loss and the optimizer state
 *     are always fp32;
 *   - the caller owns every buffer (including workspaces); nothing here allocates, frees or
 *     synchronises; every launch goes to `stream` (a hipStream_t passed as void*);
 *   - return value: 0 on success, a negative DML_E* code for a rejected argument, or a positive
 *     hipError_t from the launch.  Nothing throws.
 */
#ifndef DMLNET_HIP_H
#define DMLNET_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DML_ABI_VERSION 6

enum { DML_F32 = 0, DML_BF16 = 1 };
enum { DML_EINVAL = -1, DML_EALIGN = -2, DML_EUNSUPPORTED = -3 };

int dml_abi_version(void);
/* Name of the code object the library was built for ("gfx950"). */
const char* dml_target_arch(void);

/* ------------------------------------------------------------------------------------------------
 * Convolution as implicit GEMM on MFMA (nn.Conv2d call sites: network/backbone/resnet.py:24-32,139;
 * network/utils.py:11-23,311,322,337-352).
 * ---------------------------------------------------------------------------------------------- */
typedef struct DmlConvDesc {
    const void* x;        /* source activations [B,Hi,Wi,C] (fwd: conv input; dgrad: dY), pitch ldx    */
    const void* w;        /* fwd: w[N][R][S][C]; dgrad: transposed copy wt[N=Cin][R][S][C=Cout]         */
    void* y;              /* result [B,Ho,Wo,N], pitch ldy                                               */
    const float* bias;    /* optional [N] (only network/utils.py:23 has a bias)                          */
    float* stats;         /* optional BN partials [ceil(M/rows)][N][2], rows = dml_conv_stat_rows(): (sum, M2 about the */
                          /* group mean) taken from the fp32 accumulators (fused K9)         */
    int32_t B, Hi, Wi, C, ldx;
    int32_t Ho, Wo, N, ldy;
    int32_t R, S, stride, dil, pad;
    int32_t dtype;        /* DML_F32 | DML_BF16 for x, w, y                                              */
    int32_t y_f32;        /* 1: write y as float regardless of dtype                                     */
    int32_t accum;        /* 1: y += result                                                              */
    int32_t mode;         /* 0 = forward gather, 1 = data-gradient gather (transposed conv)              */
    /* (ABI 2: the pre_scale / pre_shift / pre_relu fields of ABI 1 -- the producer's BatchNorm + ReLU applied on the operand
     * load -- are gone: measured on the register-staged kernel in round 1 and on the loader-wave structure in round 4, the
     * transform costs the convolution more than the BatchNorm apply pass it would remove) */
    /* mode 1 only, optional (all NULL/0 otherwise): the result y is the output gradient dz of a BatchNorm
     * (+ReLU) whose pre-normalisation tensor is bnr_y [M][N] (pitch bnr_ldy) with the 1-bit ReLU mask of
     * dml_bn_apply; the epilogue then also writes that BN's backward partial sums, exactly what
     * dml_bn_bwd_reduce would produce from the stored dz: bnr_partials[ceil(M/rows)][N][2], rows = dml_conv_stat_rows() =
     * (sum g, sum g*(bnr_y - mean)*invstd), g = dz * [mask bit] -- pass it with nblocks = ceil(M/rows) to
     * dml_bn_bwd_finalize.  DML_BF16: N % 8 == 0, N > 32.  DML_F32: only launches the two-plane kernel takes (f32_split == 2 with
     * planes; dml_conv_stat_rows() == 48), bnr_y fp32, the mask one byte per four channels, N % 64 == 0; bnr_gmax (optional there):
     * 1024 zeroed floats whose maximum afterwards is max |g| (as `gmax` of dml_bn_bwd_reduce, for dml_h2_bound_bn_bwd). */
    const void* bnr_y;
    const uint8_t* bnr_mask;
    const float* bnr_mean;
    const float* bnr_invstd;
    float* bnr_partials;
    int32_t bnr_ldy, bnr_relu;
    /* mode 0 only, optional: inference epilogue -- y = act((conv - post_mean) * post_scale + post_shift [+ post_res]),
     * i.e. BatchNorm with running statistics (dml_bn_eval_coeffs gives scale / shift, post_mean = running_mean),
     * the bottleneck's residual add (resnet.py:110-113) and ReLU applied to the fp32 accumulators, so the
     * pre-normalisation tensor is never written.  post_res [M][N] has the storage dtype and pitch post_ldres. */
    const float* post_scale;
    const float* post_shift;
    const float* post_mean;
    const void* post_res;
    int32_t post_ldres, post_relu;
    /* optional (bf16, LDS-DMA kernel): balance a launch whose tile count leaves the last round of workgroups mostly
     * empty (576 tiles of 128 x 128 on 256 CUs x 3 slots: a quarter of the CUs carry three tiles, the rest two).  The
     * tiles beyond the last multiple of the CU count are split along K into q workgroups each, so that every CU gets
     * the same share; the parts park their fp32 accumulators in tail_ws ([tiles][q][128*128] floats) with write-through
     * (sc1) stores, wait for them to drain, and take a RELAXED ticket from tail_counters; the part that draws the last
     * ticket reads all parts back with sc1 loads, adds them in part order (deterministic) and runs the epilogue -- no
     * release / acquire fences (an agent-scope release would write back every dirty line of the XCD's L2).
     * tail_counters must be zero on entry and is zero again when the launch has completed.  tail_ws / tail_counters may
     * be shared by every conv of a plan ONLY if those launches are serialised on one stream (two concurrent launches
     * would mix their slabs and tickets); after an aborted launch the caller must zero tail_counters.  NULL = never
     * split. */
    float* tail_ws;
    int64_t tail_ws_elems;
    int32_t* tail_counters;
    int32_t tail_counters_len, tail_reserved;
    /* mode 1 only, optional (same conditions as bnr_*: bf16, or fp32 on the two-plane kernel; not with accum): y = conv + res_dz (.)
     * [res_mask bit], i.e. the gradient that reaches a bottleneck's input through its identity branch -- the block output's gradient
     * res_dz [M][N] (pitch res_ld) masked by the 1-bit ReLU mask of that output (dml_bn_apply's, one byte per 16-byte vector) --
     * is added in the epilogue of conv1's data gradient (resnet.py:96-113 backward), so that the BN-backward apply of
     * the block's bn3 need not write the masked copy (dres = NULL there) for this launch to read back. */
    const void* res_dz;
    const uint8_t* res_mask;
    int32_t res_ld, res_reserved;
    /* mode 1 only, optional (bf16, same conditions as bnr_*; not with accum / y_f32 / res_dz / bnr_*): y = bf16(conv + acc32).
     * A gradient with several producers (autograd's fp32 sum at network/utils.py:360 -- the five ASPP branches reading
     * `out` -- and at resnet.py:96-113 where a block input also feeds the downsample branch) is accumulated in an fp32
     * staging tensor acc32 [M][N] (pitch acc32_ld floats) by the earlier producers (y_f32 = 1, accum = 0 / 1) and the
     * LAST producer adds it to its accumulators and rounds the total once into the bf16 tensor y. */
    const float* acc32;
    int32_t acc32_ld;
    /* dtype DML_F32 only (2: see x_planes below): 1 = compute the products on the bf16 matrix cores through a three-term split of both operands
     * (x = hi + mid + lo, six bf16 MFMAs per block: fp32-level error, 2.7x fewer matrix cycles than v_mfma_f32_16x16x4_f32);
     * 0 = the exact fp32 MFMA, the reference's arithmetic (network/utils.py:84-118 computes in fp32).  Shapes the split kernel
     * does not take (C % 32 != 0, N <= 32) run exact either way.
     * Both split modes REQUIRE finite operands within the narrow format's range: with f32_split == 1 an Inf (or |x| > 3.39e38,
     * which rounds to a bf16 Inf) becomes hi = Inf, mid = Inf - Inf = NaN, so the output is NaN where exact fp32 would carry
     * the Inf; with f32_split == 2 the scale of a tensor that holds an Inf or NaN is 0 / NaN and the whole output is NaN.
     * Either way a non-finite input is visible in the output (never clipped or dropped), but not element for element as in
     * the exact mode.  Residuals below 2^-126 (fp32 subnormals) are flushed. */
    int32_t f32_split;
    /* 1 = `w` is the tile-major copy dml_prep_weights writes (DmlPrepDesc.w_tiled: [N / 64][K / 32][64][32], K = R * S * C): the
     * weight half of every LDS-DMA instruction is one contiguous KB (whole 128-byte lines) instead of sixteen 64-byte row
     * segments.  bf16, C % 32 == 0, N % 64 == 0, LDS-DMA kernels only: DML_EUNSUPPORTED otherwise (never a silent re-layout). */
    int32_t w_tiled;
    /* smallest tile count for which the wave-specialised kernel (one persistent workgroup per CU, dml_conv_stat_rows) takes an
     * eligible launch: 0 = the library's default (192 tiles = three quarters of the CUs), 1 = whenever the shape allows
     * (tests), INT32_MAX = never. */
    int32_t ws_min_tiles;
    /* dtype DML_F32, f32_split == 2: the products on the fp16 matrix cores through a TWO-term split of the power-of-two-scaled
     * operands -- x s = xh + xl, w t = wh + wl (fp16 hi / lo planes written by dml_h2_split), per block wl xh + wh xl + wh xh, fp32
     * accumulation, result unscaled by the exact 1 / (s t): relative error ~2^-21 per element, the reference's own fp32-vs-fp64
     * level on the parity fixtures (tests/tools/emu_split_terms.py) at three MFMAs per block instead of the six of
     * f32_split == 1.  x_planes / w_planes: [2][...] fp16, the hi plane then the lo plane `*_plane_stride` elements further, the
     * activation planes with the geometry (pitch ldx) of `x`, the weight planes tile-major ([N / 64][K / 32][64][32] per plane);
     * x_unscale / w_unscale: device scalars 1 / s and 1 / t (dml_h2_split).  Shapes the planes kernel does not take (C % 32,
     * N % 64, more than 32 taps, misaligned y) run the three-term split on x / w, which must therefore be valid as well --
     * UNLESS the caller keeps the activation as planes only and says so by passing x == x_planes: such a launch returns
     * DML_EUNSUPPORTED when the planes kernel cannot take it (never a fallback that would read the planes as floats). */
    const void* x_planes;
    const void* w_planes;
    const float* x_unscale;
    const float* w_unscale;
    int64_t x_plane_stride, w_plane_stride;
    float* bnr_gmax;      /* see bnr_* above                                                             */
    /* mode 1, two-plane launches (f32_split == 2) only -- one PARITY CLASS of the data gradient of a stride-2 convolution, issued as
     * a stride-1 launch on the grid of dY (round 6): a pixel of dX only sees the filter taps of its own row / column parity, so the
     * data gradient of a 3x3 stride-2 convolution is four stride-1 correlations with 1, 2, 2 and 4 taps (nine instead of 36 tap
     * products per four pixels), that of a 1x1 stride-2 convolution one 1x1 launch on the even pixels.  sub_grid = 1: the rows of this
     * launch (b, y2, x2 on the Ho x Wo grid of the descriptor) are the pixels (b, 2 y2 + sub_y, 2 x2 + sub_x) of y, res_dz, bnr_y and
     * the masks, tensors of B x 2 Ho x 2 Wo pixels; bnr_partials receives ceil(B Ho Wo / 48) groups (the caller offsets the pointer
     * per class).  pad_w_set = 1: `pad_w` replaces `pad` along the width (a class's sub-filter is R x S = 1|2 x 1|2 taps with
     * padding R - 1 / S - 1).  Any other kernel family returns DML_EUNSUPPORTED for such a descriptor. */
    int32_t sub_grid, sub_y, sub_x;
    int32_t pad_w_set, pad_w;
    /* with bnr_* and an accumulating launch (accum = 1): the sums are taken over THIS launch's increment (the convolution result under
     * the mask, before the old value is added) instead of over the stored total -- the BatchNorm-backward sums are linear in the
     * gradient, so the producers of a gradient with several producers can each emit the sums of their own share into their own partial
     * groups (the 1x1 stride-2 data gradient as a parity-class launch only visits a quarter of the pixels: the first producer writes
     * the sums of its share over all pixels).  bnr_gmax still receives the maximum of the stored total. */
    int32_t bnr_inc;
} DmlConvDesc;

#define DML_STAT_ROWS 64   /* rows of the GEMM covered by one statistics partial */

int dml_conv_igemm(const DmlConvDesc* d, void* stream);
/* Rows of the GEMM covered by one partial of `stats` / `bnr_partials` for THIS launch: 48 where the wave-specialised kernel
 * (one persistent workgroup per CU on 144-row tiles: bf16, tile-major weights, N % 128 == 0, enough tiles to fill the chip)
 * takes it -- 144 for a two-plane (f32_split == 2) FORWARD launch on 144-row wave tiles, whose statistics are taken per wave tile
 * (round 6) -- DML_STAT_ROWS otherwise.  Size the partial buffers as ceil(M / rows) * N * 2 floats, pass `rows` to
 * dml_bn_finalize / dml_bn_moments and ceil(M / rows) as `nblocks` to dml_bn_bwd_finalize / dml_bn_bwd_sums. */
int dml_conv_stat_rows(const DmlConvDesc* desc);

/* fp32 tensor x[rows][ld] (C used columns) -> two fp16 planes of s * x, s the power of two that puts max|x| into [2^14, 2^15):
 * hi = fp16(s x), lo = fp16(s x - hi), both round-to-nearest-even, s x = hi + lo up to 2^-22 relative.  planes[0 .. ] = hi,
 * planes[plane_stride ..] = lo (fp16 elements); layout 0: row pitch ldp, element (r, c) at r * ldp + c; layout 1: tile-major
 * [rows / 64][C / 32][64][32] (rows % 64 == 0, C % 32 == 0: the weight operand of the LDS-DMA kernels).  `work`: >= 1025 floats of
 * scratch owned by this tensor; work[1024] receives 1 / s (DmlConvDesc.x_unscale / w_unscale).  Two launches (per-workgroup
 * maxima, then the split: no atomics, no memset, deterministic) -- or one, with amax_known = 1: the maximum over work[0 .. 1024)
 * then already is max |x| (any upper bound within a few binades works: fp16 keeps 11 bits down to 2^-14 of the scaled range),
 * collected by the tensor's producer (dml_bn_apply / dml_bn_bwd_apply, `amax` = work: order-independent atomic maxima spread over
 * those 1024 words, which the caller zeroed).
 * Non-finite values propagate (s becomes 0 or NaN). */
int dml_h2_split(const float* x, int64_t rows, int32_t C, int32_t ld, void* planes, int64_t plane_stride, int32_t ldp,
                 int32_t layout, float* work, int32_t amax_known, void* stream);

/* The same for `count` tensors in two launches; `table_device` is a DEVICE array (the tensors' own constraints as above, checked by the
 * caller: this entry point does not see the descriptors). */
typedef struct DmlH2Desc {
    const float* x;
    void* planes;
    float* work;
    int64_t rows, plane_stride;
    int32_t C, ld, ldp, layout;
} DmlH2Desc;
int dml_h2_split_table(const DmlH2Desc* table_device, int count, void* stream);

typedef struct DmlWgradDesc {
    const void* x;        /* conv input [B,Hi,Wi,C], pitch ldx                                           */
    const void* dy;       /* output gradient [B,Ho,Wo,N], pitch ldy                                      */
    float* dw;            /* fp32 weight gradient [N][R][S][Cm], accumulated (+=): atomics, or via `ws`  */
    int32_t B, Hi, Wi, C, ldx;
    int32_t Ho, Wo, N, ldy;
    int32_t R, S, stride, dil, pad;
    int32_t dtype;
    int32_t splitk;       /* number of slices of the pixel dimension (>=1); 0 = pick automatically      */
    int32_t Cm;           /* un-padded input channels of dw ([N][R][S][Cm]); 0 = C.  Cm < C needs `ws`        */
    float* ws;            /* optional workspace: slices store partials [splitk][N][R*S*C] with plain stores  */
    int64_t ws_elems;     /* and a second kernel folds them into dw (no fp32 atomics); capacity in floats     */
    int32_t f32_split;    /* dtype DML_F32 only: products through the three-term bf16 split (DmlConvDesc.f32_split) */
    int32_t reserved;
    /* f32_split == 2 (dtype DML_F32): both operands as two fp16 planes of the scaled tensors (dml_h2_split, layout 0, the pitches
     * of x / dy), three MFMAs per block on the fp16 matrix cores (DmlConvDesc.x_planes has the arithmetic); NULL planes or
     * misaligned shapes: the three-term split on x / dy -- except that x == x_planes (dy == dy_planes) declares the operand
     * planes-only: DML_EUNSUPPORTED instead of the fallback, as for DmlConvDesc. */
    const void* x_planes;
    const void* dy_planes;
    const float* x_unscale;
    const float* dy_unscale;
    int64_t x_plane_stride, dy_plane_stride;
} DmlWgradDesc;

int dml_conv_wgrad(const DmlWgradDesc* d, void* stream);

/* Weight gradients of several layers in ONE launch: the 48 x 48 layers have 4-9 output tiles each, so a launch of
 * its own needs ~28 pixel slabs per tile to fill 256 CUs and every slab is a 256 KB fp32 partial written and re-read
 * (64 + 64 MB per layer).  Grouped, the same workgroups cover several layers with a few slabs each.  `descs`: HOST
 * array of n <= 12 pointers to descriptors that pass dml_conv_wgrad_group_eligible (bf16, N % 256 == 0, >= 4 output
 * tiles, 31-bit addressable); their ws / ws_elems / splitk are ignored, `ws` (ws_elems floats) is partitioned here. */
int dml_conv_wgrad_group_eligible(const DmlWgradDesc* d);
int dml_conv_wgrad_group(const DmlWgradDesc* const* descs, int n, float* ws, int64_t ws_elems, void* stream);

/* master fp32 weight [N][RS][Cm] -> compute copy [N][RS][Cp] (zero padded, dtype) and, if wt != NULL,
 * the transposed copy wt[Cp][RS][N] used by the data gradient. */
int dml_prep_weight(const float* w_master, void* w, void* wt, int N, int RS, int Cm, int Cp, int dtype,
                    void* stream);
/* The same for every convolution of a model in one launch; `descs_device` is a DEVICE array. */
typedef struct DmlPrepDesc {
    const float* src;     /* master [N][RS][Cm] */
    void* w;              /* [N][RS][Cp] */
    void* wt;             /* [Cp][RS][N] or NULL */
    int32_t N, RS, Cm, Cp;
    /* tile-major copies for the LDS-DMA kernels (DmlConvDesc.w_tiled): the matrix [rows][K] (w: rows = N, K = RS * Cp; wt: rows =
     * Cp, K = RS * N) is stored as [rows / 64][K / 32][64][32], so that the 16 filter rows x 64 bytes one DMA instruction
     * fetches per K step are ONE contiguous KB.  Requires rows % 64 == 0 and K % 32 == 0 (w: Cp % 32 == 0; wt: N % 32 == 0). */
    int32_t w_tiled, wt_tiled;
} DmlPrepDesc;
int dml_prep_weights(const DmlPrepDesc* descs_device, int count, int dtype, void* stream);
/* dst[N][RS][Cm] += src[N][RS][Cp] (drops the padding channels of a padded weight gradient). */
int dml_unpad_wgrad(const float* src, float* dst, int N, int RS, int Cm, int Cp, void* stream);
/* bias gradient: db[n] += sum_m dy[m][n]  (network/utils.py:23) */
int dml_bias_grad(const void* dy, float* db, int64_t M, int N, int ldy, int dtype, void* stream);
/* the same as a fixed-order sum (bitwise reproducible): per-workgroup partial sums in ws (>= 1024 * N floats), then a fold.
 * dml_bias_grad itself combines its workgroups with fp32 atomics. */
int dml_bias_grad_ws(const void* dy, float* db, int64_t M, int N, int ldy, int dtype, float* ws, int64_t ws_elems, void* stream);

/* x[B,C,H,W] fp32 (the reference's input layout) -> NHWC with C padded to Cp, dtype. */
int dml_pack_input(const float* x_nchw, void* y_nhwc, int B, int C, int H, int W, int Cp, int dtype,
                   void* stream);
/* Space-to-depth form of a k x k stride-2 convolution with padding (k - 1) / 2 (odd) on few channels -- the stem, 7x7 s2 on the 3
 * image channels (backbone/resnet.py:139): x2[B][H/2][W/2][4 C] fp32 with channel (dy * 2 + dx) * C + c = x[b][c][2 y2 + dy][2 x2 + dx]
 * (H, W even), w2[N][k2][k2][4 C] with k2 = (k + 1) / 2 holding w[n][2 r2 + dy - 1][2 s2 + dx - 1][c] (zero outside 0 .. k - 1).  The
 * convolution is then a k2 x k2 STRIDE-1 one with padding ((k - 1) / 2 + 1) / 2 on x2 (DmlConvDesc: Ho = H / 2, Wo = W / 2 given
 * explicitly): the same products on K = 4 C k2^2 (192) instead of 8 k^2 (392) for the image padded to 8 channels.
 * dml_s2d_wgrad adds a weight gradient computed in that form (dw2[N][k2][k2][4 C]) to the parameter's dw[N][k][k][C]. */
int dml_pack_input_s2d(const float* x_nchw, float* x2, int B, int C, int H, int W, void* stream);
int dml_s2d_weights(const float* w, float* w2, int N, int k, int C, void* stream);
/* Sub-filter of a prepared fp32 weight copy [rows][taps_src][n]: dst[row][j][:] = src[row][t_j][:] for j < ntaps <= 4 (n % 4 == 0,
 * 16-byte aligned).  The data gradient of a stride-2 convolution (backbone/resnet.py:171-193 of the reference: layer2.0 / layer3.0)
 * runs as one stride-1 launch per pixel-parity class on the taps that class sees (DmlConvDesc.sub_grid); this builds the class's
 * operand from the transposed copy wt[C][R S][N] of dml_prep_weights. */
int dml_gather_taps(const float* src, float* dst, int rows, int taps_src, int n, int ntaps, int t0, int t1, int t2, int t3,
                    void* stream);
int dml_s2d_wgrad(const float* dw2, float* dw, int N, int k, int C, void* stream);

/* ------------------------------------------------------------------------------------------------
 * BatchNorm2d (112 instances; train: batch statistics, eval: running statistics).
 * ---------------------------------------------------------------------------------------------- */
/* Chan-merge the conv epilogue partials -> mean / biased var -> scale = gamma*invstd, shift = beta,
 * save_mean = mean (required; dml_bn_apply subtracts it before scaling so that low-variance channels do
 * not cancel); updates running stats with the unbiased variance and `momentum` exactly as
 * nn.BatchNorm2d does; saves invstd for the backward.  `partials` is scratch: the call may fold it in place
 * (large feature maps), so it is not valid input for a second call. */
/* stat_rows: rows of the GEMM covered by one partial = dml_conv_stat_rows() of the launch that wrote them (DML_STAT_ROWS for
 * dml_bn_stats). */
int dml_bn_finalize(float* partials, int64_t M, int N, int stat_rows, const float* gamma, const float* beta,
                    float* running_mean, float* running_var, float momentum, float eps,
                    float* scale, float* shift, float* save_mean, float* save_invstd, void* stream);
/* Synchronised BatchNorm (statistics over all ranks' samples; anomaly/lib/nn/modules/batchnorm.py:56-139 of the
 * reference -- biased variance for the normalisation, unbiased over the global count for the running estimate).
 * The library stays collective-free: the caller all-gathers / all-reduces the small double buffers (RCCL).
 *   forward : dml_bn_moments(partials -> moments[N][2] = (mean, M2) of this rank's M rows)
 *             -> all_gather -> dml_bn_finalize_moments(moments[ranks][N][2], every rank holding M_each rows)
 *   backward: dml_bn_bwd_sums(partials -> sums[N][2] = (sum g, sum g*xhat); dgamma/dbeta += the LOCAL sums)
 *             -> all_reduce(sum) -> dml_bn_bwd_coef(sums, M_total) -> dml_bn_bwd_apply as usual. */
int dml_bn_moments(float* partials, int64_t M, int N, int stat_rows, double* moments, void* stream);
int dml_bn_finalize_moments(const double* moments, int ranks, int64_t M_each, int N, const float* gamma,
                            const float* beta, float* running_mean, float* running_var, float momentum,
                            float eps, float* scale, float* shift, float* save_mean, float* save_invstd,
                            void* stream);
int dml_bn_bwd_sums(float* partials, int nblocks, int N, double* sums, float* dgamma, float* dbeta,
                    void* stream);
int dml_bn_bwd_coef(const double* sums, int64_t M_total, int N, const float* gamma, const float* save_mean,
                    const float* save_invstd, float* coef, void* stream);
/* Standalone statistics for tensors that were not produced by dml_conv_igemm (writes the same
 * partial format). */
int dml_bn_stats(const void* y, float* partials, int64_t M, int N, int ldy, int dtype, void* stream);
/* eval mode: scale = gamma/sqrt(running_var+eps), shift = beta; pass running_mean as `mean` to dml_bn_apply. */
int dml_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, float eps, float* scale, float* shift, int N,
                       void* stream);
/* the same for a whole network in one launch: `table` is a DEVICE array of `count` descriptors */
typedef struct DmlBnEvalDesc {
    const float* gamma; const float* beta; const float* running_var; float* scale; float* shift;
    int32_t N; float eps;
} DmlBnEvalDesc;
int dml_bn_eval_coeffs_table(const DmlBnEvalDesc* table, int count, void* stream);
/* z = act((y - mean)*scale + shift [+ res]) with optional inverted dropout (network/utils.py:354).
 * y, res, z have independent pitches.  `mask` (optional): one bit per element, z > 0, one BYTE per 16-byte vector of z --
 * DML_BF16: mask[m*(N/8) + c/8] bit c%8; DML_F32: mask[m*(N/4) + c/4] bit c%4 -- the backward passes then read 1 byte
 * instead of 16 bytes of z.  `amax` (optional): 1024 floats the caller zeroed; their maximum afterwards is max |z| (one
 * order-independent atomic maximum per workgroup, spread over the words: dml_h2_split, amax_known).
 * `planes` (optional, DML_F32): the output also as the two fp16 planes a conv of the f16x2 mode reads (dml_h2_split's arithmetic and
 * layout 0: hi at planes[m * ldp + c], lo `plane_stride` elements further), scaled by 1 / unscale[0], a power of two the caller
 * fixed BEFORE this launch from a bound on |z| (dml_h2_bound_bn) -- no dml_h2_split pass over z then; z may be NULL when nothing
 * reads the fp32 tensor.  `res_unscale` (optional, DML_F32; ABI 4): the residual operand exists as fp16 planes only -- `res` then
 * points at its hi plane (pitch ldres, lo plane `res_plane_stride` elements further) and the value added is (hi + lo) *
 * res_unscale[0], exactly what the convolutions reading those planes see. */
int dml_bn_apply(const void* y, const void* res, void* z, const float* scale, const float* shift,
                 const float* mean, uint8_t* mask, int64_t M, int N, int ldy, int ldres, int ldz, int relu,
                 int dtype, float drop_p, uint64_t drop_seed, float* amax, void* planes, int64_t plane_stride,
                 int32_t ldp, const float* unscale, int64_t res_plane_stride, const float* res_unscale, void* stream);
/* backward, pass 1: per-channel sums of g = dz*[z>0]*gscale and g*xhat -> partials[blocks][N][2]
 * ([z>0] from `mask` when given, else from z).
 * returns the number of partial rows through *nblocks (host int).  `gmax` (optional): 1024 zeroed floats, their maximum
 * afterwards is max |g| (as `amax` of dml_bn_apply; input of dml_h2_bound_bn_bwd). */
int dml_bn_bwd_reduce(const void* dz, const void* y, const void* z, const uint8_t* mask, const float* save_mean,
                      const float* save_invstd, float* partials, int64_t M, int N, int lddz, int ldy,
                      int ldz, int relu, float gscale, int dtype, int* nblocks, float* gmax, void* stream);
/* backward, pass 1b: fold partials, write dgamma/dbeta (+=) and the per-channel coefficients
 * coef[4][N] with dy = coef0*g + coef1*(y - coef3) + coef2  (coef3 = batch mean).
 * M = 0 selects a layer that normalised with FIXED statistics (BatchNorm2d.eval() inside a training step, the
 * reference's main_self_distillation.py:432-435): save_mean / save_invstd are then the running statistics, the two
 * correction terms are zero (coef1 = coef2 = 0) and dgamma / dbeta are unchanged.
 * With more than 2048 partial rows they are first folded in place (two coalesced stages): `partials` is clobbered. */
int dml_bn_bwd_finalize(float* partials, int nblocks, int64_t M, int N, const float* gamma,
                        const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                        float* coef, void* stream);
/* backward, pass 2: dy = coef0*g + coef1*(y - coef3) + coef2; optionally dres (+)= g for the identity branch.  `amax`
 * (optional): max |dy|, as in dml_bn_apply.  `planes` / `plane_stride` / `ldp` / `unscale` (optional, DML_F32): dy as fp16 planes,
 * as in dml_bn_apply (scale from dml_h2_bound_bn_bwd); dy may then be NULL. */
int dml_bn_bwd_apply(const void* dz, const void* y, const void* z, const uint8_t* mask, const float* coef, void* dy,
                     void* dres, int64_t M, int N, int lddz, int ldy, int ldz, int lddy, int lddres,
                     int relu, float gscale, int dres_accum, int dtype, float* amax, void* planes, int64_t plane_stride,
                     int32_t ldp, const float* unscale, void* stream);
/* Scale of a batch-statistics BatchNorm output's fp16 planes from a BOUND on its magnitude, available before the tensor is
 * written: every normalised element obeys |y - mean| * invstd <= sqrt(count) (count = elements per channel over all ranks that
 * share the statistics), so |z| <= max_c (|gamma_c| sqrt(count) + |beta_c|) * mult + max |res| (`mult`: 1, or the dropout's
 * 1 / (1 - p); `res_amax`: NULL, or the 1024 amax words of the residual tensor) and, in the backward, |dy| <= max_c (|coef0_c| max|g|
 * + |coef1_c| sqrt(count) / invstd_c + |coef2_c|) (`g_amax`: the words dml_bn_bwd_reduce raised).  work[1024] receives 1 / s, s the
 * power of two that puts the bound into [2^14, 2^15) -- the `unscale` of the apply kernels and DmlConvDesc.x_unscale.  A bound 2^k
 * above the true maximum costs the planes k of the ~12 binades over which an element keeps its full 2^-22 relative precision. */
int dml_h2_bound_bn(const float* gamma, const float* beta, int N, int64_t count, float mult, const float* res_amax,
                    float* work, void* stream);
int dml_h2_bound_bn_bwd(const float* coef, const float* save_invstd, int N, int64_t count, const float* g_amax,
                        float* work, void* stream);
/* The two pairs above as ONE launch each (round 6: 145 dependent one-block launches per train step removed).  Every block of the
 * finalize kernel raises state[0] to the largest bound term of its own channels and takes a ticket in state[1]; the block that
 * arrives last writes work[1024] and leaves both words zero for the next call.  `state`: two zero-initialised 32-bit words owned by
 * this BatchNorm (not shared between launches that may run concurrently).  Results equal the two-call sequence's.  Replaces
 * nn.BatchNorm2d's statistics step in backbone/resnet.py:97-108 of the reference like the calls it fuses. */
int dml_bn_finalize_bound(float* partials, int64_t M, int N, int stat_rows, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, float momentum, float eps,
                          float* scale, float* shift, float* save_mean, float* save_invstd,
                          int64_t count, float mult, const float* res_amax, float* work, uint32_t* state, void* stream);
int dml_bn_bwd_finalize_bound(float* partials, int nblocks, int64_t M, int N, const float* gamma,
                              const float* save_mean, const float* save_invstd, float* dgamma, float* dbeta,
                              float* coef, int64_t count, const float* g_amax, float* work, uint32_t* state, void* stream);
/* dml_h2_bound_bn for `count` residual-free BatchNorms in one launch (`table_device`: DEVICE array; root_count = sqrt(elements per
 * channel) * 1.0001, as the single call computes it). */
typedef struct DmlH2BoundDesc {
    const float* gamma;
    const float* beta;
    float* work;
    int32_t N;
    float root_count, mult;
    int32_t reserved;
} DmlH2BoundDesc;
int dml_h2_bound_bn_table(const DmlH2BoundDesc* table_device, int count, void* stream);
/* ONE scale for a tensor that several batch-statistics BatchNorms write channel slices of (directly, or through a bilinear resize,
 * which keeps the bound): work[1024] = 1 / s from the LARGEST of the `count` entries' bounds (their own `work` fields are ignored).
 * The decoder's concat buffer of network/utils.py:28-32 (low-level projection + upsampled ASPP projection) then exists as fp16 planes
 * only: no dml_h2_split pass (an amax pass + a split pass over 755 MB at 16 x 768 x 768). */
int dml_h2_bound_bn_multi(const DmlH2BoundDesc* table_device, int count, float* work, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pooling (network/backbone/resnet.py:143; network/utils.py:320,326-329).
 * ---------------------------------------------------------------------------------------------- */
int dml_maxpool3x3s2_fwd(const void* x, void* y, uint8_t* argmax, int B, int H, int W, int C, int dtype,
                         void* stream);
int dml_maxpool3x3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int B, int H, int W, int C,
                         int dtype, void* stream);
/* y[b][c] = mean over HW of x[b,:,:,c]  (AdaptiveAvgPool2d(1)); y has dtype. */
int dml_global_avgpool_fwd(const void* x, void* y, int B, int HW, int C, int ldx, int dtype, void* stream);
/* z[b,h,w,c] = v[b][c] (bilinear upsample of a 1x1 map is a broadcast) */
int dml_broadcast_hw(const void* v, void* z, int B, int HW, int C, int ldz, int dtype, void* stream);
/* dv[b][c] = sum over HW of dz[b,:,:,c] */
int dml_reduce_hw(const void* dz, void* dv, int B, int HW, int C, int lddz, int dtype, void* stream);
/* out[b][c] = scale * sum over HW of x[b,:,:,c], x in `dtype`, out ALWAYS fp32.  The ASPP image-pooling branch
 * (network/utils.py:318-329) runs a BatchNorm over only B samples per channel: its input (scale = 1/HW) and its output
 * gradient (scale = 1) are B x C values whose sample-to-sample DIFFERENCES decide 1/sigma and the backward's
 * cancellation, so the bf16 plans keep them unrounded. */
int dml_reduce_hw_f32(const void* x, float* out, int B, int HW, int C, int ldx, int dtype, float scale, void* stream);
/* dx[b,:,:,c] += dv[b][c] / HW */
int dml_avgpool_bwd_add(const void* dv, void* dx, int B, int HW, int C, int lddx, int dtype, void* stream);
/* dx[b,:,:,c] = dv[b][c] / HW.  In bf16 the plan lets this INITIALISE the gradient buffer and the data gradients of the
 * other ASPP branches accumulate on top: added last, a per-pixel term of 1/HW of a pooled gradient is below half an ulp
 * of the running sum at every pixel and would vanish entirely, although it is coherent over the image. */
int dml_avgpool_bwd_set(const void* dv, void* dx, int B, int HW, int C, int lddx, int dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=False (F.interpolate call sites network/utils.py:30,88,329).
 * in_f32 / out_f32 select float storage for that side, otherwise `dtype`.
 * ---------------------------------------------------------------------------------------------- */
int dml_bilinear_fwd(const void* x, void* y, int B, int h, int w, int H, int W, int C, int ldx, int ldy,
                     int dtype, int in_f32, int out_f32, void* stream);
/* dml_bilinear_fwd of an fp32 tensor with the result as the two fp16 planes of the scaled value (hi at `planes`, lo `plane_stride`
 * elements further, pitch ldp, scale = 1 / unscale[0]: dml_h2_split's arithmetic on the value dml_bilinear_fwd would store). */
int dml_bilinear_fwd_planes(const float* x, void* planes, int64_t plane_stride, int ldp, const float* unscale, int B, int h, int w, int H,
                            int W, int C, int ldx, void* stream);
int dml_bilinear_bwd(const void* dy, void* dx, int B, int h, int w, int H, int W, int C, int lddy,
                     int lddx, int dtype, int in_f32, int out_f32, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pixel -> prototype squared-distance head (network/utils.py:92-118; anomaly/models/models.py:636-657)
 * and what the drivers compute from it on the host (test_embedding.py:339-350,428-445;
 * anomaly/eval_ood_traditional.py:301-305).
 * ---------------------------------------------------------------------------------------------- */
/* x: embedding at full resolution, NCHW fp32 (what F.interpolate returns at utils.py:88).
 * logits[b,k,h,w] = -sum_c (x[b,c,h,w]-protos[k][c])^2 (NCHW), features_out = x in NHWC.
 * Any of logits / feats / argmax / dissum may be NULL.  192 B per pixel at K=C=16. */
int dml_proto_dist_fwd(const float* x_nchw, const float* protos, float* logits, float* feats,
                       uint8_t* argmax, float* dissum, int B, int C, int K, int H, int W, void* stream);
/* Fused: bilinear x(H/h) upsample of the low-resolution embedding e[B,h,w,C] (NHWC fp32) + the
 * distance head, writing logits NCHW and features_out NHWC at [H,W]. */
int dml_upsample_dist_fwd(const float* e, const float* protos, float* logits, float* feats,
                          uint8_t* argmax, float* dissum, int B, int h, int w, int C, int K, int H,
                          int W, void* stream);
/* df[b,h,w,c] = -2 * sum_k glogits[b,k,h,w] * (feats[b,h,w,c]-protos[k][c]) (+ gfeats[b,h,w,c]) */
int dml_proto_dist_bwd(const float* glogits, const float* gfeats, const float* feats,
                       const float* protos, float* df, int B, int C, int K, int H, int W, void* stream);

/* Backward of DML loss + distance head + final x4 upsample in one pass, for a loss whose only input is this head's
 * logits: de[B,h,w,C] (dtype) = bilinear^T( d loss / d features ), d loss / d logits exactly as dml_loss_bwd defines it
 * (sums / gout / alpha / n_images / ignore_index as there) with the logits recomputed from feats[B,H,W,C].
 * C = K = 16, H = 4h, W = 4w; anything else returns DML_EUNSUPPORTED (use dml_loss_bwd + dml_proto_dist_bwd +
 * dml_bilinear_bwd).  72 B per full-resolution pixel instead of 394. */
int dml_head_bwd_fused(const float* feats, const int64_t* labels, const double* sums, const float* gout,
                       const float* protos, void* de, int B, int h, int w, int C, int K, int H, int W,
                       int64_t ignore_index, float alpha, float n_images, int dtype, void* stream);

/* preds = argmax_k logits (ties -> lowest k), msp = 1 - max softmax  (test_embedding.py:339-341) */
int dml_argmax_msp(const float* logits, int64_t* preds, float* msp, int B, int K, int H, int W,
                   void* stream);
/* s = -sum_k logit_k, clipped (inclusive: s>=clip -> clip, else s>clip -> clip), then per-image
 * min-max normalised.  work: 2*B floats. */
int dml_dissum_score(const float* logits, float* score, float* work, int B, int K, int H, int W,
                     float clip, int inclusive, void* stream);
/* preds[p] = new_label where -|f_p - proto|^2 > thresh and > max_k logits[k][p] */
int dml_novel_relabel(const float* feats, const float* logits, const float* proto, int64_t* preds,
                      int B, int C, int K, int H, int W, float thresh, int64_t new_label, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DML loss = CE(-dist^2)/n + alpha * VAR/n (anomaly/models/models.py:42-78; live part of
 * utils/loss.py:34-42 is alpha = 0).
 * sums (device, double[4]) = { sum nll over valid px, #valid px, VAR, #correct }.
 * ---------------------------------------------------------------------------------------------- */
int dml_loss_fwd(const float* logits, const int64_t* labels, double* sums, float* block_partials,
                 int B, int K, int H, int W, int64_t ignore_index, void* stream);
/* loss = (sums[0]/sums[1] + alpha*sums[2]) / n_images  (written to *loss, device float).
 * n_images <= 0 (here and in dml_loss_bwd): the image count is read from sums[4] on the device -- the data-parallel
 * caller all-reduces {sums[0..3], local batch} in one 5-double message and never brings the count to the host. */
int dml_loss_finalize(const double* sums, float* loss, float alpha, float n_images, void* stream);
/* glogits = gout * d loss / d logits; gout is a device scalar. */
int dml_loss_bwd(const float* logits, const int64_t* labels, const double* sums, const float* gout,
                 float* glogits, int B, int K, int H, int W, int64_t ignore_index, float alpha,
                 float n_images, void* stream);

/* ------------------------------------------------------------------------------------------------
 * SGD with momentum and weight decay, torch semantics (main_embedding.py:385-388):
 *   g = gscale*g + wd*p ; v = mu*v + g ; p -= lr*v
 * ---------------------------------------------------------------------------------------------- */
int dml_sgd_step(float* p, const float* g, float* v, int64_t n, float lr, float momentum,
                 float weight_decay, float gscale, void* stream);

/* fill / scale helpers used by the host runtime */
int dml_fill_f32(float* p, int64_t n, float value, void* stream);
/* dst[i] = (dst_dtype) src[i] -- the bf16 plans keep the ASPP image-pooling branch (network/utils.py:318-329: a
 * BatchNorm over B x 256 x 1 x 1, i.e. over only B samples) in fp32 storage: with two near-equal samples a bf16-rounded
 * pre-normalisation value makes that layer's gradient blow up by up to 1/sqrt(eps). */
int dml_convert_dtype(const void* src, void* dst, int64_t n, int src_dtype, int dst_dtype, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Input pipeline on the device (the step before the path): the Cityscapes train
 * transform of main_embedding.py:148-157 -- random crop, colour jitter (brightness / contrast /
 * saturation in a random order), horizontal flip, ToTensor, Normalize (utils/ext_transforms.py:222-230,
 * 282-293, 313-322, 357-393, 469-504) -- on a batch of uint8 NHWC frames resident in HBM.  The random
 * draws stay on the host (utils/ext_transforms.py of this package mirrors the reference's order of
 * `random` calls); the kernels take one DmlAugSample per image.  Pixel arithmetic is Pillow's, bit for
 * bit: L = (19595 R + 38470 G + 7471 B + 0x8000) >> 16; blend(a, b, f) = (uint8)(a + f*(b - a)) in
 * float32 without fused multiply-add, truncating, clipped to [0,255] when f is outside [0,1]; the
 * contrast pivot is int(mean(L over the crop, after the ops that precede it) + 0.5).
 * ---------------------------------------------------------------------------------------------- */
typedef struct DmlAugSample {
    int32_t i, j;       /* crop origin (row, column) in the source frame */
    int32_t flip;       /* horizontal flip (applied after the jitter, ext_transforms.py:229) */
    int32_t n_ops;      /* 0..3 jitter ops */
    int32_t op[3];      /* 0 brightness, 1 contrast, 2 saturation, in application order */
    float factor[3];
} DmlAugSample;
/* lsum[b] (device uint32, zeroed by the call) = sum of L over sample b's crop as seen by its contrast
 * op (after the ops that precede it); untouched semantics for samples without a contrast op. */
int dml_aug_contrast_sum(const uint8_t* img, const DmlAugSample* samples, uint32_t* lsum, int B, int H,
                         int W, int th, int tw, void* stream);
/* out_img[B,3,th,tw] fp32 = ((jittered, flipped crop)/255 - mean)/std; out_lbl[B,th,tw] int64 = the
 * same crop / flip of lbl[B,H,W] (uint8).  lsum from dml_aug_contrast_sum on the same stream. */
int dml_aug_apply(const uint8_t* img, const uint8_t* lbl, const DmlAugSample* samples,
                  const uint32_t* lsum, float* out_img, int64_t* out_lbl, int B, int H, int W, int th,
                  int tw, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                  void* stream);
/* Same, with the dataset's label encoding folded in (datasets/cityscapes.py:132-154 of the reference:
 * Cityscapes.encode_target = raw id -> train id -> unknown classes to 255, higher ids shifted down; pointwise, so it
 * commutes with crop / flip): out_lbl = lut[raw], out_lbl_true = lut_true[raw] (optional second output, the reference's
 * `target_true`).  lut / lut_true: device uint8[256]. */
int dml_aug_apply_encoded(const uint8_t* img, const uint8_t* lbl, const DmlAugSample* samples,
                          const uint32_t* lsum, float* out_img, int64_t* out_lbl, int B, int H, int W,
                          int th, int tw, float mean0, float mean1, float mean2, float std0, float std1,
                          float std2, const uint8_t* lut, const uint8_t* lut_true, int64_t* out_lbl_true,
                          void* stream);
/* The same table applied to a whole uint8 label tensor (validation path: no crop / jitter). */
int dml_label_encode(const uint8_t* raw, int64_t n, const uint8_t* lut, const uint8_t* lut_true,
                     int64_t* out, int64_t* out_true, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pyramid-pooling decoder of the anomaly model (anomaly/models/models.py:586-687,
 * eval_ood_traditional.py:198-210 of the reference), inference only.
 * ---------------------------------------------------------------------------------------------- */
/* nn.AdaptiveAvgPool2d(S) on NHWC x[B,H,W,C] (pitch ldx) -> y[B,S,S,C] (dense): bin i = [floor(i*H/S), ceil((i+1)*H/S)).
 * ws: caller-owned fp32 workspace of dml_adaptive_avgpool_ws_elems(...) elements (two deterministic stages). */
int64_t dml_adaptive_avgpool_ws_elems(int B, int H, int W, int C, int S);
int dml_adaptive_avgpool_fwd(const void* x, void* y, float* ws, int B, int H, int W, int C, int ldx, int S,
                             int dtype, void* stream);
/* out[m][k] = -sum_c (emb[m][c] - protos[k][c])^2 for k < K on an NHWC fp32 embedding with Kp (<= 32) channels
 * (pad channels zero in emb and protos[K][Kp]); out channels K..Kp-1 = 0.  models.py:633-657: the distance is
 * taken at 1/8 resolution, BEFORE the upsample. */
int dml_proto_dist_nhwc(const float* emb, const float* protos, float* out, int64_t M, int K, int Kp, int lde,
                        int ldo, void* stream);
/* dst[B,C,H,W] (+)= alpha * bilinear(src[B,h,w,ld] channels 0..C-1), align_corners = False -- F.interpolate to
 * segSize (models.py:659-668) with the multi-scale mean `scores = scores + scores_tmp / n` folded in. */
int dml_upsample_nhwc_to_nchw(const float* src, float* dst, int B, int h, int w, int ld, int C, int H, int W,
                              float alpha, int accumulate, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Streaming segmentation metrics on the device (the step after the path):
 * hist[n*t + p] += 1 for every pixel with 0 <= t < n (and 0 <= p < n), the np.bincount of
 * metrics/stream_metrics.py:49-55.  hist is a device int64[n*n] that the call accumulates into;
 * n <= 64.  The scores of :57-83 are computed from the n x n matrix on the host.
 * ---------------------------------------------------------------------------------------------- */
int dml_confusion_update(const int64_t* label_true, const int64_t* label_pred, int64_t* hist,
                         int64_t count, int n_classes, void* stream);

/* Few-shot prototype extraction (the recipe at test_embedding.py:413-425 of the reference: mean of features_out over
 * the pixels of one class): sums[c] (device double[C], zeroed by the call) = sum of feats[p][c] over pixels with
 * labels[p] == class_id, count (device uint64) = their number.  C <= 32. */
int dml_class_feature_sum(const float* feats, const int64_t* labels, int64_t n_px, int C, int64_t class_id,
                          double* sums, unsigned long long* count, void* stream);

/* Pixel-level OOD measures (anomaly/anom_utils.py:25-78 as called by eval_ood_traditional.py:128-148):
 * scores = -conf; positives = pixels whose label is one of out_labels (host array, n_out <= 8), negatives = the
 * rest; pixels with mask[i] == 0 are left out (mask optional).  result (device double[5]) = { AUROC, AUPR,
 * FPR at recall_level, #positives, #negatives }; the three measures are NaN when either class is empty (the
 * reference prints a notice and skips the image).  work: >= dml_ood_workspace_bytes(n) bytes, 256-byte aligned.
 * Device-side radix sort + rank statistics; nothing is copied to the host. */
int64_t dml_ood_workspace_bytes(int64_t n);
int dml_ood_measures(const float* conf, const int64_t* seg_label, const uint8_t* mask, int64_t n,
                     const int64_t* out_labels, int n_out, double recall_level, void* work,
                     int64_t work_bytes, double* result, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Native replay of a static launch list.  The reference dispatches one Python call per layer per step
 * (nn.Module.forward / autograd, network/utils.py:84-118, backbone/resnet.py:95-115); the drop-in's plan is a fixed
 * list of this library's own entry points, and dml_plan_run walks a packed copy of it in one call -- same launches,
 * same order, same streams, no interpreter in between.  HOST-side only: `ops` is a host array.
 *   fn      index from dml_plan_fn_id("dml_...") (any entry point above whose last parameter is the stream)
 *   args    one 64-bit word per parameter in declaration order, stream excluded: pointers and integers as they are
 *           (sign-extended), float in the low 32 bits, double as its 64 bits; host pointers to descriptors
 *           (DmlConvDesc / DmlWgradDesc) must stay valid for the call
 *   indirect bit k set: args[k] is the HOST address of an int32 read when the op is issued (the *nblocks that
 *           dml_bn_bwd_reduce wrote a few ops earlier)
 *   stream  0 = `stream`, 1 = `side_stream` (weight gradients); wait = 1: the side stream first waits for everything
 *           enqueued on `stream` so far (one of `events`, a ring of hipEvent_t, is recorded there)
 * Ops [first, last) are issued; on failure the index goes to *failed_op and the op's code is returned.
 * ---------------------------------------------------------------------------------------------- */
#define DML_PLAN_MAX_ARGS 24
typedef struct DmlPlanOp {
    int32_t fn, nargs;
    int32_t stream, wait;
    uint32_t indirect, reserved;
    uint64_t args[DML_PLAN_MAX_ARGS];
} DmlPlanOp;
int dml_plan_fn_id(const char* name);
int dml_plan_fn_nargs(int fn);
int dml_plan_run(const DmlPlanOp* ops, int first, int last, void* stream, void* side_stream, void* const* events,
                 int n_events, int* failed_op);
/* dml_plan_run with MARKS: after op marks[k] (ascending op indices; those outside [first, last) are ignored) mark_events[2 k] is
 * recorded on `stream` and, with a side stream, mark_events[2 k + 1] on `side_stream` (hipEvent_t handles).  Replaces the reference's
 * per-step nn.DataParallel reduce (main_embedding.py:425,438-439) together with dmlnet/parallel.py: the reducer's communication stream
 * waits on these events, so the host enqueues the whole backward in one call. */
int dml_plan_run_marks(const DmlPlanOp* ops, int first, int last, void* stream, void* side_stream, void* const* events,
                       int n_events, const int32_t* marks, int n_marks, void* const* mark_events, int* failed_op);

#ifdef __cplusplus
}
#endif
#endif /* DMLNET_HIP_H */
