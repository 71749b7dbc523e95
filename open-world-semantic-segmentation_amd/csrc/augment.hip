// Input pipeline on the device: random crop + colour jitter + horizontal flip + ToTensor + Normalize for a batch of
// uint8 NHWC frames (the Cityscapes train transform of the reference, main_embedding.py:148-157 with
// utils/ext_transforms.py:222-230,282-293,313-322,357-393,469-504).  HBM-bound: 4 bytes read, 20 written per pixel.
// The arithmetic is Pillow's (ImageEnhance / Image.blend / convert("L")), reproduced bit for bit -- see the header.
#include "common.h"

namespace {

constexpr int TPB = 256;          // threads per workgroup
constexpr int PPT = 4;            // consecutive output pixels per thread
constexpr int SEG = TPB * PPT;    // output pixels of one image row per workgroup

struct Px { uint32_t r, g, b; };

__device__ __forceinline__ uint32_t luma(const Px p) { return (p.r * 19595u + p.g * 38470u + p.b * 7471u + 0x8000u) >> 16; }

// Image.blend: float32 a + f*(b - a), multiply and add rounded separately (no contraction), truncation to uint8;
// clipped first when f is outside [0, 1]
__device__ __forceinline__ uint32_t blend1(uint32_t a, uint32_t b, float f, bool interp) {
    const float t = __fadd_rn((float)a, __fmul_rn(f, __fsub_rn((float)b, (float)a)));
    if (interp) return (uint32_t)t;
    return t <= 0.f ? 0u : (t >= 255.f ? 255u : (uint32_t)t);
}
__device__ __forceinline__ Px blend(const Px d, const Px x, float f) {
    const bool interp = f >= 0.f && f <= 1.f;
    Px o;
    o.r = blend1(d.r, x.r, f, interp);
    o.g = blend1(d.g, x.g, f, interp);
    o.b = blend1(d.b, x.b, f, interp);
    return o;
}
__device__ __forceinline__ Px apply_op(const Px x, int op, float f, uint32_t pivot) {
    Px d;
    if (op == 0) d.r = d.g = d.b = 0u;                      // brightness: towards black
    else if (op == 1) d.r = d.g = d.b = pivot;              // contrast: towards the mean luminance
    else d.r = d.g = d.b = luma(x);                         // saturation: towards the pixel's own luminance
    return blend(d, x, f);
}

// The source bytes of a row segment are contiguous (read mirrored when flipped): stage them in LDS as whole dwords
// from the 4-byte-aligned address below the first byte, then every thread picks its PPT pixels.
// Returns the byte offset of source pixel 0 inside `sm`.
__device__ __forceinline__ int stage_row(const uint8_t* __restrict__ img, uint32_t* sm, const DmlAugSample& s, int b, int y,
                                         int x0, int n, int H, int W, int tw, const uint8_t* img_end) {
    const int c0 = s.flip ? (tw - x0 - n) : x0;             // output columns x0.. <- source columns c0.. (or mirrored)
    const uint8_t* src = img + (((int64_t)b * H + s.i + y) * W + s.j + c0) * 3;
    const int shift = (int)(reinterpret_cast<uintptr_t>(src) & 3);
    const uint32_t* src4 = reinterpret_cast<const uint32_t*>(src - shift);
    const int nd = (n * 3 + shift + 3) >> 2;
    for (int t = threadIdx.x; t < nd; t += TPB) {
        const uint8_t* q = reinterpret_cast<const uint8_t*>(src4 + t);
        uint32_t w;
        if (q + 4 <= img_end) {
            w = src4[t];
        } else {                                            // last dword of the whole batch: stay inside the buffer
            w = 0;
            for (int k = 0; k < 4 && q + k < img_end; ++k) w |= (uint32_t)q[k] << (8 * k);
        }
        sm[t] = w;
    }
    __syncthreads();
    return shift;
}
__device__ __forceinline__ Px pick(const uint32_t* sm, int byte_off) {
    const uint8_t* p = reinterpret_cast<const uint8_t*>(sm) + byte_off;
    Px o;
    o.r = p[0];
    o.g = p[1];
    o.b = p[2];
    return o;
}

__global__ __launch_bounds__(TPB) void aug_contrast_sum_kernel(const uint8_t* __restrict__ img,
                                                               const DmlAugSample* __restrict__ samples,
                                                               uint32_t* __restrict__ lsum, int H, int W, int th, int tw) {
    __shared__ uint32_t sm[SEG * 3 / 4 + 2];
    __shared__ uint32_t red[TPB / 64];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * SEG;
    const DmlAugSample s = samples[b];
    int cpos = -1;
    for (int k = 0; k < s.n_ops; ++k)
        if (s.op[k] == 1) cpos = k;
    if (cpos < 0) return;                                   // uniform per block
    const int n = min(SEG, tw - x0);
    const int shift = stage_row(img, sm, s, b, y, x0, n, H, W, tw, img + (int64_t)gridDim.z * H * W * 3);
    uint32_t l = 0;
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
        const int k = threadIdx.x * PPT + e;                // the sum does not care about the mirror
        if (k < n) {
            Px p = pick(sm, shift + k * 3);
            for (int q = 0; q < cpos; ++q) p = apply_op(p, s.op[q], s.factor[q], 0u);
            l += luma(p);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int k = 0; k < TPB / 64; ++k) t += red[k];
        atomicAdd(lsum + b, t);
    }
}

__global__ __launch_bounds__(TPB) void aug_apply_kernel(const uint8_t* __restrict__ img, const uint8_t* __restrict__ lbl,
                                                        const DmlAugSample* __restrict__ samples,
                                                        const uint32_t* __restrict__ lsum, float* __restrict__ out_img,
                                                        int64_t* __restrict__ out_lbl, int H, int W, int th, int tw,
                                                        float m0, float m1, float m2, float s0, float s1, float s2,
                                                        const uint8_t* __restrict__ lut, const uint8_t* __restrict__ lut_true,
                                                        int64_t* __restrict__ out_lbl_true) {
    __shared__ uint32_t sm[SEG * 3 / 4 + 2];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * SEG;
    const DmlAugSample s = samples[b];
    const int n = min(SEG, tw - x0);
    const int shift = stage_row(img, sm, s, b, y, x0, n, H, W, tw, img + (int64_t)gridDim.z * H * W * 3);
    const int xl = threadIdx.x * PPT;
    if (xl >= n) return;
    // ImageStat mean in double, + 0.5, truncated (ImageEnhance.Contrast)
    const uint32_t pivot = (uint32_t)(int)((double)lsum[b] / (double)((int64_t)th * tw) + 0.5);
    float v[3][PPT];
    int64_t lab[PPT], lab_true[PPT];
    const int64_t lrow = ((int64_t)b * H + s.i + y) * W + s.j;
#pragma unroll
    for (int e = 0; e < PPT; ++e) {
        const int k = xl + e;
        if (k < n) {
            Px p = pick(sm, shift + (s.flip ? (n - 1 - k) : k) * 3);
            for (int q = 0; q < s.n_ops; ++q) p = apply_op(p, s.op[q], s.factor[q], pivot);
            // F.to_tensor: float / 255; F.normalize: sub then div
            v[0][e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.r, 255.f), m0), s0);
            v[1][e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.g, 255.f), m1), s1);
            v[2][e] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)p.b, 255.f), m2), s2);
            if (lbl != nullptr) {
                const int x = x0 + k;
                const uint8_t raw = lbl[lrow + (s.flip ? (tw - 1 - x) : x)];
                // dataset label encoding (Cityscapes.encode_target) is a pointwise table, so it commutes with crop / flip
                lab[e] = (int64_t)(lut != nullptr ? lut[raw] : raw);
                lab_true[e] = (int64_t)(lut_true != nullptr ? lut_true[raw] : raw);
            }
        }
    }
    const int64_t plane = (int64_t)th * tw;
    const int64_t o = (int64_t)y * tw + x0 + xl;
    float* oi = out_img + (int64_t)b * 3 * plane + o;
    const bool vec = (tw & 3) == 0 && xl + PPT <= n && (reinterpret_cast<uintptr_t>(out_img) & 15) == 0;
    if (vec) {
#pragma unroll
        for (int c = 0; c < 3; ++c)
            *reinterpret_cast<float4*>(oi + c * plane) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
    } else {
        for (int e = 0; e < PPT && xl + e < n; ++e)
            for (int c = 0; c < 3; ++c) oi[c * plane + e] = v[c][e];
    }
    if (lbl != nullptr && out_lbl != nullptr) {
        int64_t* ol = out_lbl + (int64_t)b * plane + o;
        if (vec && (reinterpret_cast<uintptr_t>(out_lbl) & 15) == 0) {
            typedef long long ll2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<ll2*>(ol) = (ll2){lab[0], lab[1]};
            *reinterpret_cast<ll2*>(ol + 2) = (ll2){lab[2], lab[3]};
        } else {
            for (int e = 0; e < PPT && xl + e < n; ++e) ol[e] = lab[e];
        }
    }
    if (lbl != nullptr && out_lbl_true != nullptr) {
        int64_t* ol = out_lbl_true + (int64_t)b * plane + o;
        for (int e = 0; e < PPT && xl + e < n; ++e) ol[e] = lab_true[e];
    }
}

__global__ __launch_bounds__(256) void label_encode_kernel(const uint8_t* __restrict__ raw, int64_t n,
                                                           const uint8_t* __restrict__ lut, const uint8_t* __restrict__ lut_true,
                                                           int64_t* __restrict__ out, int64_t* __restrict__ out_true) {
    __shared__ uint8_t t[2][256];
    t[0][threadIdx.x] = lut[threadIdx.x];
    t[1][threadIdx.x] = lut_true != nullptr ? lut_true[threadIdx.x] : (uint8_t)threadIdx.x;
    __syncthreads();
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const uint8_t r = raw[i];
        out[i] = (int64_t)t[0][r];
        if (out_true != nullptr) out_true[i] = (int64_t)t[1][r];
    }
}

}  // namespace

extern "C" int dml_aug_contrast_sum(const uint8_t* img, const DmlAugSample* samples, uint32_t* lsum, int B, int H, int W,
                                    int th, int tw, void* stream) {
    if (!img || !samples || !lsum || B <= 0 || th <= 0 || tw <= 0 || th > H || tw > W) return DML_EINVAL;
    if ((int64_t)th * tw * 255 >= (1ll << 32)) return DML_EUNSUPPORTED;       // uint32 luminance sum
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemsetAsync(lsum, 0, sizeof(uint32_t) * B, st) != hipSuccess) return DML_EINVAL;
    hipLaunchKernelGGL(aug_contrast_sum_kernel, dim3((tw + SEG - 1) / SEG, th, B), dim3(TPB), 0, st, img, samples, lsum, H, W,
                       th, tw);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_aug_apply(const uint8_t* img, const uint8_t* lbl, const DmlAugSample* samples, const uint32_t* lsum,
                             float* out_img, int64_t* out_lbl, int B, int H, int W, int th, int tw, float mean0, float mean1,
                             float mean2, float std0, float std1, float std2, void* stream) {
    if (!img || !samples || !lsum || !out_img || B <= 0 || th <= 0 || tw <= 0 || th > H || tw > W) return DML_EINVAL;
    if (std0 == 0.f || std1 == 0.f || std2 == 0.f) return DML_EINVAL;
    hipLaunchKernelGGL(aug_apply_kernel, dim3((tw + SEG - 1) / SEG, th, B), dim3(TPB), 0, static_cast<hipStream_t>(stream), img,
                       lbl, samples, lsum, out_img, out_lbl, H, W, th, tw, mean0, mean1, mean2, std0, std1, std2, nullptr, nullptr,
                       nullptr);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_aug_apply_encoded(const uint8_t* img, const uint8_t* lbl, const DmlAugSample* samples, const uint32_t* lsum,
                                     float* out_img, int64_t* out_lbl, int B, int H, int W, int th, int tw, float mean0,
                                     float mean1, float mean2, float std0, float std1, float std2, const uint8_t* lut,
                                     const uint8_t* lut_true, int64_t* out_lbl_true, void* stream) {
    if (!img || !lbl || !samples || !lsum || !out_img || !out_lbl || !lut || B <= 0 || th <= 0 || tw <= 0 || th > H || tw > W)
        return DML_EINVAL;
    if (std0 == 0.f || std1 == 0.f || std2 == 0.f) return DML_EINVAL;
    hipLaunchKernelGGL(aug_apply_kernel, dim3((tw + SEG - 1) / SEG, th, B), dim3(TPB), 0, static_cast<hipStream_t>(stream), img,
                       lbl, samples, lsum, out_img, out_lbl, H, W, th, tw, mean0, mean1, mean2, std0, std1, std2, lut, lut_true,
                       out_lbl_true);
    DML_LAUNCH_CHECK();
    return 0;
}

extern "C" int dml_label_encode(const uint8_t* raw, int64_t n, const uint8_t* lut, const uint8_t* lut_true, int64_t* out,
                                int64_t* out_true, void* stream) {
    if (n == 0) return 0;
    if (!raw || !lut || !out || n < 0 || (out_true && !lut_true)) return DML_EINVAL;
    const int64_t want = (n + 256 * 8 - 1) / (256 * 8);
    hipLaunchKernelGGL(label_encode_kernel, dim3((unsigned)(want < 65536 ? want : 65536)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), raw, n, lut, lut_true, out, out_true);
    DML_LAUNCH_CHECK();
    return 0;
}
