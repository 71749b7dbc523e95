// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, MFMA, LDS).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdlib>

#ifndef DML_GRID_CAP_DEFAULT
#define DML_GRID_CAP_DEFAULT 0
#endif
#include "../../include/dmlnet_hip.h"

#define DML_LAUNCH_CHECK()                        \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

typedef unsigned short bf16_t;   // raw bfloat16 bits

typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) short bf16x8;   // MFMA operand: 8 bf16 in 4 VGPRs
typedef __attribute__((ext_vector_type(4))) short bf16x4;

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// round-to-nearest-even, NaN preserved
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
// two floats -> packed bf16 pair (lo in bits 0-15), round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    const f32x2_hw v = {lo, hi};
    const bf16x2_hw b = __builtin_convertvector(v, bf16x2_hw);
    return __builtin_bit_cast(uint32_t, b);
}

template <typename T> struct Elem;
template <> struct Elem<float> {
    static constexpr int VEC = 4;   // elements per 16-byte vector
    __device__ static __forceinline__ float ld(const float* p) { return *p; }
    __device__ static __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct Elem<bf16_t> {
    static constexpr int VEC = 8;
    __device__ static __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
    __device__ static __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 16-byte vector of T unpacked to floats and back.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
    static constexpr int N = 4;
    __device__ static __forceinline__ void load(const float* p, float (&v)[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    __device__ static __forceinline__ void store(float* p, const float (&v)[4]) {
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    }
};
template <> struct Vec16<bf16_t> {
    static constexpr int N = 8;
    __device__ static __forceinline__ void load(const bf16_t* p, float (&v)[8]) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const uint32_t w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __uint_as_float(w[i] << 16);
            v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ static __forceinline__ void store(bf16_t* p, const float (&v)[8]) {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = pack_bf16x2(v[2 * i], v[2 * i + 1]);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};

// unsigned division by a runtime constant: q = n / d for n < 2^31
struct FastDiv {
    uint32_t mul, shr, d;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    if (d == 1) { f.mul = 0; f.shr = 0; return f; }
    uint32_t l = 0;
    while ((1u << l) < d) ++l;                       // ceil(log2 d)
    const uint64_t p = 31 + l;
    f.mul = (uint32_t)(((1ull << p) + d - 1) / d);
    f.shr = (uint32_t)(p - 32);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv& f) {
    return f.d == 1 ? n : (__umulhi(n, f.mul) >> f.shr);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Workgroups for a grid-stride streaming kernel.  Caps of 2048 and above are "enough to fill the chip" choices, not
// buffer sizes (short-lived workgroups stream faster than persistent ones, see bn.hip).
static inline int grid_for(int64_t work_items, int block, int max_blocks = 256 * 8) {
    constexpr int cap_override = DML_GRID_CAP_DEFAULT;      // (tuning builds: -DDML_GRID_CAP_DEFAULT=n)
    if (cap_override > 0 && max_blocks >= 2048) max_blocks = cap_override;
    int64_t g = (work_items + block - 1) / block;
    if (g < 1) g = 1;
    if (g > max_blocks) g = max_blocks;
    return (int)g;
}
