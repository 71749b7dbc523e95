// Probe (GPU box): what is the "MFMAs alone" ceiling of the two-plane K loop made of -- instruction shape or clock?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_mfma_ceiling.hip -o tools/build/probe_mfma_ceiling && tools/build/probe_mfma_ceiling
// MFMA-only bodies, no memory traffic inside the timed loop, one workgroup per CU, one wave per SIMD (the consumer waves of
// conv_ws_kernel), operands from random or zero fp16 data (DVFS: the chip clocks to its power budget):
//   shape 0: v_mfma_f32_16x16x32_f16 on the product's 144 x 64 wave tile -- 36 accumulators of 4 registers, 108 MFMAs per K step
//            in the product's order (three quads per row group: wh*xh, wh*xl, wl*xh)
//   shape 1: v_mfma_f32_32x32x16_f16 on a 128 x 64 wave tile -- 8 accumulators of 16 registers, 2 k-halves x 3 products = 48
//            MFMAs per K step of 32
// Reported per body: wall time (HIP events), shader cycles per wave (s_memtime), cycles per MFMA, effective clock = cycles / wall,
// TFLOP/s of executed fp16 MFMA work and its fraction of the 2.5 PFLOP/s dense peak.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma16_kernel(const f16x8* __restrict__ src, float* __restrict__ out,
                                                           unsigned long long* __restrict__ cyc, const int iters) {
    const int lane = threadIdx.x & 63;
    f16x8 bh[4], bl[4], ah[9], al[9];
#pragma unroll
    for (int i = 0; i < 4; ++i) { bh[i] = src[lane + 64 * i]; bl[i] = src[lane + 64 * (4 + i)]; }
#pragma unroll
    for (int j = 0; j < 9; ++j) { ah[j] = src[lane + 64 * (8 + j)]; al[j] = src[lane + 64 * (17 + j)]; }
    f32x4 acc[4][9];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[i], ah[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh[i], al[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl[i], ah[j], acc[i][j], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    f32x4 s = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 9; ++j) s += acc[i][j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
    if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void mfma32_kernel(const f16x8* __restrict__ src, float* __restrict__ out,
                                                           unsigned long long* __restrict__ cyc, const int iters) {
    const int lane = threadIdx.x & 63;
    // per K step of 32: two k-halves; weights 64 rows = 2 fragments of 32, activations 128 rows = 4 fragments of 32, hi + lo planes
    f16x8 bh[2][2], bl[2][2], ah[2][4], al[2][4];
    int q = 0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
#pragma unroll
        for (int i = 0; i < 2; ++i) { bh[k][i] = src[lane + 64 * q++]; bl[k][i] = src[lane + 64 * q++]; }
#pragma unroll
        for (int j = 0; j < 4; ++j) { ah[k][j] = src[lane + 64 * q++]; al[k][j] = src[lane + 64 * q++]; }
    }
    f32x16 acc[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) acc[i][j][v] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[k][i], ah[k][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[k][i], al[k][j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[k][i], ah[k][j], acc[i][j], 0, 0, 0);
            }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int v = 0; v < 16; ++v) s += acc[i][j][v];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * WAVES + (threadIdx.x >> 6)] = t1 - t0;
}

// s_memtime / readcyclecounter may tick at a fixed reference clock instead of the shader clock: calibrate against a chain of
// dependent v_fma_f32 (4 cycles each per the guide) -- reported, not relied on
__global__ void fma_chain_kernel(float* out, unsigned long long* cyc, const int n) {
    float x = (float)threadIdx.x * 1e-9f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) x = __builtin_fmaf(x, 1.0000001f, 1e-9f);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) cyc[0] = t1 - t0;
}

template <class K>
static void run(const char* name, K kernel, int waves, double mfma_per_iter, double flop_per_mfma, const f16x8* src, float* out,
                unsigned long long* cyc, int iters, int ncu) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kernel, dim3(ncu), dim3(waves * 64), 0, 0, src, out, cyc, iters);
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    std::vector<unsigned long long> h(ncu * waves);
    double cyc_mean = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(kernel, dim3(ncu), dim3(waves * 64), 0, 0, src, out, cyc, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) {
            best = ms;
            CHECK(hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost));
            cyc_mean = 0;
            for (auto v : h) cyc_mean += (double)v;
            cyc_mean /= h.size();
        }
    }
    const double n_mfma = mfma_per_iter * iters;                       // per wave
    const double flops = n_mfma * flop_per_mfma * waves * ncu;
    const double tf = flops / (best * 1e-3) / 1e12;
    printf("%-44s %8.1f us  %7.1f TFLOP/s = %5.1f %% of 2.5 PF   counter ticks / MFMA %6.2f   ticks / wall %6.3f GHz\n", name,
           best * 1e3, tf, 100.0 * tf / 2500.0, cyc_mean / n_mfma, cyc_mean / (best * 1e-3) / 1e9);
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    printf("device %s, %d CUs, clockRate %d kHz\n", prop.name, ncu, prop.clockRate);
    const int NV = 64 * 64;
    std::vector<f16x8> h(NV);
    f16x8 *d_rand, *d_zero;
    float* out;
    unsigned long long* cyc;
    CHECK(hipMalloc(&d_rand, NV * sizeof(f16x8)));
    CHECK(hipMalloc(&d_zero, NV * sizeof(f16x8)));
    CHECK(hipMalloc(&out, ncu * 512 * sizeof(float)));
    CHECK(hipMalloc(&cyc, ncu * 8 * 8));
    srand(1);
    for (auto& v : h)
        for (int k = 0; k < 8; ++k) v[k] = (_Float16)(((rand() & 0xffff) / 32768.0f - 1.0f) * 0.05f);
    CHECK(hipMemcpy(d_rand, h.data(), NV * sizeof(f16x8), hipMemcpyHostToDevice));
    CHECK(hipMemset(d_zero, 0, NV * sizeof(f16x8)));
    {
        fma_chain_kernel<<<1, 64>>>(out, cyc, 4096);
        CHECK(hipDeviceSynchronize());
        unsigned long long c;
        CHECK(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost));
        printf("counter calibration: %.3f ticks per dependent v_fma_f32 (idle chip, one wave)\n", (double)c / (4096.0 * 64));
    }
    const int iters = 4000;          // ~ the K steps of 7 ASPP 3x3 tiles
    const double f16 = 2.0 * 16 * 16 * 32, f32 = 2.0 * 32 * 32 * 16;
    for (int pass = 0; pass < 2; ++pass) {
        const f16x8* src = pass == 0 ? d_rand : d_zero;
        printf("---- operands: %s\n", pass == 0 ? "random fp16" : "zeros");
        run("16x16x32_f16, 144x64 wave tile, 4 waves/CU", mfma16_kernel<4>, 4, 108, f16, src, out, cyc, iters, ncu);
        run("32x32x16_f16, 128x64 wave tile, 4 waves/CU", mfma32_kernel<4>, 4, 48, f32, src, out, cyc, iters, ncu);
        run("16x16x32_f16, 144x64 wave tile, 8 waves/CU", mfma16_kernel<8>, 8, 108, f16, src, out, cyc, iters / 2, ncu);
        run("32x32x16_f16, 128x64 wave tile, 8 waves/CU", mfma32_kernel<8>, 8, 48, f32, src, out, cyc, iters / 2, ncu);
    }
    return 0;
}
