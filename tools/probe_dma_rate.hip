// Probe (GPU box): per-CU operand-load rate of the conv kernels' LDS-DMA pattern, L2-resident source.
// The conv K step reads, per tile row (a pixel / a filter), BK bf16 = 64 contiguous bytes of a row that is 512+ bytes long:
// HALF a 128-byte cache line.  Question: is the rate per CU the same with 128 contiguous bytes per row (BK = 64)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_dma_rate.hip -o /tmp/probe_dma_rate && /tmp/probe_dma_rate
// Patterns: seg = 64 / 128 / 256 bytes per row and K step; workgroups per CU 1..3; 128-row tiles; every workgroup walks its
// own 128 rows x (row length) region over and over (L2-resident after the first pass), like a conv tile walks K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;
template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// SEG bytes per row per step; a stage holds ROWS x SEG bytes; NST stages; 256 threads = 4 waves
template <int SEG, int ROWS, int NST>
__device__ __forceinline__ void dma_body(const char* src, int row_bytes, int rows_total, int steps, float* sink) {
    constexpr int STAGE = ROWS * SEG;                 // bytes
    constexpr int LPR = SEG / 16;                     // lanes per row
    constexpr int RPI = 64 / LPR;                     // rows per instruction (1 KB)
    constexpr int NI = ROWS / RPI / 4;                // instructions per wave per stage
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rows_total * row_bytes, 0x00020000);
    const int row0 = (blockIdx.x * ROWS) % (rows_total - ROWS + 1);
    uint32_t base[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = (wave * NI + j) * RPI + lane / LPR;
        base[j] = (uint32_t)((row0 + r) * row_bytes + (lane % LPR) * 16);
    }
    const int segs_per_row = row_bytes / SEG;
#define ISSUE(s_, stage_)                                                                                                  \
    do {                                                                                                               \
        const uint32_t koff = (uint32_t)(((s_) % segs_per_row) * SEG);                                                 \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                       \
            rs, (lds_ptr_t)(smem + (stage_) * STAGE + (wave * NI + j) * 1024), 16, base[j] + koff, 0, 0, 0);             \
    } while (0)
    constexpr int LA = NST - 1;
#pragma unroll
    for (int t = 0; t < LA; ++t) ISSUE(t, t);
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (s + LA - 1 < steps) wait_vmcnt<(LA - 1) * NI>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + LA < steps) ISSUE(s + LA, (s + LA) % NST);
        acc += reinterpret_cast<const float*>(smem + (s % NST) * STAGE)[tid];      // touch the stage
    }
    if (acc == 123.456f) sink[0] = acc;
}

// Mixed stage of 16 KB: 128 rows x 64-byte segments (the activation tile as it is) + 8 KB read as ONE contiguous run (a weight
// tile pre-arranged tile-major: every 1 KB instruction covers eight whole 128-byte lines).  CONTIG_A: both halves contiguous.
template <int NST, bool CONTIG_A>
__device__ __forceinline__ void dma_mixed_body(const char* src, int row_bytes, int rows_total, int steps, float* sink) {
    constexpr int STAGE = 16384, NI = 2;
    __shared__ __attribute__((aligned(1024))) char smem[NST * STAGE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rows_total * row_bytes, 0x00020000);
    const int row0 = (blockIdx.x * 256) % (rows_total - 256 + 1);
    const int segs_per_row = row_bytes / 64;
    uint32_t base_a[NI], base_b[NI];
#pragma unroll
    for (int j = 0; j < NI; ++j) {
        const int r = (wave * NI + j) * 16 + lane / 4;
        base_a[j] = CONTIG_A ? (uint32_t)(row0 * row_bytes + (wave * NI + j) * 1024 + lane * 16)
                             : (uint32_t)((row0 + r) * row_bytes + (lane % 4) * 16);
        base_b[j] = (uint32_t)((row0 + 128) * row_bytes + (wave * NI + j) * 1024 + lane * 16);      // 128 rows x row_bytes region, linear
    }
#define ISSUE_M(s_, stage_)                                                                                                \
    do {                                                                                                               \
        const uint32_t ka = CONTIG_A ? (uint32_t)(((s_) % segs_per_row) * 8192) : (uint32_t)(((s_) % segs_per_row) * 64); \
        const uint32_t kb = (uint32_t)(((s_) % segs_per_row) * 8192);                                                  \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                       \
            rs, (lds_ptr_t)(smem + (stage_) * STAGE + (wave * NI + j) * 1024), 16, base_a[j] + ka, 0, 0, 0);             \
        _Pragma("unroll") for (int j = 0; j < NI; ++j) __builtin_amdgcn_raw_ptr_buffer_load_lds(                       \
            rs, (lds_ptr_t)(smem + (stage_) * STAGE + 8192 + (wave * NI + j) * 1024), 16, base_b[j] + kb, 0, 0, 0);      \
    } while (0)
    constexpr int LA = NST - 1;
#pragma unroll
    for (int t = 0; t < LA; ++t) ISSUE_M(t, t);
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        if (s + LA - 1 < steps) wait_vmcnt<(LA - 1) * 2 * NI>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + LA < steps) ISSUE_M(s + LA, (s + LA) % NST);
        acc += reinterpret_cast<const float*>(smem + (s % NST) * STAGE)[tid];
    }
    if (acc == 123.456f) sink[0] = acc;
}
__global__ __launch_bounds__(256) void k_mixed(const char* src, int rb, int rows, int steps, float* sink) {
    dma_mixed_body<3, false>(src, rb, rows, steps, sink);
}
__global__ __launch_bounds__(256) void k_contig(const char* src, int rb, int rows, int steps, float* sink) {
    dma_mixed_body<3, true>(src, rb, rows, steps, sink);
}

// Asymmetric ring in the same 48 KB: the activation tile as UNITS of two K steps (128 rows x 128 bytes = 16 KB, whole lines,
// two units), the weight tile per K step (8 KB contiguous, two slots, issued ONE step ahead).  Per step 16 KB as before.
__global__ __launch_bounds__(256) void k_asym(const char* src, int row_bytes, int rows_total, int steps, float* sink) {
    __shared__ __attribute__((aligned(1024))) char smem[2 * 16384 + 2 * 8192];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, rows_total * row_bytes, 0x00020000);
    const int row0 = (blockIdx.x * 256) % (rows_total - 256 + 1);
    const int units_per_row = row_bytes / 128, segs_per_row = row_bytes / 64;
    uint32_t base_a[4], base_b[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) base_a[j] = (uint32_t)((row0 + (wave * 4 + j) * 8 + lane / 8) * row_bytes + (lane % 8) * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) base_b[j] = (uint32_t)((row0 + 128) * row_bytes + (wave * 2 + j) * 1024 + lane * 16);
    auto issue_a = [&](int u) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + (u & 1) * 16384 + (wave * 4 + j) * 1024), 16,
                                                     base_a[j] + (uint32_t)((u % units_per_row) * 128), 0, 0, 0);
    };
    auto issue_b = [&](int s) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + 32768 + (s & 1) * 8192 + (wave * 2 + j) * 1024), 16,
                                                     base_b[j] + (uint32_t)((s % segs_per_row) * 8192), 0, 0, 0);
    };
    issue_a(0); issue_b(0);
    float acc = 0.f;
    for (int s = 0; s < steps; ++s) {
        // outstanding, oldest first: [A unit s/2 + B(s) needed now] ... at odd s the A unit issued last step stays in flight
        if (s & 1) wait_vmcnt<4>(); else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (s + 1 < steps) issue_b(s + 1);
        if (!(s & 1) && s + 2 < steps) issue_a(s / 2 + 1);
        acc += reinterpret_cast<const float*>(smem + ((s >> 1) & 1) * 16384)[tid] + reinterpret_cast<const float*>(smem + 32768 + (s & 1) * 8192)[tid];
    }
    if (acc == 123.456f) sink[0] = acc;
}

#define KERNEL(SEG, ROWS, NST)                                                                                            \
    __global__ __launch_bounds__(256) void k_##SEG##_##ROWS##_##NST(const char* src, int rb, int rows, int steps, float* sink) { \
        dma_body<SEG, ROWS, NST>(src, rb, rows, steps, sink);                                                             \
    }
KERNEL(64, 256, 3) KERNEL(128, 128, 3) KERNEL(256, 64, 3) KERNEL(64, 512, 2) KERNEL(128, 256, 2) KERNEL(256, 128, 2)

typedef void (*kern_t)(const char*, int, int, int, float*);
double run(kern_t k, int seg, int rows_per_stage, const char* src, int row_bytes, int rows, int wg_per_cu, int steps, float* sink) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int grid = 256 * wg_per_cu;
    for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, row_bytes, rows, steps, sink);
    hipEventRecord(e0);
    const int reps = 5;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, src, row_bytes, rows, steps, sink);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)grid * steps * rows_per_stage * seg;
    return bytes / (ms / reps * 1e-3) / 256.0 / 1e9;           // GB/s per CU
}

int main(int argc, char** argv) {
    const int row_bytes = (argc > 1 ? atoi(argv[1]) : 512), rows = (argc > 2 ? atoi(argv[2]) : 16384);
    char* src; float* sink;
    hipMalloc(&src, (size_t)rows * row_bytes); hipMemset(src, 1, (size_t)rows * row_bytes); hipMalloc(&sink, 64);
    printf("bytes per step and workgroup fixed at 16 KB (256 rows x 64 B | 128 rows x 128 B | 64 rows x 256 B) and 32 KB; GB/s per CU (x256 = chip)\n");
    for (int w = 1; w <= 3; ++w) {
        const int total = 16384;        // KB per workgroup
        printf("%d WG/CU | 64 B/row: 16KB x3 stages %.1f | 128 B/row: 16KB x3 %.1f | 256 B/row: 16KB x3 %.1f | 64 B/row 32KB x2 %.1f | 128 B/row 32KB x2 %.1f | 256 B/row 32KB x2 %.1f\n", w,
               run(k_64_256_3, 64, 256, src, row_bytes, rows, w, total / 16, sink), run(k_128_128_3, 128, 128, src, row_bytes, rows, w, total / 16, sink),
               run(k_256_64_3, 256, 64, src, row_bytes, rows, w, total / 16, sink), run(k_64_512_2, 64, 512, src, row_bytes, rows, w, total / 32, sink),
               run(k_128_256_2, 128, 256, src, row_bytes, rows, w, total / 32, sink), run(k_256_128_2, 256, 128, src, row_bytes, rows, w, total / 32, sink));
    }
    printf("16 KB x3 stages, half of it (the weight tile) as one contiguous run | both halves contiguous\n");
    for (int w = 1; w <= 3; ++w)
        printf("%d WG/CU | 8 KB of 64 B rows + 8 KB contiguous %.1f | 16 KB contiguous %.1f\n", w,
               run(k_mixed, 64, 256, src, row_bytes, rows, w, 1024, sink), run(k_contig, 64, 256, src, row_bytes, rows, w, 1024, sink));
    printf("asymmetric ring: activation units of two K steps (16 KB of 128-byte rows, 2 units) + weight tile 8 KB contiguous per step (2 slots, one step ahead)\n");
    for (int w = 1; w <= 3; ++w) printf("%d WG/CU | %.1f\n", w, run(k_asym, 64, 256, src, row_bytes, rows, w, 1024, sink));
    return 0;
}
