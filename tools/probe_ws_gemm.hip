// Probe (GPU box): wave-specialised implicit-GEMM tile for the bf16 forward / data-gradient convolutions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/probe_ws_gemm.hip -o tools/build/probe_ws_gemm && tools/build/probe_ws_gemm
// Structure under test: dedicated LOADER waves issue every buffer_load ... lds of a K step into an LDS ring
// and publish "stage landed" counters in LDS; CONSUMER waves never touch the vector-memory path: they poll the counter, read
// fragments (next step's fragments are fetched under this step's MFMAs, register by register as they become free) and issue
// MFMAs back to back.  No s_barrier in the K loop; the ring decouples the two sides.  Persistent workgroups walk a tile list,
// the loaders run ahead into the next tile while the consumers store the finished one.
// C[m][n] = sum_{t, c} A[m + shift(t)][c] * B[n][t][c]   (taps t: 1, or 9 with shift = (r - 1) * W + (s - 1); rows outside
// [0, M) read as zeros) -- the access pattern of a 1x1 / 3x3 convolution over an NHWC tensor, without image borders.
// B is stored tile-major [N / 64][K / 32][64][32] like the product's `w_tiled` weight copies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>

typedef unsigned short bf16_t;
typedef __bf16 mfma_bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    typedef __bf16 bf16x2_hw __attribute__((ext_vector_type(2)));
    typedef float f32x2_hw __attribute__((ext_vector_type(2)));
    const f32x2_hw v = {lo, hi};
    const bf16x2_hw b = __builtin_convertvector(v, bf16x2_hw);
    return __builtin_bit_cast(uint32_t, b);
}

struct Args {
    const bf16_t* A;
    const bf16_t* Bt;
    bf16_t* Cout;
    int M, N, C, taps, W;
    uint32_t a_bytes, b_bytes;
    int ntiles_m, ntiles_n;
};

// weight-tile row order of the product kernels (conv_igemm.hip frag_chan / b_row / b_rho, NT = 4)
__device__ __forceinline__ int frag_chan(int i, int g) { return (i >> 1) * 32 + g * 8 + (i & 1) * 4; }
__device__ __forceinline__ int b_row(int i, int rho) { return frag_chan(i, rho >> 2) + (rho & 3); }
__device__ __forceinline__ int b_rho(int row) { return ((row >> 3) & 3) * 4 + (row & 3); }
__device__ __forceinline__ int swz(int row, int chunk) { return chunk ^ ((0x78 >> (((row >> 2) & 3) * 2)) & 3); }

// LDS-DMA piece (16 B per lane, 1 KB per wave) issued from inline asm: hipcc does not see it, so it neither counts it nor
// waits vmcnt(0) before the loader's own LDS accesses (its flag polls); completion is tracked by hand (wait_vmcnt).
// m0 = LDS byte address of the piece (wave-uniform); voff per lane; out-of-range offsets write zeros.
__device__ __forceinline__ void dma16(const u32x4_t rsrc, const uint32_t lds_addr, const uint32_t voff, const uint32_t soff) {
    uint32_t keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff)
                 : "memory");
}
__device__ __forceinline__ u32x4_t make_rsrc(const void* base, uint32_t bytes) {
    const uint64_t b = reinterpret_cast<uint64_t>(base);
    u32x4_t r;
    r[0] = __builtin_amdgcn_readfirstlane((uint32_t)b);
    r[1] = __builtin_amdgcn_readfirstlane((uint32_t)(b >> 32) & 0xffffu);
    r[2] = __builtin_amdgcn_readfirstlane(bytes);
    r[3] = 0x00020000u;
    return r;
}

__device__ __forceinline__ uint32_t lds_ld(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_st(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// one loader wave: pieces p = q * NLD + LW of every stage (compile-time ownership: no branches in the issue loop)
template <int MT, int MW, int NW, int NST, int NLD, int D, int LW, int ABL>
__device__ __forceinline__ void ws_loader(const Args& a, char* smem, uint32_t* ready, uint32_t* consumed, const int lane,
                                          const int ntiles, const int KT) {
    constexpr int NCW = MW * NW;
    constexpr int BM = 16 * MT * MW, BN = 64 * NW;
    constexpr int PA = BM / 16, PB = BN / 16, NP = PA + PB;
    constexpr int SB = (BM + BN) * 64;
    constexpr int MYP = (NP - LW + NLD - 1) / NLD;            // pieces of this wave per stage
    constexpr uint32_t OOB = 0x80000000u;
    const u32x4_t rs_a = make_rsrc(a.A, a.a_bytes), rs_b = make_rsrc(a.Bt, a.b_bytes);
    const uint32_t lds0 = (uint32_t)(uintptr_t)smem;
    const int prow = lane >> 2;
    const int lchunk = (lane & 3) ^ ((0x78 >> (((lane >> 4) & 3) * 2)) & 3);
    uint32_t g = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int mt = tile % a.ntiles_m, nt = tile / a.ntiles_m;
        const int m0 = mt * BM, n0 = nt * BN;
        int arow[MYP];
        uint32_t off[MYP];           // A pieces: byte offset of (row, chunk) at shift 0, c0 0; B pieces: byte offset at K step 0
#pragma unroll
        for (int q = 0; q < MYP; ++q) {
            const int p = q * NLD + LW;
            if (p < PA) {
                arow[q] = m0 + p * 16 + prow;
                off[q] = (uint32_t)((arow[q] * a.C + lchunk * 8) * 2);
            } else {
                const int row = (p - PA) * 16 + prow, n = n0 + row;
                const int bchunk = swz(b_rho(row), lane & 3);
                arow[q] = 0;
                off[q] = n < a.N ? (uint32_t)((((int64_t)(n >> 6) * KT) * 2048 + (n & 63) * 32 + bchunk * 8) * 2) : OOB;
            }
        }
        int tap = 0, c0 = 0;
        for (int kt = 0; kt < KT; ++kt) {
            if (g >= (uint32_t)NST) {
                const uint32_t need = g - NST + 1;
                for (;;) {
                    uint32_t mn = lds_ld(consumed);
#pragma unroll
                    for (int w = 1; w < NCW; ++w) mn = min(mn, lds_ld(consumed + w));
                    if (mn >= need) break;
                    __builtin_amdgcn_s_sleep(1);
                }
                asm volatile("" ::: "memory");
            }
            const uint32_t sbase = lds0 + (g % NST) * SB;
            const int shift = a.taps == 1 ? 0 : ((tap / 3) - 1) * a.W + (tap % 3) - 1;
            const uint32_t soff_a = (uint32_t)((shift * a.C + c0) * 2);
            if (ABL != 2) {
#pragma unroll
                for (int q = 0; q < MYP; ++q) {
                    const int p = q * NLD + LW;
                    if (p < PA) {
                        const uint32_t voff = ((unsigned)(arow[q] + shift) < (unsigned)a.M) ? off[q] + soff_a : OOB;
                        dma16(rs_a, sbase + p * 1024, voff, 0u);
                    } else {
                        dma16(rs_b, sbase + p * 1024, off[q], (uint32_t)kt * 4096u);
                    }
                }
            }
            c0 += 32;
            if (c0 >= a.C) { c0 = 0; ++tap; }
            ++g;
            if (g > (uint32_t)D) {
                wait_vmcnt<D * MYP>();
                lds_st(ready + LW, g - D);
            }
        }
    }
    wait_vmcnt<0>();
    lds_st(ready + LW, g);
}

// MT: 16-row fragments per wave tile (wave tile = 16 MT x 64); MW x NW consumer waves; NST ring stages of one 32-deep K step;
// NLD loader waves, each keeps D stages in flight behind the one it publishes
// ABL (ablations): 1 = consumers issue no MFMA, 2 = loaders issue no DMA (consumers run on whatever LDS holds)
template <int MT, int MW, int NW, int NST, int NLD, int D, int ABL = 0>
__global__ __launch_bounds__((MW * NW + NLD) * 64) void ws_gemm_kernel(const Args a) {
    constexpr int NCW = MW * NW;
    constexpr int BM = 16 * MT * MW, BN = 64 * NW;
    constexpr int PA = BM / 16, PB = BN / 16, NP = PA + PB;      // 1 KB DMA pieces (16 rows x 64 B) per stage
    constexpr int SB = (BM + BN) * 64;                            // bytes per stage
    constexpr uint32_t OOB = 0x80000000u;
    static_assert(NST >= D + 2, "ring too shallow");
    __shared__ __attribute__((aligned(1024))) char smem[NST * SB + 64];
    uint32_t* const ready = reinterpret_cast<uint32_t*>(smem + NST * SB);          // [NLD] stages landed
    uint32_t* const consumed = ready + 4;                                          // [NCW] stages whose reads were issued

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    if (tid < 16) reinterpret_cast<uint32_t*>(smem + NST * SB)[tid] = 0;
    __syncthreads();

    const int ntiles = a.ntiles_m * a.ntiles_n;
    const int cpt = a.C / 32, KT = a.taps * cpt;

    if (wave >= NCW) {
        const int lw = wave - NCW;
        if (NLD > 0 && lw == 0) ws_loader<MT, MW, NW, NST, NLD, D, 0, ABL>(a, smem, ready, consumed, lane, ntiles, KT);
        if (NLD > 1 && lw == 1) ws_loader<MT, MW, NW, NST, NLD, D, (NLD > 1 ? 1 : 0), ABL>(a, smem, ready, consumed, lane, ntiles, KT);
        if (NLD > 2 && lw == 2) ws_loader<MT, MW, NW, NST, NLD, D, (NLD > 2 ? 2 : 0), ABL>(a, smem, ready, consumed, lane, ntiles, KT);
        if (NLD > 3 && lw == 3) ws_loader<MT, MW, NW, NST, NLD, D, (NLD > 3 ? 3 : 0), ABL>(a, smem, ready, consumed, lane, ntiles, KT);
        return;
    }

    // ---------------------------------------------------------------------- consumer
    const int wm = wave / NW, wn = wave % NW;
    const int lr = lane & 15, lq = lane >> 4;
    uint32_t g = 0;
    uint32_t rflag = 0;
    auto read_ready = [&]() -> uint32_t {
        uint32_t v = lds_ld(ready);
#pragma unroll
        for (int w = 1; w < NLD; ++w) v = min(v, lds_ld(ready + w));
        return v;
    };
    auto wait_ready = [&](uint32_t need) {
        while (rflag < need) {
            __builtin_amdgcn_s_sleep(1);
            rflag = read_ready();
        }
        asm volatile("" ::: "memory");
    };
    // fragment addresses inside a stage (bytes)
    int a_off[MT], b_off[4];
#pragma unroll
    for (int j = 0; j < MT; ++j) {
        const int row = wm * (16 * MT) + j * 16 + lr;
        a_off[j] = row * 64 + swz(j * 16 + lr, lq) * 16;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) b_off[i] = BM * 64 + (wn * 64 + b_row(i, lr)) * 64 + swz(lr, lq) * 16;

    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const int mt = tile % a.ntiles_m, nt = tile / a.ntiles_m;
        const int m0 = mt * BM + wm * (16 * MT), n0 = nt * BN + wn * 64;
        f32x4 acc[4][MT];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

        mfma_bf16x8 bfA[4], bfB[4], af[MT], alA, alB;       // af[MT - 1] is unused: the last fragment alternates alA / alB
        // first K step of the tile: fragments straight from the ring (exposed once per tile)
        wait_ready(g + 1);
        {
            const char* sb = smem + (g % NST) * SB;
#pragma unroll
            for (int i = 0; i < 4; ++i) bfA[i] = *reinterpret_cast<const mfma_bf16x8*>(sb + b_off[i]);
#pragma unroll
            for (int j = 0; j < MT - 1; ++j) af[j] = *reinterpret_cast<const mfma_bf16x8*>(sb + a_off[j]);
            alA = *reinterpret_cast<const mfma_bf16x8*>(sb + a_off[MT - 1]);
        }
        // one K step: MFMAs of step g from (bc, af, alc); fragments of step g + 1 into (bn, af, aln) as registers become free
        auto step = [&](mfma_bf16x8 (&bc)[4], mfma_bf16x8 (&bn)[4], mfma_bf16x8& alc, mfma_bf16x8& aln, const bool has_next) {
            asm volatile("" ::: "memory");
            lds_st(consumed + wave, g + 1);          // every read of stage g has been issued (DS executes a wave's ops in order)
            const char* sn = smem + ((g + 1) % NST) * SB;
            if (has_next) {
                wait_ready(g + 2);
#pragma unroll
                for (int i = 0; i < 4; ++i) bn[i] = *reinterpret_cast<const mfma_bf16x8*>(sn + b_off[i]);
                aln = *reinterpret_cast<const mfma_bf16x8*>(sn + a_off[MT - 1]);
            }
#pragma unroll
            for (int j = 0; j < MT - 1; ++j) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (ABL == 1) asm volatile("" :: "v"(bc[i]), "v"(af[j]));
                    else acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[i], af[j], acc[i][j], 0, 0, 0);
                }
                if (has_next) af[j] = *reinterpret_cast<const mfma_bf16x8*>(sn + a_off[j]);
                if (j == (MT - 1) / 2) rflag = read_ready();      // the next step's poll, answered under the MFMAs
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (ABL == 1) asm volatile("" :: "v"(bc[i]), "v"(alc));
                else acc[i][MT - 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bc[i], alc, acc[i][MT - 1], 0, 0, 0);
            }
            ++g;
        };
        for (int kt = 0; kt < KT; kt += 2) {
            step(bfA, bfB, alA, alB, true);                 // KT is even
            step(bfB, bfA, alB, alA, kt + 2 < KT);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < MT; ++j) asm volatile("" : "+v"(acc[i][j]));
        // epilogue: lane holds, per row fragment j, two runs of 8 consecutive channels of row j * 16 + lr
#pragma unroll
        for (int j = 0; j < MT; ++j) {
            const int m = m0 + j * 16 + lr;
            if (m >= a.M) continue;
#pragma unroll
            for (int gq = 0; gq < 2; ++gq) {
                const int n = n0 + frag_chan(gq * 2, lq);
                if (n >= a.N) continue;
                const u32x4_t v = {pack_bf16x2(acc[gq * 2][j][0], acc[gq * 2][j][1]), pack_bf16x2(acc[gq * 2][j][2], acc[gq * 2][j][3]),
                                   pack_bf16x2(acc[gq * 2 + 1][j][0], acc[gq * 2 + 1][j][1]),
                                   pack_bf16x2(acc[gq * 2 + 1][j][2], acc[gq * 2 + 1][j][3])};
                *reinterpret_cast<u32x4_t*>(a.Cout + (int64_t)m * a.N + n) = v;
            }
        }
    }
}

// naive reference (fp32 accumulate, logical B[N][K])
__global__ void ref_kernel(const bf16_t* A, const bf16_t* B, float* Cref, int M, int N, int C, int taps, int W, int row0, int rows) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int m = row0 + blockIdx.y;
    if (n >= N || blockIdx.y >= rows) return;
    float acc = 0.f;
    for (int t = 0; t < taps; ++t) {
        const int shift = taps == 1 ? 0 : ((t / 3) - 1) * W + (t % 3) - 1;
        const int r = m + shift;
        if (r < 0 || r >= M) continue;
        for (int c = 0; c < C; ++c) {
            const float av = __uint_as_float(((uint32_t)A[(int64_t)r * C + c]) << 16);
            const float bv = __uint_as_float(((uint32_t)B[((int64_t)n * taps + t) * C + c]) << 16);
            acc += av * bv;
        }
    }
    Cref[(int64_t)blockIdx.y * N + n] = acc;
}

static bf16_t f2bf(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}
static float bf2f(bf16_t b) {
    uint32_t u = ((uint32_t)b) << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

struct Shape { int M, N, C, taps, W; const char* name; };

template <int MT, int MW, int NW, int NST, int NLD, int D, int ABL = 0>
static void run_cfg(const char* cfg, const Shape& s, const bf16_t* dA, const bf16_t* dBt, const bf16_t* dB, bf16_t* dC, float* dRef,
                    int wg_per_cu, bool check) {
    constexpr int BM = 16 * MT * MW, BN = 64 * NW;
    if (s.N % BN != 0) { printf("  %-34s skipped (N %% %d)\n", cfg, BN); return; }
    Args a;
    a.A = dA; a.Bt = dBt; a.Cout = dC; a.M = s.M; a.N = s.N; a.C = s.C; a.taps = s.taps; a.W = s.W;
    a.a_bytes = (uint32_t)((int64_t)s.M * s.C * 2);
    a.b_bytes = (uint32_t)((int64_t)s.N * s.taps * s.C * 2);
    a.ntiles_m = (s.M + BM - 1) / BM;
    a.ntiles_n = s.N / BN;
    const int ntiles = a.ntiles_m * a.ntiles_n;
    const int grid = std::min(ntiles, 256 * wg_per_cu);
    const int threads = (MW * NW + NLD) * 64;
#define KERN (ws_gemm_kernel<MT, MW, NW, NST, NLD, D, ABL>)
    CHECK(hipMemset(dC, 0, (size_t)s.M * s.N * 2));
    hipLaunchKernelGGL(KERN, dim3(grid), dim3(threads), 0, 0, a);
    CHECK(hipGetLastError());
    CHECK(hipDeviceSynchronize());
    double maxerr = 0.0, maxref = 0.0;
    if (check) {
        // rows checked: the first tile, a middle one and the last rows
        const int rows = 160;
        const int starts[3] = {0, (s.M / 2 / 16) * 16 - 16, s.M - rows};
        std::vector<float> ref((size_t)rows * s.N);
        std::vector<bf16_t> out((size_t)rows * s.N);
        for (int q = 0; q < 3; ++q) {
            hipLaunchKernelGGL(ref_kernel, dim3((s.N + 63) / 64, rows), dim3(64), 0, 0, dA, dB, dRef, s.M, s.N, s.C, s.taps, s.W, starts[q], rows);
            CHECK(hipMemcpy(ref.data(), dRef, ref.size() * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(out.data(), dC + (size_t)starts[q] * s.N, out.size() * 2, hipMemcpyDeviceToHost));
            for (size_t i = 0; i < ref.size(); ++i) {
                maxerr = std::max(maxerr, (double)fabsf(bf2f(out[i]) - ref[i]));
                maxref = std::max(maxref, (double)fabsf(ref[i]));
            }
        }
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 20;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(threads), 0, 0, a);
    CHECK(hipEventRecord(e0));
    for (int i = 0; i < iters; ++i) hipLaunchKernelGGL(KERN, dim3(grid), dim3(threads), 0, 0, a);
    CHECK(hipEventRecord(e1));
    CHECK(hipEventSynchronize(e1));
    float ms;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double t = ms / iters * 1e-3;
    const double fl = 2.0 * s.M * s.N * (double)s.taps * s.C;
    printf("  %-34s tiles %5d grid %4d  %8.1f us  %7.1f TFLOP/s", cfg, ntiles, grid, t * 1e6, fl / t / 1e12);
    if (check) printf("   max|err| %.4f of %.1f (%s)", maxerr, maxref, maxerr <= 0.02 * maxref + 1e-3 ? "ok" : "WRONG");
    printf("\n");
    fflush(stdout);
#undef KERN
}

int main(int argc, char** argv) {
    const Shape shapes[] = {
        {36864, 256, 256, 9, 48, "layer3 3x3 256->256 @48^2 x16"},
        {36864, 256, 1024, 1, 48, "layer3 1x1 1024->256"},
        {36864, 1024, 256, 1, 48, "layer3 1x1 256->1024"},
        {36864, 256, 2048, 9, 48, "aspp 3x3 2048->256"},
        {36864, 512, 512, 9, 48, "layer4 3x3 512->512"},
        {589824, 256, 320, 9, 192, "decoder 3x3 320->256 @192^2 x16"},
        {147456, 512, 128, 1, 96, "layer2 1x1 128->512 @96^2"},
        {147456, 128, 512, 1, 96, "layer2 1x1 512->128"},
        {589824, 256, 64, 1, 192, "layer1 1x1 64->256 @192^2"},
    };
    const int only = argc > 1 ? atoi(argv[1]) : -1;
    for (int si = 0; si < (int)(sizeof(shapes) / sizeof(shapes[0])); ++si) {
        if (only >= 0 && si != only) continue;
        const Shape& s = shapes[si];
        const int64_t K = (int64_t)s.taps * s.C;
        std::vector<bf16_t> hA((size_t)s.M * s.C), hB((size_t)s.N * K), hBt((size_t)s.N * K);
        uint32_t seed = 12345u + si;
        auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
        for (auto& v : hA) v = f2bf(rnd());
        for (auto& v : hB) v = f2bf(rnd() * 0.1f);
        const int KT = (int)(K / 32);
        for (int n = 0; n < s.N; ++n)
            for (int64_t k = 0; k < K; ++k)
                hBt[(((size_t)(n >> 6) * KT + k / 32) * 64 + (n & 63)) * 32 + k % 32] = hB[(size_t)n * K + k];
        bf16_t *dA, *dB, *dBt, *dC;
        float* dRef;
        CHECK(hipMalloc(&dA, hA.size() * 2));
        CHECK(hipMalloc(&dB, hB.size() * 2));
        CHECK(hipMalloc(&dBt, hBt.size() * 2));
        CHECK(hipMalloc(&dC, (size_t)s.M * s.N * 2));
        CHECK(hipMalloc(&dRef, (size_t)160 * s.N * 4));
        CHECK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dB, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dBt, hBt.data(), hBt.size() * 2, hipMemcpyHostToDevice));
        printf("%s: M %d N %d K %lld (%.1f GFLOP)\n", s.name, s.M, s.N, (long long)K, 2.0 * s.M * s.N * K / 1e9);
        //       MT MW NW NST NLD D
        run_cfg<9, 1, 4, 6, 2, 2>("144x256 4c+2l ring6 D2", s, dA, dBt, dB, dC, dRef, 1, true);
        run_cfg<9, 1, 4, 6, 3, 2>("144x256 4c+3l ring6 D2", s, dA, dBt, dB, dC, dRef, 1, true);
        run_cfg<9, 1, 4, 6, 4, 2>("144x256 4c+4l ring6 D2", s, dA, dBt, dB, dC, dRef, 1, true);
        run_cfg<9, 1, 4, 6, 4, 3>("144x256 4c+4l ring6 D3", s, dA, dBt, dB, dC, dRef, 1, false);
        run_cfg<9, 1, 4, 6, 4, 1>("144x256 4c+4l ring6 D1", s, dA, dBt, dB, dC, dRef, 1, false);
        run_cfg<9, 1, 4, 4, 4, 2>("144x256 4c+4l ring4 D2", s, dA, dBt, dB, dC, dRef, 1, false);
        run_cfg<9, 1, 4, 6, 4, 2, 1>("144x256 4c+4l D2 ABL no-MFMA", s, dA, dBt, dB, dC, dRef, 1, false);
        run_cfg<9, 1, 4, 6, 4, 2, 2>("144x256 4c+4l D2 ABL no-DMA", s, dA, dBt, dB, dC, dRef, 1, false);
        run_cfg<8, 1, 4, 6, 4, 2>("128x256 4c+4l ring6 D2", s, dA, dBt, dB, dC, dRef, 1, true);
        run_cfg<8, 2, 2, 6, 4, 2>("256x128 4c+4l ring6 D2", s, dA, dBt, dB, dC, dRef, 1, true);
        run_cfg<4, 2, 4, 6, 4, 2>("128x256 8c(64x64)+4l ring6 D2", s, dA, dBt, dB, dC, dRef, 1, true);
        run_cfg<4, 2, 2, 4, 2, 2>("128x128 4c+2l ring4 D2 x2/CU", s, dA, dBt, dB, dC, dRef, 2, true);
        CHECK(hipFree(dA)); CHECK(hipFree(dB)); CHECK(hipFree(dBt)); CHECK(hipFree(dC)); CHECK(hipFree(dRef));
    }
    return 0;
}
