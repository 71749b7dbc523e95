// Probe (GPU box): where do the waves of a 2-workgroups-per-CU launch land?  512 workgroups x 256 threads, 74 KB of LDS and <= 256
// registers each (the shape of the half-tile configuration of conv_ws_kernel).  Per wave: blockIdx, wave index, XCC, SE, CU, SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_wg_placement.hip -o tools/build/probe_wg_placement && tools/build/probe_wg_placement
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2))) void probe(uint32_t* out, int spin) {
    __shared__ char smem[74 * 1024];
    smem[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    const uint64_t t0 = wall_clock64();
    while (wall_clock64() - t0 < (uint64_t)spin) __builtin_amdgcn_s_sleep(8);      // keep every workgroup resident for a while
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        out[2 * w] = hw;
        out[2 * w + 1] = xcc + (uint32_t)smem[threadIdx.x & 7] * 0u;
    }
}

int main() {
    const int NWG = 512;
    uint32_t* d;
    CHECK(hipMalloc(&d, NWG * 4 * 8));
    probe<<<NWG, 256>>>(d, 5000);       // 50 us
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h(NWG * 8);
    CHECK(hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost));
    std::map<uint32_t, std::vector<int>> by_cu;      // (xcc, se, sh, cu) -> workgroups
    int same_simd_order = 0;
    for (int b = 0; b < NWG; ++b) {
        const uint32_t hw0 = h[(b * 4) * 2], xcc = h[(b * 4) * 2 + 1] & 0xf;
        const uint32_t cu = (hw0 >> 8) & 0xf, sh = (hw0 >> 12) & 1, se = (hw0 >> 13) & 7;
        by_cu[(xcc << 12) | (se << 8) | (sh << 4) | cu].push_back(b);
        bool inorder = true;
        for (int w = 0; w < 4; ++w) inorder = inorder && (((h[(b * 4 + w) * 2] >> 4) & 3) == (uint32_t)w);
        same_simd_order += inorder;
    }
    printf("workgroups whose wave w sits on SIMD w: %d of %d\n", same_simd_order, NWG);
    printf("distinct CUs used: %zu\n", by_cu.size());
    int pairs_256 = 0, pairs_adj = 0, singles = 0, more = 0;
    for (auto& kv : by_cu) {
        auto& v = kv.second;
        if (v.size() == 1) ++singles;
        else if (v.size() == 2) {
            const int d2 = abs(v[0] - v[1]);
            if (d2 == 256) ++pairs_256;
            if (d2 == 1 || d2 == 8) ++pairs_adj;
        } else ++more;
    }
    printf("CUs with 1 workgroup: %d, with 2: %zu (blockIdx apart by 256: %d, by 1 or 8: %d), with more: %d\n", singles,
           by_cu.size() - singles - more, pairs_256, pairs_adj, more);
    for (int b : {0, 1, 2, 8, 255, 256, 257, 264, 511}) {
        printf("block %3d:", b);
        for (int w = 0; w < 4; ++w) {
            const uint32_t hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            printf("  w%d xcc%u se%u sh%u cu%2u simd%u slot%u", w, xcc, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 0xf, (hw >> 4) & 3, hw & 0xf);
        }
        printf("\n");
    }
    int shown = 0;
    for (auto& kv : by_cu) {
        if (shown++ >= 6) break;
        printf("CU key %05x: blocks", kv.first);
        for (int b : kv.second) {
            printf(" %d[simd", b);
            for (int w = 0; w < 4; ++w) printf("%u", (h[(b * 4 + w) * 2] >> 4) & 3);
            printf("]");
        }
        printf("\n");
    }
    return 0;
}
