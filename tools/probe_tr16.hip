// Debug probe (not part of the product): prints what ds_read_b64_tr_b16 returns per lane so that the
// operand addressing of the bf16 weight-gradient kernel can be checked on real gfx950 hardware.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_tr16.hip -o gpurun_out/probe_tr16 && gpurun_out/probe_tr16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short v4i16 __attribute__((ext_vector_type(4)));

__global__ void probe(unsigned short* out, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[64 * 64];
    for (int i = threadIdx.x; i < 64 * 64; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    const int lane = threadIdx.x;
    int idx;
    if (mode == 0) {
        idx = lane * 4;                          // linear: lane L reads elements 4L..4L+3
    } else {
        // H1: 16-lane group g = lane>>4 works on rows 8g..8g+3 (pitch 64), lane L -> row L/4, cols 4*(L%4)
        const int L = lane & 15, g = lane >> 4;
        idx = (g * 8 + (L >> 2)) * 64 + (L & 3) * 4;
    }
    v4i16 r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4i16 __attribute__((address_space(3)))*)(lds + idx));
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = (unsigned short)r[j];
}

int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    std::vector<unsigned short> h(256);
    for (int mode = 0; mode < 2; ++mode) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, mode);
        hipMemcpy(h.data(), d, 512, hipMemcpyDeviceToHost);
        printf("mode %d (value = LDS element index; row = v/64, col = v%%64 in mode 1)\n", mode);
        for (int l = 0; l < 64; ++l) {
            printf("lane %2d:", l);
            for (int j = 0; j < 4; ++j) {
                if (mode == 0) printf(" %4d", h[l * 4 + j]);
                else printf(" (r%d,c%d)", h[l * 4 + j] / 64, h[l * 4 + j] % 64);
            }
            printf("\n");
        }
    }
    return 0;
}
