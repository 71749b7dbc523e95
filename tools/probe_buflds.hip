// Debug probe: raw_buffer_load_lds (LDS-DMA through a buffer descriptor) -- does an out-of-range voffset
// write zeros into LDS?  Lane l copies 16 B; odd lanes are sent out of range.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(const float* src, float* out, int nbytes) {
    __shared__ __attribute__((aligned(16))) float lds[64 * 4];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = -7.f;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, nbytes, 0x00020000);
    const int lane = threadIdx.x;
    unsigned voff = (lane & 1) ? 0x7ffffff0u : lane * 16;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 256; i += 64) out[i] = lds[i];
}
int main() {
    float *s, *o; hipMalloc(&s, 1024); hipMalloc(&o, 1024);
    std::vector<float> h(256); for (int i = 0; i < 256; ++i) h[i] = i + 1;
    hipMemcpy(s, h.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, s, o, 1024);
    hipMemcpy(h.data(), o, 1024, hipMemcpyDeviceToHost);
    for (int l = 0; l < 8; ++l) printf("lane %d: %g %g %g %g\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
