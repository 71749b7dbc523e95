// Stand-in for an RCCL all-reduce kernel's LOCAL footprint (tools/probe_comm_coresidency.py): `nblocks` workgroups of 512 threads,
// `lds_bytes` of dynamic LDS each, streaming a buffer in place (read + write, 16 bytes per lane, grid-stride) `passes` times.
// Built by tools/probe_comm_coresidency.py's build step: hipcc --offload-arch=gfx950 -shared -fPIC -o tools/build/libcomm_standin.so
#include <hip/hip_runtime.h>
#include <stdint.h>
extern "C" __global__ __launch_bounds__(512) void standin_kernel(float4* __restrict__ buf, size_t n16, int passes, float add) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    if (threadIdx.x == 0) lds[0] = 1;                       // (the LDS is only there to be allocated)
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (int p = 0; p < passes; ++p)
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
            float4 v = buf[i];
            v.x += add; v.y += add; v.z += add; v.w += add;      // (add = 0 at run time: the buffer keeps its values, the traffic is real)
            buf[i] = v;
        }
}
extern "C" int standin_launch(void* buf, size_t bytes, int nblocks, int lds_bytes, int passes, void* stream) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(standin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(standin_kernel, dim3(nblocks), dim3(512), lds_bytes, static_cast<hipStream_t>(stream),
                       static_cast<float4*>(buf), bytes / 16, passes, 0.0f);
    return (int)hipGetLastError();
}
