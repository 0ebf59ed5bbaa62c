// Best-case HBM read rate: every workgroup streams its slice with 16-byte loads (UNROLL independent loads in flight per lane) and keeps a
// running sum; one float per workgroup is written.  hipcc --offload-arch=gfx950 -shared -fPIC -O3 -o tools/probe/read_probe.so tools/probe/read_probe.hip
#include <hip/hip_runtime.h>
template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, long n4, float* __restrict__ out) {
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = x[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) s += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    for (; i < n4; i += stride) { float4 v = x[i]; s += v.x + v.y + v.z + v.w; }
    if (s == 123456.789f) out[blockIdx.x] = s;      // never true for the probe data: the loads cannot be dropped
}
extern "C" void read_probe(const void* x, long n4, void* out, int blocks, int unroll, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (unroll == 1) read_kernel<1><<<blocks, 256, 0, s>>>((const float4*)x, n4, (float*)out);
    else if (unroll == 4) read_kernel<4><<<blocks, 256, 0, s>>>((const float4*)x, n4, (float*)out);
    else read_kernel<8><<<blocks, 256, 0, s>>>((const float4*)x, n4, (float*)out);
}
