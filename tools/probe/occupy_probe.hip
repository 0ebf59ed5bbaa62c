// occupy_probe.hip — a stand-in for a collective's kernel: `n` workgroups of 256 threads that hold their CU slot (dyn_lds bytes of LDS, a register
// footprint of `REGS` per lane) for `ticks` ticks of the 100 MHz real-time counter and touch no memory.  tools/probe/occupy_probe.py runs the
// camera encoder's conv backward chain beside it.
#include <hip/hip_runtime.h>
#include <stdint.h>

template <int REGS>
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks, float* sink) {
    extern __shared__ char lds[];
    float r[REGS];
#pragma unroll
    for (int i = 0; i < REGS; ++i) r[i] = (float)(threadIdx.x + i);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {
#pragma unroll
        for (int i = 0; i < REGS; ++i) r[i] = r[i] * 1.0001f + 0.5f;
        __builtin_amdgcn_s_sleep(32);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < REGS; ++i) s += r[i];
    if (s == 12345.678f) { sink[0] = s; lds[0] = 1; }
}

// a paced streaming pass (the shape of an Adam pass: 4 arrays read, 3 written, 16 bytes per lane and array): `n` workgroups of 256 threads walk
// `total` float4 quads grid-stride and sleep `pace` x 64 clocks after every quad
__global__ __launch_bounds__(256) void stream_kernel(float4* a, const float4* b, float4* c, float4* d, long total, int pace) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        float4 x = a[i], y = b[i], z = c[i], w = d[i];
        x.x += y.x * 0.5f; x.y += y.y * 0.5f; x.z += y.z * 0.5f; x.w += y.w * 0.5f;
        z.x = z.x * 0.9f + y.x; z.y = z.y * 0.9f + y.y; z.z = z.z * 0.9f + y.z; z.w = z.w * 0.9f + y.w;
        w.x = w.x * 0.99f + y.x * y.x; w.y = w.y * 0.99f + y.y * y.y; w.z = w.z * 0.99f + y.z * y.z; w.w = w.w * 0.99f + y.w * y.w;
        a[i] = x; c[i] = z; d[i] = w;
        for (int k = 0; k < pace; ++k) __builtin_amdgcn_s_sleep(64);
    }
}

extern "C" int stream_launch(int n, void* a, const void* b, void* c, void* d, long total, int pace, void* stream) {
    stream_kernel<<<n, 256, 0, (hipStream_t)stream>>>((float4*)a, (const float4*)b, (float4*)c, (float4*)d, total, pace);
    return (int)hipGetLastError();
}

extern "C" int occupy_launch(int n, int regs, int dyn_lds, unsigned long long ticks, float* sink, void* stream) {
    if (dyn_lds > 48 * 1024) {
        hipFuncSetAttribute((const void*)occupy_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)occupy_kernel<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (regs <= 32) occupy_kernel<32><<<n, 256, dyn_lds, (hipStream_t)stream>>>(ticks, sink);
    else occupy_kernel<120><<<n, 256, dyn_lds, (hipStream_t)stream>>>(ticks, sink);
    return (int)hipGetLastError();
}
