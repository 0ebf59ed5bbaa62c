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

extern "C" int occupy_launch(int n, int regs, int dyn_lds, unsigned long long ticks, float* sink, void* stream) {
    if (dyn_lds > 48 * 1024) {
        hipFuncSetAttribute((const void*)occupy_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void*)occupy_kernel<120>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (regs <= 32) occupy_kernel<32><<<n, 256, dyn_lds, (hipStream_t)stream>>>(ticks, sink);
    else occupy_kernel<120><<<n, 256, dyn_lds, (hipStream_t)stream>>>(ticks, sink);
    return (int)hipGetLastError();
}
