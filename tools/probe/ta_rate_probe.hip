// What does the CU's vector-memory path deliver per clock, by load width?  One 512-thread workgroup per CU streams an L2-resident region
// (2 MiB shared by all workgroups: every access misses the 32 KB L1 and hits L2) with dword / dwordx2 / dwordx4 loads, U independent loads in
// flight per lane; prints bytes per clock and CU (s_memtime cycles of wave 0) and GB/s over all CUs.
// hipcc --offload-arch=gfx950 -O3 -o tools/probe/ta_rate_probe tools/probe/ta_rate_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
template <typename T, int U>
__global__ __launch_bounds__(512) void k(const T* __restrict__ x, long nelem, int iters, float* out, unsigned long long* cyc) {
    const int tid = threadIdx.x;
    float s = 0.f;
    const unsigned long long t0 = __builtin_readcyclecounter();
    long base = ((long)blockIdx.x * 8191) % nelem;
    for (int it = 0; it < iters; ++it) {
        T v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            long i = base + (long)u * 512 + tid; if (i >= nelem) i -= nelem;
            v[u] = x[i];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) { const float* f = (const float*)&v[u]; for (int e = 0; e < (int)(sizeof(T) / 4); ++e) s += f[e]; }
        base += U * 512; if (base >= nelem) base -= nelem;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (s == 123456.789f) out[blockIdx.x] = s;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <typename T, int U>
void run(const char* name, const void* x, long bytes, int blocks, float* out, unsigned long long* cyc) {
    const int iters = 2000 / U;
    const long nelem = bytes / sizeof(T);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<T, U><<<blocks, 512>>>((const T*)x, nelem, iters, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<T, U><<<blocks, 512>>>((const T*)x, nelem, iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double c = 0; for (int i = 0; i < blocks; ++i) c += (double)h[i]; c /= blocks;
    const double per_wg = (double)iters * U * 512 * sizeof(T);
    printf("%-8s U=%d blocks=%4d: %6.1f bytes per cycle and CU (%.0f cycles), %7.1f GB/s over the chip, %.0f MHz implied\n", name, U, blocks, per_wg / c, c,
           per_wg * blocks / (ms * 1e-3) / 1e9, c / (ms * 1e-3) / 1e6);
}
int main(int argc, char** argv) {
    const long bytes = (argc > 1 ? atol(argv[1]) : 2) << 20;
    void* x; hipMalloc(&x, bytes); hipMemset(x, 0, bytes);
    float* out; hipMalloc(&out, 4096 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 1024 * 8);
    for (int blocks : {1, 256, 512}) {
        run<float, 8>("dword", x, bytes, blocks, out, cyc);
        run<f2, 8>("dwordx2", x, bytes, blocks, out, cyc);
        run<f4, 4>("dwordx4", x, bytes, blocks, out, cyc);
        run<f4, 8>("dwordx4", x, bytes, blocks, out, cyc);
    }
    return 0;
}
