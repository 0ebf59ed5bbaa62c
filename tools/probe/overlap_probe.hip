// Do VALU instructions of one wave overlap the MFMAs of ANOTHER wave on the same SIMD, and which waves of a 512-thread workgroup share a SIMD?
// Wave roles by mask: bit w of mfma_mask -> wave w runs a dependent MFMA chain; bit w of valu_mask -> wave w runs a VALU chain.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/overlap_probe.hip -o tools/probe/overlap_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

__global__ __launch_bounds__(512) void probe(float* out, unsigned mfma_mask, unsigned valu_mask, int n_mfma, int n_valu, int reps) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    float v0 = threadIdx.x * 0.001f, v1 = 1.0001f, v2 = 0.5f, v3 = 0.25f;
    for (int rep = 0; rep < reps; ++rep) {
        if ((mfma_mask >> wave) & 1u) {
            for (int i = 0; i < n_mfma; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        }
        if ((valu_mask >> wave) & 1u) {
#pragma unroll 8
            for (int i = 0; i < n_valu; ++i) {      // four independent chains: issue-bound, not latency-bound
                v0 = __builtin_fmaf(v0, v1, v2); v1 = __builtin_fmaf(v1, v2, v3); v2 = __builtin_fmaf(v2, v3, v0); v3 = __builtin_fmaf(v3, v0, v1);
            }
        }
        if (rep & 1) __syncthreads();       // (phases: both run in every rep; barrier keeps the workgroup together like the band kernel)
    }
    float s = v0 + v1 + v2 + v3;
    for (int e = 0; e < 16; ++e) s += acc[e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

static float run(unsigned mm, unsigned vm, int nm, int nv, int reps, float* out) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<<<256, 512>>>(out, mm, vm, nm, nv, reps);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<<<256, 512>>>(out, mm, vm, nm, nv, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out; hipMalloc(&out, 4096);
    const int nm = 36, nv = 72, reps = 200;       // 36 MFMAs (1152 pipe cycles) vs 288 VALU ops (1152 issue cycles) per rep
    printf("all 8 waves MFMA only            : %8.1f us\n", run(0xff, 0, nm, nv, reps, out));
    printf("all 8 waves VALU only            : %8.1f us\n", run(0, 0xff, nm, nv, reps, out));
    printf("all 8 waves MFMA then VALU       : %8.1f us\n", run(0xff, 0xff, nm, nv, reps, out));
    printf("waves 0-3 MFMA only              : %8.1f us\n", run(0x0f, 0, nm, nv, reps, out));
    printf("waves 0-3 VALU only              : %8.1f us\n", run(0, 0x0f, nm, nv, reps, out));
    printf("waves 0-3 MFMA | waves 4-7 VALU  : %8.1f us\n", run(0x0f, 0xf0, nm, nv, reps, out));
    printf("even waves MFMA | odd waves VALU : %8.1f us\n", run(0x55, 0xaa, nm, nv, reps, out));
    printf("waves 0,1,4,5 MFMA | 2,3,6,7 VALU: %8.1f us\n", run(0x33, 0xcc, nm, nv, reps, out));
    printf("waves 0-3 MFMA x2 | 4-7 VALU x2   : %8.1f us  (same total work as 'all 8 MFMA then VALU')\n", run(0x0f, 0xf0, 2 * nm, 2 * nv, reps, out));
    printf("even MFMA x2 | odd VALU x2        : %8.1f us\n", run(0x55, 0xaa, 2 * nm, 2 * nv, reps, out));
    return 0;
}
