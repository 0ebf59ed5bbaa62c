// Follow-up to overlap_probe: independent MFMA chains per wave, VALU interleaved in the SAME wave, and LDS reads next to MFMAs.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// MODE bits: NCH = independent accumulator chains (1, 2, 4); KV = VALU fmas issued after every MFMA in the same wave (independent of the MFMAs);
// KL = 1: one ds_read_b128 per MFMA (value unused by the MFMA: pure LDS-pipe load)
template <int NCH, int KV, int KL>
__global__ __launch_bounds__(512) void probe(float* out, unsigned mfma_mask, unsigned valu_mask, int n_mfma, int n_valu, int reps) {
    __shared__ float4 lds[4096];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = make_float4(i, 1, 2, 3);
    __syncthreads();
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    bf16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x & 3); b[e] = (__bf16)1.0f; }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = threadIdx.x * 0.001f + i;
    float4 l = make_float4(0, 0, 0, 0);
    for (int rep = 0; rep < reps; ++rep) {
        if ((mfma_mask >> wave) & 1u) {
            for (int i = 0; i < n_mfma; i += NCH) {
#pragma unroll
                for (int c = 0; c < NCH; ++c) {
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int k = 0; k < KV; ++k) v[k % 8] = __builtin_fmaf(v[k % 8], 1.0001f, 0.5f);    // 8 independent chains
                    if (KL) { const float4 t = lds[(threadIdx.x * 9 + i + c) & 4095]; l.x += t.x; }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if ((valu_mask >> wave) & 1u) {
#pragma unroll 4
            for (int i = 0; i < n_valu; ++i) {
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = __builtin_fmaf(v[k], 1.0001f, 0.5f);
            }
        }
        if (rep & 1) __syncthreads();
    }
    float s = l.x;
    for (int i = 0; i < 8; ++i) s += v[i];
    for (int c = 0; c < NCH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NCH, int KV, int KL>
static float run(unsigned mm, unsigned vm, int nm, int nv, int reps, float* out) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NCH, KV, KL><<<256, 512>>>(out, mm, vm, nm, nv, reps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<NCH, KV, KL><<<256, 512>>>(out, mm, vm, nm, nv, reps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f;
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    const int nm = 32, reps = 200;                 // per rep per MFMA wave: 32 MFMAs
    const double per = 1e3 / (nm * reps);          // us -> ns per MFMA of ONE wave
    printf("ns per MFMA (one wave's count); 17 ns = pipe-bound for 2 waves per SIMD at this clock\n");
    printf("1 wave/SIMD  1 chain             : %6.1f\n", run<1, 0, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  2 chains            : %6.1f\n", run<2, 0, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  4 chains            : %6.1f\n", run<4, 0, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("2 waves/SIMD 1 chain             : %6.1f\n", run<1, 0, 0>(0xff, 0, nm, 0, reps, out) * per);
    printf("2 waves/SIMD 2 chains            : %6.1f\n", run<2, 0, 0>(0xff, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  2 chains + 2 fma/MFMA same wave : %6.1f\n", run<2, 2, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  2 chains + 4 fma/MFMA same wave : %6.1f\n", run<2, 4, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  2 chains + 8 fma/MFMA same wave : %6.1f\n", run<2, 8, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  4 chains + 8 fma/MFMA same wave : %6.1f\n", run<4, 8, 0>(0x0f, 0, nm, 0, reps, out) * per);
    printf("2 waves/SIMD 1 chain  + 4 fma/MFMA same wave : %6.1f\n", run<1, 4, 0>(0xff, 0, nm, 0, reps, out) * per);
    printf("2 waves/SIMD 1 chain  + 8 fma/MFMA same wave : %6.1f\n", run<1, 8, 0>(0xff, 0, nm, 0, reps, out) * per);
    printf("1 wave/SIMD  2 chains + ds_read_b128/MFMA    : %6.1f\n", run<2, 0, 1>(0x0f, 0, nm, 0, reps, out) * per);
    printf("2 waves/SIMD 1 chain  + ds_read_b128/MFMA    : %6.1f\n", run<1, 0, 1>(0xff, 0, nm, 0, reps, out) * per);
    printf("2 waves/SIMD 2 chains + ds_read_b128/MFMA    : %6.1f\n", run<2, 0, 1>(0xff, 0, nm, 0, reps, out) * per);
    // waves 0-3: 2-chain MFMA; waves 4-7: VALU (32 * 8 fmas per rep = 256 VALU per 32 MFMAs)
    printf("waves 0-3 2-chain MFMA alone     : %6.1f\n", run<2, 0, 0>(0x0f, 0, nm, 32, reps, out) * per);
    printf("waves 4-7 VALU alone (256 fma)   : %6.1f\n", run<2, 0, 0>(0, 0xf0, nm, 32, reps, out) * per);
    printf("waves 0-3 2-chain MFMA | 4-7 VALU: %6.1f\n", run<2, 0, 0>(0x0f, 0xf0, nm, 32, reps, out) * per);
    printf("waves 0-3 1-chain MFMA | 4-7 VALU: %6.1f\n", run<1, 0, 0>(0x0f, 0xf0, nm, 32, reps, out) * per);
    return 0;
}
