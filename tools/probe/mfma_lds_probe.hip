// How fast does ONE wave per SIMD issue v_mfma_f32_32x32x16_bf16 when its B operands come out of the LDS through a ring of prefetched
// ds_read_b128 (consumed RD steps after they were issued, as the band kernels do) — per number of independent accumulator chains and reads per MFMA?
// (round 6: the question behind conv_band4.hip; overlap_probe2's LDS variant consumed every read where it was issued.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;

// NCH chains; RPM = LDS reads per group of NCH MFMAs (0: none, 1: one shared B, NCH: one B per MFMA); PS = pixel stride in bytes
template <int NCH, int RPM, int NT>
__global__ __launch_bounds__(NT) void probe(float* out, int steps, int reps, int ps) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    for (int i = threadIdx.x; i < 40960 / 4; i += NT) ((float*)lds)[i] = 1.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63, r = lane & 31, h = lane >> 5;
    const char* a0 = lds + r * ps + h * 16;
    f32x16 acc[NCH];
    for (int c = 0; c < NCH; ++c) for (int e = 0; e < 16; ++e) acc[c][e] = 0.f;
    bf16x8 a;
    for (int e = 0; e < 8; ++e) a[e] = (__bf16)(float)(threadIdx.x & 3);
    constexpr int RD = 8;
    constexpr int NR = RPM == 0 ? 1 : RPM;
    for (int rep = 0; rep < reps; ++rep) {
        bf16x8 pf[RD][NR];
#pragma unroll
        for (int i = 0; i < RD; ++i)
#pragma unroll
            for (int j = 0; j < NR; ++j) pf[i][j] = *(const bf16x8*)(a0 + (i * NR + j) * 32);
#pragma unroll
        for (int ks = 0; ks < 32; ++ks) {
#pragma unroll
            for (int c = 0; c < NCH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, pf[ks % RD][RPM == NCH ? c : 0], acc[c], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (RPM && ks + RD < 32) {
#pragma unroll
                for (int j = 0; j < NR; ++j) pf[ks % RD][j] = *(const bf16x8*)(a0 + ((ks + RD) * NR + j) * 32 + (rep & 1) * 4096);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0.f;
    for (int c = 0; c < NCH; ++c) for (int e = 0; e < 16; ++e) s += acc[c][e];
    if (s == 12345.678f) out[threadIdx.x] = s;
}

template <int NCH, int RPM, int NT>
static void run(const char* what, float* out, int ps) {
    const int reps = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    probe<NCH, RPM, NT><<<256, NT, 65536>>>(out, 32, reps, ps);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<NCH, RPM, NT><<<256, NT, 65536>>>(out, 32, reps, ps);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double mfma_per_simd = (double)reps * 32 * NCH * (NT / 256);
    printf("%-64s ps=%3d : %6.1f ns per MFMA per SIMD  (%5.0f TFLOP/s chip)\n", what, ps, ms * 1e6 / mfma_per_simd, 1024.0 * mfma_per_simd * 32768 / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; (void)hipMalloc(&out, 4096);
    for (int ps : {144, 80}) {
        run<1, 0, 256>("1 wave/SIMD, 1 chain, no LDS", out, ps);
        run<2, 0, 256>("1 wave/SIMD, 2 chains, no LDS", out, ps);
        run<4, 0, 256>("1 wave/SIMD, 4 chains, no LDS", out, ps);
        run<1, 1, 256>("1 wave/SIMD, 1 chain, 1 read per MFMA", out, ps);
        run<2, 1, 256>("1 wave/SIMD, 2 chains, 1 read per 2 MFMAs", out, ps);
        run<2, 2, 256>("1 wave/SIMD, 2 chains, 1 read per MFMA", out, ps);
        run<4, 1, 256>("1 wave/SIMD, 4 chains, 1 read per 4 MFMAs", out, ps);
        run<4, 4, 256>("1 wave/SIMD, 4 chains, 1 read per MFMA", out, ps);
        run<1, 1, 512>("2 waves/SIMD, 1 chain each, 1 read per MFMA", out, ps);
        run<2, 1, 512>("2 waves/SIMD, 2 chains each, 1 read per 2 MFMAs", out, ps);
        run<1, 0, 512>("2 waves/SIMD, 1 chain each, no LDS", out, ps);
    }
    return 0;
}
