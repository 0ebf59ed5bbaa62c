// gemm_nt128.hip — C[M][N] (+)= sum_k A[m][k] B[n][k] with BOTH operands bf16 and k-major (the reduction index contiguous): the large
// weight-gradient GEMMs of the recurrent decoder on the transposed mirrors of its state rows (rows = features, columns = tokens;
// reference: nn.RNN's weight gradients, hulc2/models/decoders/utils/rnn.py:5-14 through autograd), and any other large NT product.
//
// The recipe of gridconv.hip (which reaches 650-750 TFLOP/s with it): 128 x 128 workgroup tile, 4 waves of 64 x 64 (2 x 2 accumulators of
// 32 x 32), k-steps of 64, operand tiles [row][64 k] in LDS with 144-byte rows — one conflict-free ds_read_b128 per MFMA operand, against the
// two ds_read_b64_tr_b16 the row-major tiles of gemm_tn128 need — two LDS stages, the next k-step's eight 16-byte loads per thread in NAMED
// registers in front of the 16 MFMAs of the current one.  Epilogue: fp32 store or accumulate; the first column block also sums its A rows
// (the bias gradient) at the time it writes them to LDS, in a fixed order.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

constexpr int NT_B = 128;                // tile edge
constexpr int NT_RS = 64 * 2 + 16;       // LDS row stride (144 B)

struct NtP {
    const uint16_t* A; const uint16_t* B; float* C;
    long lda, ldb, ldc;
    int M, N, K;
    int accumulate;
    float* rowsum; int rowsum_accumulate;
};

HULC_DEVICE float sum8_bf16(const uint4& v) {
    return ((__uint_as_float(v.x << 16) + __uint_as_float(v.x & 0xffff0000u)) + (__uint_as_float(v.y << 16) + __uint_as_float(v.y & 0xffff0000u))) +
           ((__uint_as_float(v.z << 16) + __uint_as_float(v.z & 0xffff0000u)) + (__uint_as_float(v.w << 16) + __uint_as_float(v.w & 0xffff0000u)));
}

template <int TMW>      // 32-row accumulator tiles per wave along M: 2 -> 128-row workgroup tile, 1 -> 64-row (twice the workgroups: two per CU)
__global__ __launch_bounds__(256) void gemm_nt128_kernel(NtP p) {
    constexpr int BM = TMW * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [stage][A tile | B tile], rows of 144 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    // blockIdx.x walks the columns fastest: neighbouring workgroups share an A panel
    const int n0 = blockIdx.x * NT_B, m0 = blockIdx.y * BM;
    const bool do_rowsum = p.rowsum != nullptr && blockIdx.x == 0;
    f32x16_t acc[TMW][2];
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    // this thread's chunks: rows (tid / 8) + 32 q (q = 0..3), 16-byte chunk tid % 8 of the 64-k row — the same pattern for both tiles
    const int ch = tid & 7, rw = tid >> 3;
    const uint16_t* ga = p.A + (long)(m0 + rw) * p.lda + ch * 8;
    const uint16_t* gb = p.B + (long)(n0 + rw) * p.ldb + ch * 8;
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    ra2 = ra3 = make_uint4(0u, 0u, 0u, 0u);
    float rs0 = 0.f, rs1 = 0.f, rs2 = 0.f, rs3 = 0.f;
#define NT_LOAD(ks_)                                                                                                   \
    {                                                                                                                  \
        const long k_ = (long)(ks_) * 64;                                                                              \
        ra0 = *(const uint4*)(ga + k_); ra1 = *(const uint4*)(ga + 32 * p.lda + k_);                                   \
        if (TMW == 2) { ra2 = *(const uint4*)(ga + 64 * p.lda + k_); ra3 = *(const uint4*)(ga + 96 * p.lda + k_); }    \
        rb0 = *(const uint4*)(gb + k_); rb1 = *(const uint4*)(gb + 32 * p.ldb + k_);                                   \
        rb2 = *(const uint4*)(gb + 64 * p.ldb + k_); rb3 = *(const uint4*)(gb + 96 * p.ldb + k_);                      \
    }
#define NT_STORE(stage_, live_)                                                                                        \
    {                                                                                                                  \
        char* As_ = smem + (stage_) * (BM + NT_B) * NT_RS + rw * NT_RS + ch * 16;                                      \
        char* Bs_ = As_ + BM * NT_RS;                                                                                  \
        *(uint4*)(As_) = ra0; *(uint4*)(As_ + 32 * NT_RS) = ra1;                                                       \
        if (TMW == 2) { *(uint4*)(As_ + 64 * NT_RS) = ra2; *(uint4*)(As_ + 96 * NT_RS) = ra3; }                        \
        *(uint4*)(Bs_) = rb0; *(uint4*)(Bs_ + 32 * NT_RS) = rb1; *(uint4*)(Bs_ + 64 * NT_RS) = rb2; *(uint4*)(Bs_ + 96 * NT_RS) = rb3;   \
        if (do_rowsum) {                                                                                               \
            const float f_ = (live_) ? 1.f : 0.f;                                                                      \
            rs0 += f_ * sum8_bf16(ra0); rs1 += f_ * sum8_bf16(ra1);                                                    \
            if (TMW == 2) { rs2 += f_ * sum8_bf16(ra2); rs3 += f_ * sum8_bf16(ra3); }                                  \
        }                                                                                                              \
    }
    const int nk = p.K / 64;
    NT_LOAD(0)
    NT_STORE(0, true)
    __syncthreads();
    const int r = lane & 31, h = lane >> 5;
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const int nx = ks + 1 < nk ? ks + 1 : ks;                  // last trip reloads its own tile into the other stage: nobody reads it
        NT_LOAD(nx)
        __builtin_amdgcn_sched_barrier(0);
        const char* As = smem + cur * (BM + NT_B) * NT_RS + wm * TMW * 32 * NT_RS;
        const char* Bs = smem + cur * (BM + NT_B) * NT_RS + BM * NT_RS + wn * 64 * NT_RS;
#pragma unroll
        for (int q = 0; q < 4; ++q) {                              // 16 k per step
            bf16x8_t a[TMW], b[2];
#pragma unroll
            for (int i = 0; i < TMW; ++i) a[i] = *(const bf16x8_t*)(As + (i * 32 + r) * NT_RS + (q * 2 + h) * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[j] = *(const bf16x8_t*)(Bs + (j * 32 + r) * NT_RS + (q * 2 + h) * 16);
#pragma unroll
            for (int i = 0; i < TMW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        NT_STORE(cur ^ 1, ks + 1 < nk)
        __syncthreads();
    }
#undef NT_LOAD
#undef NT_STORE
    // ---- epilogue
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int n = n0 + wn * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TMW + i) * 32 + acc_row(e, lane);
                float* dst = p.C + (long)m * p.ldc + n;
                *dst = p.accumulate ? *dst + acc[i][j][e] : acc[i][j][e];
            }
        }
    if (do_rowsum) {                                               // the 8 chunk-threads of a row are 8 neighbouring lanes: fixed-order butterfly
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) {
            rs0 += __shfl_xor(rs0, o, 64); rs1 += __shfl_xor(rs1, o, 64); rs2 += __shfl_xor(rs2, o, 64); rs3 += __shfl_xor(rs3, o, 64);
        }
        if (ch == 0) {
            float* d = p.rowsum + m0 + rw;
            d[0] = p.rowsum_accumulate ? d[0] + rs0 : rs0; d[32] = p.rowsum_accumulate ? d[32] + rs1 : rs1;
            if (TMW == 2) { d[64] = p.rowsum_accumulate ? d[64] + rs2 : rs2; d[96] = p.rowsum_accumulate ? d[96] + rs3 : rs3; }
        }
    }
}

}  // namespace

// internal entry used by hulc_gemm (gemm.hip): returns 1 when the shape was taken, 0 when the generic kernel must run, < 0 on error
int hulc_gemm_nt128_try(const hulc_gemm_desc* d, hipStream_t s) {
    if (getenv("HULC_NO_GEMM_NT128")) return 0;
    if (d->compute != HULC_BF16 || !d->a_kmajor || !d->b_kmajor || d->a_dtype != HULC_BF16 || d->b_dtype != HULC_BF16 || d->c_dtype != HULC_F32) return 0;
    if (d->bias || d->add || d->mask || d->relu || d->alpha != 1.0f || d->drop_p > 0.f) return 0;
    if (d->M % NT_B || d->N % NT_B || d->K % 64 || d->M < 512 || d->N < 512 || d->K < 512) return 0;
    if (((uintptr_t)d->A | (uintptr_t)d->B) % 16 || d->lda % 8 || d->ldb % 8) return 0;
    NtP p;
    p.A = (const uint16_t*)d->A; p.B = (const uint16_t*)d->B; p.C = (float*)d->C;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.M = d->M; p.N = d->N; p.K = d->K;
    p.accumulate = d->accumulate; p.rowsum = d->rowsum_a; p.rowsum_accumulate = d->rowsum_accumulate;
    // 64-row tiles when 128-row ones would leave one workgroup per CU: a k-step's loads need ~1 us, its 16 MFMAs 0.2 — a second workgroup
    // on the CU computes meanwhile
    static const int min_wg = getenv("HULC_NT128_MINWG") ? atoi(getenv("HULC_NT128_MINWG")) : 512;
    const bool small = (long)(d->M / NT_B) * (d->N / NT_B) < min_wg;
    const size_t lds = (size_t)2 * ((small ? 64 : 128) + NT_B) * NT_RS;
    static bool attr[2] = {false, false};
    if (!attr[small]) {
        const void* k = small ? (const void*)gemm_nt128_kernel<1> : (const void*)gemm_nt128_kernel<2>;
        if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return hulc_fail(-8, "hulc_gemm: could not raise the dynamic LDS limit (nt128)");
        attr[small] = true;
    }
    if (small) gemm_nt128_kernel<1><<<dim3(d->N / NT_B, d->M / 64), 256, lds, s>>>(p);
    else gemm_nt128_kernel<2><<<dim3(d->N / NT_B, d->M / NT_B), 256, lds, s>>>(p);
    return 1;
}
