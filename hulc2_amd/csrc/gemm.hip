// gemm.hip — LDS-tiled MFMA GEMM for every dense layer of the HULC++ low-level policy.
//
//   C[M,N] = epilogue( alpha * sum_k A[m,k] * B[n,k] )
//
// serves, with one kernel template:
//   * forward Linear  y = x W^T + b              (A = x  k-major, B = W   k-major)
//       reference: nn.Linear in hulc2/models/plan_encoders/plan_proposal_net.py:26-40,
//       goal_encoders.py:21-27, vision_network.py:49-52, logistic_decoder_rnn.py:60-62
//   * data gradient   dx = dy W                   (A = dy k-major, B = W   stored [K][N])
//   * weight gradient dW = dy^T x                 (A = dy stored [K][M], B = x stored [K][N])
//   * ReLU-RNN step   h_t = relu(pre_t + h_{t-1} W_hh^T)  (add + relu epilogue)
//       reference: nn.RNN(nonlinearity="relu") hulc2/models/decoders/utils/rnn.py:5-14
//
// Tile: each wave owns TM x TN accumulators of 32x32 (v_mfma_f32_32x32x16_bf16 or the exact
// v_mfma_f32_32x32x2_f32); WM x WN waves per workgroup; operands are converted to the compute type
// while staged (global -> registers -> LDS, double-buffered, next tile's loads in flight during MFMA).
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

struct GemmP {
    const void* A; const void* B; void* C;
    const float* bias; const void* add; const void* mask;
    int M, N, K;
    long lda, ldb, ldc, ld_add, ld_mask;
    int a_dtype, b_dtype, c_dtype, add_dtype, mask_dtype;
    int relu, accumulate;
    float alpha, mask_scale, drop_p;
    unsigned long long drop_seed;
    const unsigned long long* seed_dev;   // optional device word xor-ed into drop_seed (graph-replay safe RNG stream)
    float* rowsum; int rowsum_accumulate;  // optional: rowsum[m] (+)= sum_k A[m][k] (row-major A only): the bias gradient of a weight-gradient GEMM
};

// Load the 8-element k-chunk (row r, k0..k0+7) of an operand.
//   KMAJOR : operand stored [rows][K], k contiguous   -> one or two 16-byte loads
//   !KMAJOR: operand stored [K][rows], row contiguous -> 8 strided loads (coalesced across lanes)
template <bool KMAJOR>
HULC_DEVICE void load_operand_chunk(Chunk8& c, const void* base, int dtype, long ld, int rows, int K, int r, int k0) {
    const bool in_k = k0 < K;
    const int kc = in_k ? k0 : 0;  // branch-free: load from a clamped address, zero afterwards (see chunk_keep_if)
    r = r < rows ? r : rows - 1;   // clamp: out-of-range rows are computed but never stored
    if (KMAJOR) {
        chunk_load_contig(c, base, dtype, (long)r * ld + kc);
        chunk_keep_if(c, in_k);
    } else {
        const int nvalid = in_k ? K - k0 : 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = (j < nvalid) ? kc + j : 0;
            const float v = load_elem(base, dtype, (long)kk * ld + r);
            c.v[j] = (j < nvalid) ? v : 0.f;
        }
    }
}

template <typename CT, int TM, int TN, int WM, int WN, bool AK, bool BK>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmP p, float* __restrict__ slabs, int splitk) {
    using T = MmaTraits<CT>;
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int KT = T::KT, NCH = T::NCH, CHB = T::CHB;
    constexpr int A_CH = BM * NCH, B_CH = BN * NCH;           // chunks per tile
    constexpr int A_PER = (A_CH + NT - 1) / NT, B_PER = (B_CH + NT - 1) / NT;

    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * HULC_ROWB];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    // blockIdx.x walks M fastest so that blocks sharing a weight (B) panel are co-resident
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // Row-major (reduction-major) fp32 operands — the weight-gradient GEMMs dW = dY^T X — are staged as 8(k) x 4(rows)
    // micro-tiles: eight 16-byte loads per thread give four complete k-chunks (an in-register transpose), instead of
    // 32 scalar loads.  lane -> (k-group fastest, row group) keeps the LDS chunk writes conflict-free and each load
    // instruction a set of 256-byte row segments.  A micro-tiles go to the first threads, B micro-tiles to the next.
    // Row-major bf16 operands (the recurrent decoder's weight gradients read the bf16 state copies) use 8(k) x 8(rows) micro-tiles:
    // eight 16-byte loads, a 16-bit 8x8 transpose with v_perm, eight 16-byte LDS chunk writes.
    constexpr int MTA = BM / 4 * NCH, MTB = BN / 4 * NCH;                 // fp32 micro-tiles per operand tile
    constexpr int MTA16 = BM / 8 * NCH, MTB16 = BN / 8 * NCH;             // bf16 micro-tiles
    constexpr bool IS_BF16 = sizeof(CT) == 2;
    const bool a_micro = !AK && p.a_dtype == HULC_F32 && p.lda % 4 == 0 && ((uintptr_t)p.A % 16) == 0 && m0 + BM <= p.M && m0 % 4 == 0;
    const bool b_micro = !BK && p.b_dtype == HULC_F32 && p.ldb % 4 == 0 && ((uintptr_t)p.B % 16) == 0 && n0 + BN <= p.N && n0 % 4 == 0;
    const bool a_micro16 = IS_BF16 && !AK && p.a_dtype == HULC_BF16 && p.lda % 8 == 0 && ((uintptr_t)p.A % 16) == 0 && m0 + BM <= p.M;
    const bool b_micro16 = IS_BF16 && !BK && p.b_dtype == HULC_BF16 && p.ldb % 8 == 0 && ((uintptr_t)p.B % 16) == 0 && n0 + BN <= p.N;
    const int a_thr = a_micro16 ? MTA16 : (a_micro ? MTA : 0), b_thr = b_micro16 ? MTB16 : (b_micro ? MTB : 0);
    const int MICRO_B0 = (a_thr + b_thr <= NT) ? a_thr : 0;                // first thread of the B micro-tiles
    union Stage { Chunk8 c[4]; uint4 q[8]; };                             // fp32 micro-tile (4 chunks) or bf16 micro-tile (8 raw rows)
    union StageA { Chunk8 c[AK ? A_PER : (A_PER > 4 ? A_PER : 4)]; uint4 q[8]; } sa;
    union StageB { Chunk8 c[BK ? B_PER : (B_PER > 4 ? B_PER : 4)]; uint4 q[8]; } sb;
    Chunk8* const ra = sa.c; Chunk8* const rb = sb.c;
    const int nkt_all = (p.K + KT - 1) / KT;
    const int kt_per = (nkt_all + splitk - 1) / splitk;            // blockIdx.z owns k tiles [kt0, kt1)
    const int kt0 = blockIdx.z * kt_per;
    const int kt1 = kt0 + kt_per < nkt_all ? kt0 + kt_per : nkt_all;

    // chunk id -> (row, chunk-in-row).  k-major sources: consecutive threads walk k first (one row's
    // 64..128 contiguous bytes per 2..4 threads); row-major sources: consecutive threads walk rows
    // (each strided load instruction is a contiguous row segment across the wave).
    auto a_map = [&](int id, int& r, int& ch) { if (AK) { r = id / NCH; ch = id % NCH; } else { r = id % BM; ch = id / BM; } };
    auto b_map = [&](int id, int& r, int& ch) { if (BK) { r = id / NCH; ch = id % NCH; } else { r = id % BN; ch = id / BN; } };

    // 8 x 4 micro-tile: chunks c[i] (i = row within the group) from eight float4 rows of the source
    auto load_micro = [&](Chunk8* c, const void* base, long ld, int row0, int k0, int K) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in_k = k0 + j < K;
            const float4 v = *(const float4*)((const float*)base + (long)(in_k ? k0 + j : 0) * ld + row0);
            c[0].v[j] = in_k ? v.x : 0.f; c[1].v[j] = in_k ? v.y : 0.f; c[2].v[j] = in_k ? v.z : 0.f; c[3].v[j] = in_k ? v.w : 0.f;
        }
    };
    auto load_micro16 = [&](uint4* q, const void* base, long ld, int row0, int k0, int K) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool in_k = k0 + j < K;
            const uint4 v = *(const uint4*)((const uint16_t*)base + (long)(in_k ? k0 + j : 0) * ld + row0);
            q[j].x = in_k ? v.x : 0u; q[j].y = in_k ? v.y : 0u; q[j].z = in_k ? v.z : 0u; q[j].w = in_k ? v.w : 0u;
        }
    };
    // 16-bit 8x8 transpose: chunk of row i = {q[0][i], ..., q[7][i]}; one v_perm per output dword
    auto store_micro16 = [&](char* dst_rows, const uint4* q, int ch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;
            uint4 o;
            const unsigned* w0 = (const unsigned*)&q[0];
#define HULC_QW(j) (((const unsigned*)&q[j])[i >> 1])
            o.x = __builtin_amdgcn_perm(HULC_QW(1), HULC_QW(0), sel);
            o.y = __builtin_amdgcn_perm(HULC_QW(3), HULC_QW(2), sel);
            o.z = __builtin_amdgcn_perm(HULC_QW(5), HULC_QW(4), sel);
            o.w = __builtin_amdgcn_perm(HULC_QW(7), HULC_QW(6), sel);
#undef HULC_QW
            (void)w0;
            *(uint4*)(dst_rows + i * HULC_ROWB + ch * 16) = o;
        }
    };
    // fused bias gradient: the workgroups of the first column block also sum the A rows they stage (fp32 operands: before
    // bf16 rounding)
    const bool do_rowsum = !AK && p.rowsum != nullptr && blockIdx.y == 0;
    float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    auto load_tiles = [&](int kt) {
        if (!AK && a_micro16) {
            if (tid < MTA16) {
                load_micro16(sa.q, p.A, p.lda, m0 + (tid / NCH) * 8, kt * KT + (tid % NCH) * 8, p.K);
                if (do_rowsum) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const unsigned w[4] = {sa.q[j].x, sa.q[j].y, sa.q[j].z, sa.q[j].w};
#pragma unroll
                        for (int i = 0; i < 8; ++i) rs[i] += __uint_as_float((i & 1) ? (w[i >> 1] & 0xffff0000u) : (w[i >> 1] << 16));
                    }
                }
            }
        } else if (!AK && a_micro) {
            if (tid < MTA) {
                load_micro(ra, p.A, p.lda, m0 + (tid / NCH) * 4, kt * KT + (tid % NCH) * 8, p.K);
                if (do_rowsum) {
#pragma unroll
                    for (int i = 0; i < 4; ++i)
#pragma unroll
                        for (int j = 0; j < 8; ++j) rs[i] += ra[i].v[j];
                }
            }
        } else
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            int id = tid + q * NT;
            if (A_CH % NT == 0 || id < A_CH) {
                int r, ch; a_map(id, r, ch);
                load_operand_chunk<AK>(ra[q], p.A, p.a_dtype, p.lda, p.M, p.K, m0 + r, kt * KT + ch * 8);
                if (!AK && do_rowsum && m0 + r < p.M) {           // row = id % BM is the same for every q of a thread (NT % BM == 0)
#pragma unroll
                    for (int j = 0; j < 8; ++j) rs[0] += ra[q].v[j];
                }
            }
        }
        if (!BK && b_micro16) {
            const int t = tid - MICRO_B0;
            if (t >= 0 && t < MTB16) load_micro16(sb.q, p.B, p.ldb, n0 + (t / NCH) * 8, kt * KT + (t % NCH) * 8, p.K);
        } else if (!BK && b_micro) {
            const int t = tid - MICRO_B0;
            if (t >= 0 && t < MTB) load_micro(rb, p.B, p.ldb, n0 + (t / NCH) * 4, kt * KT + (t % NCH) * 8, p.K);
        } else
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            int id = tid + q * NT;
            if (B_CH % NT == 0 || id < B_CH) {
                int r, ch; b_map(id, r, ch);
                load_operand_chunk<BK>(rb[q], p.B, p.b_dtype, p.ldb, p.N, p.K, n0 + r, kt * KT + ch * 8);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        char* As = smem + buf * (BM + BN) * HULC_ROWB;
        char* Bs = As + BM * HULC_ROWB;
        if (!AK && a_micro16) {
            if (tid < MTA16) store_micro16(As + (tid / NCH) * 8 * HULC_ROWB, sa.q, tid % NCH);
        } else if (!AK && a_micro) {
            if (tid < MTA) {
#pragma unroll
                for (int i = 0; i < 4; ++i) chunk_store_lds<CT>(As + ((tid / NCH) * 4 + i) * HULC_ROWB + (tid % NCH) * CHB, ra[i]);
            }
        } else
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            int id = tid + q * NT;
            if (A_CH % NT == 0 || id < A_CH) {
                int r, ch; a_map(id, r, ch);
                chunk_store_lds<CT>(As + r * HULC_ROWB + ch * CHB, ra[q]);
            }
        }
        if (!BK && b_micro16) {
            const int t = tid - MICRO_B0;
            if (t >= 0 && t < MTB16) store_micro16(Bs + (t / NCH) * 8 * HULC_ROWB, sb.q, t % NCH);
        } else if (!BK && b_micro) {
            const int t = tid - MICRO_B0;
            if (t >= 0 && t < MTB) {
#pragma unroll
                for (int i = 0; i < 4; ++i) chunk_store_lds<CT>(Bs + ((t / NCH) * 4 + i) * HULC_ROWB + (t % NCH) * CHB, rb[i]);
            }
        } else
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            int id = tid + q * NT;
            if (B_CH % NT == 0 || id < B_CH) {
                int r, ch; b_map(id, r, ch);
                chunk_store_lds<CT>(Bs + r * HULC_ROWB + ch * CHB, rb[q]);
            }
        }
    };

    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles(0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const char* As = smem + buf * (BM + BN) * HULC_ROWB;
        const char* Bs = As + BM * HULC_ROWB;
        MmaTile<CT, TM, TN>::run(As + wm * TM * 32 * HULC_ROWB, Bs + wn * TN * 32 * HULC_ROWB, acc, lane);
        if (kt + 1 < kt1) store_tiles(buf ^ 1);
        __syncthreads();
    }

    if (do_rowsum) {
        // partial row sums of this K slice -> rowsum slab [splitk][M] behind the C slabs (split K) or straight to the output
        float* dst = splitk > 1 ? slabs + (long)splitk * p.M * p.N + (long)blockIdx.z * p.M : p.rowsum;
        const bool acc_out = splitk == 1 && p.rowsum_accumulate;
        if (a_micro16) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                for (int o = 1; o < NCH; o <<= 1) rs[i] += __shfl_xor(rs[i], o, 64);
            if (tid < MTA16 && tid % NCH == 0) {
                const int r0 = m0 + (tid / NCH) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) dst[r0 + i] = acc_out ? dst[r0 + i] + rs[i] : rs[i];
            }
        } else if (a_micro) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                for (int o = 1; o < NCH; o <<= 1) rs[i] += __shfl_xor(rs[i], o, 64);       // the NCH k-groups of a row group are adjacent lanes
            if (tid < MTA && tid % NCH == 0) {
                const int r0 = m0 + (tid / NCH) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) dst[r0 + i] = acc_out ? dst[r0 + i] + rs[i] : rs[i];
            }
        } else {
            float* red = (float*)smem;                           // operand tiles are dead: NT floats of scratch
            __syncthreads();
            red[tid] = rs[0];
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float v = 0.f;
                for (int t = tid; t < NT && t < A_CH; t += BM) v += red[t];
                dst[m0 + tid] = acc_out ? dst[m0 + tid] + v : v;
            }
        }
    }

    if (splitk > 1) {   // raw partial sums; gemm_splitk_epilogue_kernel combines the slabs in a fixed order
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (n >= p.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + (wm * TM + i) * 32 + acc_row(e, lane);
                    if (m < p.M) slabs[((long)blockIdx.z * p.M + m) * p.N + n] = acc[i][j][e];
                }
        }
        return;
    }

    // epilogue: lane owns column n = lane & 31 of each tile; each accumulator register is one row.  All reads of the
    // tile's side operands (C when accumulating, add, mask) are issued before the first store: C is not provably distinct
    // from them for the compiler, so an interleaved read-modify-write loop serialises into one memory round trip per element.
    const unsigned long long seed = p.drop_seed ^ (p.seed_dev ? p.seed_dev[0] : 0ull);
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        const bool n_ok = n < p.N;
        const int nc = n_ok ? n : 0;
        const float bv = p.bias ? p.bias[nc] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            float cold[16], addv[16], maskv[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + acc_row(e, lane);
                const int mc = m < p.M ? m : 0;
                cold[e] = p.accumulate ? load_elem(p.C, p.c_dtype, (long)mc * p.ldc + nc) : 0.f;
                addv[e] = p.add ? load_elem(p.add, p.add_dtype, (long)mc * p.ld_add + nc) : 0.f;
                maskv[e] = p.mask ? load_elem(p.mask, p.mask_dtype, (long)mc * p.ld_mask + nc) : 1.f;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + (wm * TM + i) * 32 + acc_row(e, lane);
                if (m >= p.M || !n_ok) continue;
                float v = p.alpha * acc[i][j][e] + bv + addv[e];
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask) v = maskv[e] > 0.f ? v * p.mask_scale : 0.f;
                if (p.drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)m * (uint64_t)p.N + n, p.drop_p);
                store_elem(p.C, p.c_dtype, (long)m * p.ldc + n, v + cold[e]);
            }
        }
    }
}


// ------------------------------------------------------------------------------------------------
// skinny-M GEMM (M <= 64): the per-sequence layers (batch = 32 rows) and the recurrent step.
// These are weight-streaming problems: 0.27 GFLOP against an 8 MB weight panel, so the tile machinery
// above (few workgroups, long serial k loop) is the wrong shape.  Here every wave owns a 32-column
// strip and a private K slice, pulls its MFMA fragments straight from L2/HBM into registers (no LDS
// staging: nothing is shared between waves), keeps >= 12 loads in flight per lane, and the K slices
// are combined in a fixed order: 4 waves through LDS, SPLITK workgroups through fp32 slabs + a
// second tiny kernel that also applies the epilogue.  No atomics -> bit-reproducible.
// ------------------------------------------------------------------------------------------------
template <typename CT> struct Frag;
template <> struct Frag<bf16_t> {
    bf16x8_t v;
    HULC_DEVICE void set(const Chunk8& c) {
        union { uint4 u; bf16x8_t b; } x;
        x.u.x = pack_bf16x2(c.v[0], c.v[1]); x.u.y = pack_bf16x2(c.v[2], c.v[3]);
        x.u.z = pack_bf16x2(c.v[4], c.v[5]); x.u.w = pack_bf16x2(c.v[6], c.v[7]);
        v = x.b;
    }
};
template <> struct Frag<float> {
    float v[8];
    HULC_DEVICE void set(const Chunk8& c) {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = c.v[j];
    }
};
HULC_DEVICE void frag_mma(const Frag<bf16_t>& a, const Frag<bf16_t>& b, f32x16_t& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.v, b.v, acc, 0, 0, 0);
}
HULC_DEVICE void frag_mma(const Frag<float>& a, const Frag<float>& b, f32x16_t& acc) {
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[j], b.v[j], acc, 0, 0, 0);
}

HULC_DEVICE float gemm_epilogue(const GemmP& p, float acc, int m, int n) {
    float v = p.alpha * acc + (p.bias ? p.bias[n] : 0.f);
    if (p.add) v += load_elem(p.add, p.add_dtype, (long)m * p.ld_add + n);
    if (p.relu) v = fmaxf(v, 0.f);
    if (p.mask) v = load_elem(p.mask, p.mask_dtype, (long)m * p.ld_mask + n) > 0.f ? v * p.mask_scale : 0.f;
    if (p.drop_p > 0.f) v *= dropout_scale(p.drop_seed ^ (p.seed_dev ? p.seed_dev[0] : 0ull), (uint64_t)m * (uint64_t)p.N + n, p.drop_p);
    const long ci = (long)m * p.ldc + n;
    if (p.accumulate) v += load_elem(p.C, p.c_dtype, ci);
    return v;
}

template <typename CT, int TM, bool AK, bool BK, int NW>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmP p, float* __restrict__ slabs, int splitk, int kw) {
    __shared__ float red[NW][TM][32][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.x * 32;
    const int kbeg = (blockIdx.y * NW + (wave + blockIdx.x) % NW) * kw;   // this wave's K slice, rotated per column strip (L2 hot-spot avoidance)
    int kend = kbeg + kw; if (kend > p.K) kend = p.K;

    f32x16_t acc[TM];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;

    constexpr int UN = 4;                                      // k-steps (of 16) fetched per batch
    for (int k0 = kbeg; k0 < kend; k0 += 16 * UN) {
        Chunk8 ca[UN][TM], cb[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + u * 16 + h * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                if (k < kend) load_operand_chunk<AK>(ca[u][i], p.A, p.a_dtype, p.lda, p.M, kend, i * 32 + r, k);
                else chunk_zero(ca[u][i]);
            }
            if (k < kend) load_operand_chunk<BK>(cb[u], p.B, p.b_dtype, p.ldb, p.N, kend, n0 + r, k);
            else chunk_zero(cb[u]);
        }
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            Frag<CT> fb; fb.set(cb[u]);
#pragma unroll
            for (int i = 0; i < TM; ++i) { Frag<CT> fa; fa.set(ca[u][i]); frag_mma(fa, fb, acc[i]); }
        }
    }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) red[wave][i][acc_row(e, lane)][r] = acc[i][e];
    __syncthreads();
    for (int idx = tid; idx < TM * 1024; idx += NW * 64) {
        const int i = idx >> 10, row = (idx >> 5) & 31, col = idx & 31;
        const int m = i * 32 + row, n = n0 + col;
        if (m >= p.M || n >= p.N) continue;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) v += red[w][i][row][col];
        if (splitk == 1) store_elem(p.C, p.c_dtype, (long)m * p.ldc + n, gemm_epilogue(p, v, m, n));
        else slabs[((long)blockIdx.y * p.M + m) * p.N + n] = v;
    }
}

__global__ __launch_bounds__(256) void gemm_splitk_epilogue_kernel(GemmP p, const float* __restrict__ slabs, int splitk) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (p.rowsum && i < p.M) {                                   // fused bias gradient: row-sum slabs follow the C slabs
        const float* rsl = slabs + (long)splitk * p.M * p.N;
        float v = 0.f;
        for (int s = 0; s < splitk; ++s) v += rsl[(long)s * p.M + i];
        p.rowsum[i] = p.rowsum_accumulate ? p.rowsum[i] + v : v;
    }
    if (i >= (long)p.M * p.N) return;
    const int m = (int)(i / p.N), n = (int)(i % p.N);
    float v = 0.f;
    for (int s = 0; s < splitk; ++s) v += slabs[(long)s * p.M * p.N + i];
    store_elem(p.C, p.c_dtype, (long)m * p.ldc + n, gemm_epilogue(p, v, m, n));
}


// K slices: waves first (up to 16 per workgroup: combined through LDS in the same launch, epilogue fused),
// then workgroups (fp32 slabs + epilogue kernel) until ~256 workgroups exist; every wave keeps >= 64 k.
template <typename CT>
void launch_skinny(const GemmP& p, int ak, int bk, float* ws, long ws_bytes, hipStream_t s) {
    const int colblocks = (p.N + 31) / 32;
    const bool big = p.M <= 32 && p.K >= 1024;                  // 16 waves x TM=1: 66 KB of LDS for the reduction
    const bool big2 = !big && p.K >= 1024;                       // 33..64 rows: 8 waves x TM=2, same LDS footprint
    const int nw = big ? 16 : (big2 ? 8 : 4);
    int splitk = 1;
    const char* force = getenv("HULC_SKINNY_SPLITK");           // tuning knob for A/B runs
    const int min_kw = 64;   // measured (tools/rnn_bench.py): more, shorter K slices win even with the extra epilogue launch
    while (colblocks * splitk < 200 && p.K / (nw * (splitk * 2)) >= min_kw) splitk *= 2;
    if (force) splitk = atoi(force);
    while (splitk > 1 && (long)splitk * p.M * p.N * 4 > ws_bytes) splitk /= 2;
    int kw = (p.K + splitk * nw - 1) / (splitk * nw);
    kw = (kw + 15) / 16 * 16;
    dim3 grid(colblocks, splitk);
#define HULC_SK(TMv, AKv, BKv, NWv) gemm_skinny_kernel<CT, TMv, AKv, BKv, NWv><<<grid, NWv * 64, 0, s>>>(p, ws, splitk, kw)
    if (big) {
        if (ak && bk) HULC_SK(1, true, true, 16); else if (ak && !bk) HULC_SK(1, true, false, 16);
        else if (!ak && !bk) HULC_SK(1, false, false, 16); else HULC_SK(1, false, true, 16);
    } else if (big2) {
        if (ak && bk) HULC_SK(2, true, true, 8); else if (ak && !bk) HULC_SK(2, true, false, 8);
        else if (!ak && !bk) HULC_SK(2, false, false, 8); else HULC_SK(2, false, true, 8);
    } else if (p.M <= 32) {
        if (ak && bk) HULC_SK(1, true, true, 4); else if (ak && !bk) HULC_SK(1, true, false, 4);
        else if (!ak && !bk) HULC_SK(1, false, false, 4); else HULC_SK(1, false, true, 4);
    } else {
        if (ak && bk) HULC_SK(2, true, true, 4); else if (ak && !bk) HULC_SK(2, true, false, 4);
        else if (!ak && !bk) HULC_SK(2, false, false, 4); else HULC_SK(2, false, true, 4);
    }
#undef HULC_SK
    if (splitk > 1) {
        const long n = (long)p.M * p.N;
        gemm_splitk_epilogue_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, ws, splitk);
    }
}

template <typename CT, int TM, int TN, int WM, int WN>
void launch_cfg(const GemmP& p, int ak, int bk, float* ws, long ws_bytes, hipStream_t s) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    const int gx = (p.M + BM - 1) / BM, gy = (p.N + BN - 1) / BN;
    // few output tiles but a long reduction (dgrads into narrow layers, wgrads of narrow layers): split K over workgroups.
    // One workgroup per CU is latency-bound (a k-tile's loads are only covered by one tile of MFMAs): ~3 per CU (768) with >= 4
    // k-tiles per slice measured best (tools/gemm_heur_sweep.sh: step 6.52 -> 6.20 ms together with the short-K rule below)
    const int nkt = (p.K + MmaTraits<CT>::KT - 1) / MmaTraits<CT>::KT;
    int splitk = 1;
    static const int target = getenv("HULC_TILE_TARGET") ? atoi(getenv("HULC_TILE_TARGET")) : 768;
    static const int minkt = getenv("HULC_TILE_MINKT") ? atoi(getenv("HULC_TILE_MINKT")) : 4;
    while (gx * gy * splitk < target && nkt / (splitk * 2) >= minkt) splitk *= 2;
    while (splitk > 1 && (long)splitk * p.M * (p.N + 1) * 4 > ws_bytes) splitk /= 2;
    dim3 grid(gx, gy, splitk), block(WM * WN * 64);
    if (ak && bk) gemm_kernel<CT, TM, TN, WM, WN, true, true><<<grid, block, 0, s>>>(p, ws, splitk);
    else if (ak && !bk) gemm_kernel<CT, TM, TN, WM, WN, true, false><<<grid, block, 0, s>>>(p, ws, splitk);
    else if (!ak && !bk) gemm_kernel<CT, TM, TN, WM, WN, false, false><<<grid, block, 0, s>>>(p, ws, splitk);
    else gemm_kernel<CT, TM, TN, WM, WN, false, true><<<grid, block, 0, s>>>(p, ws, splitk);
    if (splitk > 1) {
        const long n = (long)p.M * p.N;
        gemm_splitk_epilogue_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p, ws, splitk);
    }
}

template <typename CT>
void launch_ct(const GemmP& p, int ak, int bk, float* ws, long ws_bytes, hipStream_t s) {
    // tile choice: keep >= ~256 workgroups when the problem allows it (256 CUs); M <= 64 never gets here
    // (skinny path above).
    const long blocks128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    // short reductions (K <= 256: weight gradients over 32 / 64 sequences, projections out of d_model = 128) are one or two k-tiles
    // of latency followed by a 64 KB store per tile: 64 x 64 tiles give 4x the workgroups to overlap them
    static const int smallk = getenv("HULC_TILE_SMALLK") ? atoi(getenv("HULC_TILE_SMALLK")) : 256;
    if (blocks128 < 192 || p.K <= smallk) launch_cfg<CT, 1, 1, 2, 2>(p, ak, bk, ws, ws_bytes, s);   // 64 x 64
    else launch_cfg<CT, 2, 2, 2, 2>(p, ak, bk, ws, ws_bytes, s);                    // 128 x 128
}

}  // namespace

extern "C" int hulc_gemm(const hulc_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->B || !d->C) return hulc_fail(-1, "hulc_gemm: null operand");
    if (d->M <= 0 || d->N <= 0 || d->K <= 0) return hulc_fail(-2, "hulc_gemm: non-positive dimension");
    if (d->K % 8 != 0 && (d->a_kmajor || d->b_kmajor)) return hulc_fail(-3, "hulc_gemm: K must be a multiple of 8 for k-major operands");
    const int asz = d->a_dtype == HULC_F32 ? 4 : 2, bsz = d->b_dtype == HULC_F32 ? 4 : 2;
    if (d->a_kmajor && (((uintptr_t)d->A % 16) || (d->lda * asz) % 16)) return hulc_fail(-4, "hulc_gemm: A not 16-byte aligned");
    if (d->b_kmajor && (((uintptr_t)d->B % 16) || (d->ldb * bsz) % 16)) return hulc_fail(-4, "hulc_gemm: B not 16-byte aligned");
    if (d->compute == HULC_F32 && (d->a_dtype != HULC_F32 || d->b_dtype != HULC_F32))
        return hulc_fail(-5, "hulc_gemm: f32 compute requires f32 operands");
    GemmP p;
    p.A = d->A; p.B = d->B; p.C = d->C; p.bias = d->bias; p.add = d->add; p.mask = d->mask;
    p.M = d->M; p.N = d->N; p.K = d->K;
    p.lda = d->lda; p.ldb = d->ldb; p.ldc = d->ldc; p.ld_add = d->ld_add; p.ld_mask = d->ld_mask;
    p.a_dtype = d->a_dtype; p.b_dtype = d->b_dtype; p.c_dtype = d->c_dtype;
    p.add_dtype = d->add_dtype; p.mask_dtype = d->mask_dtype;
    p.relu = d->relu; p.accumulate = d->accumulate;
    p.alpha = d->alpha; p.mask_scale = d->mask_scale; p.drop_p = d->drop_p; p.drop_seed = d->drop_seed;
    p.seed_dev = d->seed_dev;
    p.rowsum = d->rowsum_a; p.rowsum_accumulate = d->rowsum_accumulate;
    if (p.rowsum && (d->a_kmajor || d->M <= 64)) return hulc_fail(-6, "hulc_gemm: rowsum_a needs a row-major A operand and M > 64 (tiled path)");
    hipStream_t s = (hipStream_t)stream;
    if (d->M <= 64) {
        if (d->compute == HULC_F32) launch_skinny<float>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
        else launch_skinny<bf16_t>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
    } else if (d->compute == HULC_F32) launch_ct<float>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
    else launch_ct<bf16_t>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
    return hulc_check_launch("hulc_gemm");
}
