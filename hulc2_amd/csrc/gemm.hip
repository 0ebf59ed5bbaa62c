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

// Staging is split in two so that a prefetch really stays in flight behind the MFMAs of the current tile: the LOAD half only issues
// global loads into registers (raw bits, addresses clamped to stay valid) and never touches a loaded value — any ALU use (bf16 -> f32
// expansion, bounds select, a bias partial sum) would make the compiler wait for the load right there, in front of the MFMA loop, and
// turn every k-tile into a full memory round trip; the FINISH half runs after the MFMAs, redoes the cheap index arithmetic, converts /
// masks, and writes the LDS tile.
//
// The operand dtypes (DT) and the staging mode are template parameters: a run-time branch around a load makes hipcc wait for it at the
// join, which is the same serialisation again.
//
// raw 8-element k-chunk (row r, k0..k0+7), held in the 8 dwords of a Chunk8 as bit patterns:
//   KMAJOR (operand stored [rows][K], k contiguous): f32 -> 8 float bit patterns; bf16 -> 4 dwords of packed pairs (v[0..3])
//   !KMAJOR (stored [K][rows]): element j's raw bits (f32 bits, or the bf16 bits zero-extended) in dword j — 8 strided loads, coalesced across lanes
template <bool KMAJOR, int DT>
HULC_DEVICE void load_operand_chunk_raw(Chunk8& c, const void* base, long ld, int rows, int K, int r, int k0) {
    constexpr int dtype = DT;
    const int kc = k0 < K ? k0 : 0;   // clamped (always valid) address; FINISH zeroes what lies past K
    r = r < rows ? r : rows - 1;      // out-of-range rows are computed but never stored
    if (KMAJOR) {
        if (dtype == HULC_F32) {
            const float4* q = (const float4*)((const float*)base + (long)r * ld + kc);
            const float4 a = q[0], b = q[1];
            c.v[0] = a.x; c.v[1] = a.y; c.v[2] = a.z; c.v[3] = a.w; c.v[4] = b.x; c.v[5] = b.y; c.v[6] = b.z; c.v[7] = b.w;
        } else {
            const uint4 u = *(const uint4*)((const uint16_t*)base + (long)r * ld + kc);
            c.v[0] = __uint_as_float(u.x); c.v[1] = __uint_as_float(u.y); c.v[2] = __uint_as_float(u.z); c.v[3] = __uint_as_float(u.w);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int kk = kc + j < K ? kc + j : 0;
            if (dtype == HULC_F32) c.v[j] = ((const float*)base)[(long)kk * ld + r];
            else c.v[j] = __uint_as_float((unsigned)((const uint16_t*)base)[(long)kk * ld + r]);
        }
    }
}

template <bool KMAJOR, int DT>
HULC_DEVICE void finish_operand_chunk(Chunk8& c, int K, int k0) {
    constexpr int dtype = DT;
    const int nvalid = k0 < K ? K - k0 : 0;               // k-major operands have K % 8 == 0: nvalid is 0 or >= 8
    if (KMAJOR) {
        if (dtype != HULC_F32) {
            const unsigned u[4] = {__float_as_uint(c.v[0]), __float_as_uint(c.v[1]), __float_as_uint(c.v[2]), __float_as_uint(c.v[3])};
#pragma unroll
            for (int j = 0; j < 4; ++j) { c.v[2 * j] = __uint_as_float(u[j] << 16); c.v[2 * j + 1] = __uint_as_float(u[j] & 0xffff0000u); }
        }
        chunk_keep_if(c, nvalid > 0);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = dtype == HULC_F32 ? c.v[j] : __uint_as_float(__float_as_uint(c.v[j]) << 16);
            c.v[j] = j < nvalid ? v : 0.f;
        }
    }
}

// bf16 tile product with both operands held in LDS as [k][rows] (the natural layout of a row-major operand): every fragment — lane = row,
// 8 consecutive k — is two ds_read_b64_tr_b16 (a 16-lane group hands in a [4 k][16 rows] block, lane i gets row i of the 4 k; see
// tools/probe/tr_probe.py).  a_tile / b_tile point at this wave's first row inside the tile, ra / rb are the k-row strides in bytes.
template <int TM, int TN>
HULC_DEVICE void mma_tile_bf16_tr(const char* a_tile, int ra, const char* b_tile, int rb, f32x16_t (&acc)[TM][TN], int lane) {
    typedef short v4s __attribute__((ext_vector_type(4)));
    typedef v4s __attribute__((address_space(3))) * lds_v4s;
    auto tr = [](const char* q) -> v4s { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)q); };
    const int krow = (lane >> 5) * 8 + ((lane & 15) >> 2), col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            union { v4s v[2]; bf16x8_t f; } x;
            const char* q = a_tile + (ks * 16 + krow) * ra + col + i * 64;
            x.v[0] = tr(q); x.v[1] = tr(q + 4 * ra);
            a[i] = x.f;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            union { v4s v[2]; bf16x8_t f; } x;
            const char* q = b_tile + (ks * 16 + krow) * rb + col + j * 64;
            x.v[0] = tr(q); x.v[1] = tr(q + 4 * rb);
            b[j] = x.f;
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

// ADT / BDT: operand storage types (HULC_F32 / HULC_BF16).  MICRO (both operands row-major, tiles fully inside the matrices, aligned:
// checked by the launcher): 8(k) x 4 / 8 x 8 micro-tile staging, see below.
template <typename CT, int TM, int TN, int WM, int WN, bool AK, bool BK, int ADT, int BDT, bool MICRO>
__global__ __launch_bounds__(WM* WN * 64) void gemm_kernel(GemmP p, float* __restrict__ slabs, int splitk, unsigned* __restrict__ tile_ctr) {
    static_assert(!MICRO || (!AK && !BK), "micro-tile staging is for row-major operands");
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

    // Row-major (reduction-major) operands — the weight-gradient GEMMs dW = dY^T X — are staged as micro-tiles (MICRO): a thread
    // loads eight 16-byte rows (k0 .. k0+7) of 4 fp32 or 8 bf16 consecutive matrix rows and transposes them in registers into
    // complete k-chunks (fp32: a renaming; bf16: a 16-bit 8x8 transpose with v_perm) instead of 32 / 64 scalar loads.
    // lane -> (k-group fastest, row group) keeps the LDS chunk writes conflict-free.  A micro-tiles go to the first threads, B micro-tiles
    // to the next; a thread's operand is chosen by SELECTS on the base / stride / row (no branch around the loads), spare threads
    // repeat the last B tile's loads and skip the store.
    constexpr int A_MR = ADT == HULC_F32 ? 4 : 8, B_MR = BDT == HULC_F32 ? 4 : 8;      // matrix rows per micro-tile
    constexpr int MTA = BM / A_MR * NCH, MTB = BN / B_MR * NCH;
    static_assert(!MICRO || MTA + MTB <= NT, "one micro-tile per thread");
    // TRT (bf16 MFMA): the row-major tiles are copied as they are — LDS holds [k][rows], the lanes of a load walk the contiguous row
    // direction (full cache lines, every thread busy), and the MFMA fragments come out of ds_read_b64_tr_b16.  The micro-tile scheme
    // (in-register transposes, half the threads idle, 64 tag lookups per load instruction) remains for the exact fp32 MFMA.
    constexpr bool TRT = MICRO && sizeof(CT) == 2;
    constexpr int RA_T = BM * 2 + 32, RB_T = BN * 2 + 32;          // [k] row strides: 4 consecutive k rows of a 32-byte column window hit distinct banks
    static_assert(!TRT || KT * (RA_T + RB_T) <= (BM + BN) * HULC_ROWB, "the [k][rows] tiles fit the double buffer");
    Chunk8 ra[(MICRO && !TRT) ? 1 : A_PER], rb[(MICRO && !TRT) ? 1 : B_PER];
    uint4 mq[(MICRO && !TRT) ? 8 : 1];
    const int nkt_all = (p.K + KT - 1) / KT;
    const int kt_per = (nkt_all + splitk - 1) / splitk;            // blockIdx.z owns k tiles [kt0, kt1)
    const int kt0 = blockIdx.z * kt_per;
    const int kt1 = kt0 + kt_per < nkt_all ? kt0 + kt_per : nkt_all;

    // chunk id -> (row, chunk-in-row).  k-major sources: consecutive threads walk k first (one row's
    // 64..128 contiguous bytes per 2..4 threads); row-major sources: consecutive threads walk rows
    // (each strided load instruction is a contiguous row segment across the wave).
    auto a_map = [&](int id, int& r, int& ch) { if (AK) { r = id / NCH; ch = id % NCH; } else { r = id % BM; ch = id / BM; } };
    auto b_map = [&](int id, int& r, int& ch) { if (BK) { r = id / NCH; ch = id % NCH; } else { r = id % BN; ch = id / BN; } };

    // micro-tile role of this thread
    const bool m_is_b = tid >= MTA;
    const int m_t = m_is_b ? (tid - MTA < MTB ? tid - MTA : MTB - 1) : tid;
    const bool m_live = tid < MTA + MTB;
    // (row-group-fastest lanes — every load instruction a few full cache lines — measured: bf16 tiles unchanged, the LDS chunk writes
    // then collide; fp32 tiles 115 -> 89 us at 2048^3, not taken because the fused row sums rely on the k-groups being adjacent lanes)
    const int m_kg = m_t % NCH, m_rg = m_t / NCH;
    const char* m_base = (const char*)(m_is_b ? p.B : p.A);
    const long m_ld = m_is_b ? p.ldb : p.lda;
    const int m_esz = (m_is_b ? BDT : ADT) == HULC_F32 ? 4 : 2;
    // edge blocks: a micro-tile that would leave the matrix is moved back inside (its rows are computed twice and stored never)
    const int m_row_want = m_is_b ? n0 + m_rg * B_MR : m0 + m_rg * A_MR;
    const int m_row_max = m_is_b ? p.N - B_MR : p.M - A_MR;
    const int m_row0 = m_row_want < m_row_max ? m_row_want : m_row_max;
    const bool m_inside = m_row_want <= m_row_max;

    // fused bias gradient: the workgroups of the first column block also sum the A rows they stage (fp32 operands: before
    // bf16 rounding)
    const bool do_rowsum = p.rowsum != nullptr && blockIdx.y == 0;
    float rs[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};

    // 16-bit 8x8 transpose: chunk of row i = {q[0][i], ..., q[7][i]}; one v_perm per output dword
    auto store_micro16 = [&](char* dst_rows, const uint4* q, int ch) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;
            uint4 o;
#define HULC_QW(j) (((const unsigned*)&q[j])[i >> 1])
            o.x = __builtin_amdgcn_perm(HULC_QW(1), HULC_QW(0), sel);
            o.y = __builtin_amdgcn_perm(HULC_QW(3), HULC_QW(2), sel);
            o.z = __builtin_amdgcn_perm(HULC_QW(5), HULC_QW(4), sel);
            o.w = __builtin_amdgcn_perm(HULC_QW(7), HULC_QW(6), sel);
#undef HULC_QW
            *(uint4*)(dst_rows + i * HULC_ROWB + ch * 16) = o;
        }
    };

    auto load_tiles = [&](int kt) {                          // LOAD half: issues loads, touches no loaded value
        if constexpr (TRT) {
#pragma unroll
            for (int q = 0; q < A_PER; ++q) {                // chunk id -> (k row, 8-row chunk): consecutive lanes walk the contiguous direction
                const int id = tid + q * NT, kr = id / (BM / 8), mc = id % (BM / 8);
                load_operand_chunk_raw<true, ADT>(ra[q], p.A, p.lda, p.K, p.M, kt * KT + kr, m0 + mc * 8);
            }
#pragma unroll
            for (int q = 0; q < B_PER; ++q) {
                const int id = tid + q * NT, kr = id / (BN / 8), nc = id % (BN / 8);
                load_operand_chunk_raw<true, BDT>(rb[q], p.B, p.ldb, p.K, p.N, kt * KT + kr, n0 + nc * 8);
            }
        } else if constexpr (MICRO) {
            const int k0 = kt * KT + m_kg * 8;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                mq[j] = *(const uint4*)(m_base + ((long)(k0 + j < p.K ? k0 + j : 0) * m_ld + m_row0) * m_esz);
        } else {
#pragma unroll
            for (int q = 0; q < A_PER; ++q) {
                int id = tid + q * NT;
                if (A_CH % NT != 0 && id >= A_CH) id = A_CH - 1;     // spare threads repeat the last chunk (no branch around the load)
                int r, ch; a_map(id, r, ch);
                load_operand_chunk_raw<AK, ADT>(ra[q], p.A, p.lda, p.M, p.K, m0 + r, kt * KT + ch * 8);
            }
#pragma unroll
            for (int q = 0; q < B_PER; ++q) {
                int id = tid + q * NT;
                if (B_CH % NT != 0 && id >= B_CH) id = B_CH - 1;
                int r, ch; b_map(id, r, ch);
                load_operand_chunk_raw<BK, BDT>(rb[q], p.B, p.ldb, p.N, p.K, n0 + r, kt * KT + ch * 8);
            }
        }
    };
    auto store_tiles = [&](int buf, int kt) {               // FINISH half for the tile loaded by load_tiles(kt)
        char* As = smem + buf * (BM + BN) * HULC_ROWB;
        char* Bs = As + BM * HULC_ROWB;
        if constexpr (TRT) {
            char* At = As; char* Bt = As + KT * RA_T;
            auto put = [&](char* dst, Chunk8& c, int dt, bool live, bool sum) {
                uint4 o;
                if (dt == HULC_F32) {
                    if (sum) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) rs[j] += c.v[j];
                    }
                    o = make_uint4(pack_bf16x2(c.v[0], c.v[1]), pack_bf16x2(c.v[2], c.v[3]), pack_bf16x2(c.v[4], c.v[5]), pack_bf16x2(c.v[6], c.v[7]));
                } else {
                    o = make_uint4(__float_as_uint(c.v[0]), __float_as_uint(c.v[1]), __float_as_uint(c.v[2]), __float_as_uint(c.v[3]));
                    if (sum) {
                        const unsigned w[4] = {o.x, o.y, o.z, o.w};
#pragma unroll
                        for (int j = 0; j < 8; ++j) rs[j] += __uint_as_float((j & 1) ? (w[j >> 1] & 0xffff0000u) : (w[j >> 1] << 16));
                    }
                }
                *(uint4*)dst = live ? o : make_uint4(0, 0, 0, 0);
            };
#pragma unroll
            for (int q = 0; q < A_PER; ++q) {
                const int id = tid + q * NT, kr = id / (BM / 8), mc = id % (BM / 8);
                const bool live = kt * KT + kr < p.K;
                put(At + kr * RA_T + mc * 16, ra[q], ADT, live, do_rowsum && live && m0 + mc * 8 < p.M);
            }
#pragma unroll
            for (int q = 0; q < B_PER; ++q) {
                const int id = tid + q * NT, kr = id / (BN / 8), nc = id % (BN / 8);
                put(Bt + kr * RB_T + nc * 16, rb[q], BDT, kt * KT + kr < p.K, false);
            }
        } else if constexpr (MICRO) {
            const int k0 = kt * KT + m_kg * 8;
            uint4 q[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) q[j] = k0 + j < p.K ? mq[j] : make_uint4(0, 0, 0, 0);
            if (!m_live) return;
            if (!m_is_b) {
                if constexpr (ADT == HULC_F32) {
                    Chunk8 c4[4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        c4[0].v[j] = __uint_as_float(q[j].x); c4[1].v[j] = __uint_as_float(q[j].y);
                        c4[2].v[j] = __uint_as_float(q[j].z); c4[3].v[j] = __uint_as_float(q[j].w);
                    }
                    if (do_rowsum && m_inside) {
#pragma unroll
                        for (int i = 0; i < 4; ++i)
#pragma unroll
                            for (int j = 0; j < 8; ++j) rs[i] += c4[i].v[j];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) chunk_store_lds<CT>(As + (m_rg * 4 + i) * HULC_ROWB + m_kg * CHB, c4[i]);
                } else {
                    if (do_rowsum && m_inside) {             // bf16 rows: the values ARE the unrounded inputs
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const unsigned w[4] = {q[j].x, q[j].y, q[j].z, q[j].w};
#pragma unroll
                            for (int i = 0; i < 8; ++i) rs[i] += __uint_as_float((i & 1) ? (w[i >> 1] & 0xffff0000u) : (w[i >> 1] << 16));
                        }
                    }
                    store_micro16(As + m_rg * 8 * HULC_ROWB, q, m_kg);
                }
            } else {
                if constexpr (BDT == HULC_F32) {
                    Chunk8 c4[4];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        c4[0].v[j] = __uint_as_float(q[j].x); c4[1].v[j] = __uint_as_float(q[j].y);
                        c4[2].v[j] = __uint_as_float(q[j].z); c4[3].v[j] = __uint_as_float(q[j].w);
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) chunk_store_lds<CT>(Bs + (m_rg * 4 + i) * HULC_ROWB + m_kg * CHB, c4[i]);
                } else store_micro16(Bs + m_rg * 8 * HULC_ROWB, q, m_kg);
            }
        } else {
#pragma unroll
            for (int q = 0; q < A_PER; ++q) {
                const int id = tid + q * NT;
                if (A_CH % NT == 0 || id < A_CH) {
                    int r, ch; a_map(id, r, ch);
                    finish_operand_chunk<AK, ADT>(ra[q], p.K, kt * KT + ch * 8);
                    if (do_rowsum && m0 + r < p.M) {
                        // row-major A: row = id % BM is the same for every q of a thread (NT % BM == 0) -> rs[0]; k-major A: chunk q
                        // belongs to row id / NCH -> rs[q] (A_PER <= 8), the NCH chunks of a row sit in adjacent lanes
#pragma unroll
                        for (int j = 0; j < 8; ++j) rs[AK ? q : 0] += ra[q].v[j];
                    }
                    chunk_store_lds<CT>(As + r * HULC_ROWB + ch * CHB, ra[q]);
                }
            }
#pragma unroll
            for (int q = 0; q < B_PER; ++q) {
                const int id = tid + q * NT;
                if (B_CH % NT == 0 || id < B_CH) {
                    int r, ch; b_map(id, r, ch);
                    finish_operand_chunk<BK, BDT>(rb[q], p.K, kt * KT + ch * 8);
                    chunk_store_lds<CT>(Bs + r * HULC_ROWB + ch * CHB, rb[q]);
                }
            }
        }
    };

    if (kt0 < kt1) {
        load_tiles(kt0);
        store_tiles(0, kt0);
    }
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (kt + 1 < kt1) load_tiles(kt + 1);
        const char* As = smem + buf * (BM + BN) * HULC_ROWB;
        const char* Bs = As + BM * HULC_ROWB;
        if constexpr (TRT) mma_tile_bf16_tr<TM, TN>(As + wm * TM * 64, RA_T, As + KT * RA_T + wn * TN * 64, RB_T, acc, lane);
        else MmaTile<CT, TM, TN>::run(As + wm * TM * 32 * HULC_ROWB, Bs + wn * TN * 32 * HULC_ROWB, acc, lane);
        if (kt + 1 < kt1) store_tiles(buf ^ 1, kt + 1);
        __syncthreads();
    }

    if (do_rowsum) {
        // partial row sums of this K slice -> rowsum slab [splitk][M] behind the C slabs (split K) or straight to the output
        float* dst = splitk > 1 ? slabs + (long)splitk * p.M * p.N + (long)blockIdx.z * p.M : p.rowsum;
        const bool acc_out = splitk == 1 && p.rowsum_accumulate;
        // split K: the partials are picked up by another workgroup of this launch (the last one to arrive at the tile): agent-scope stores
        auto put_rs = [&](float* q, float v) {
            if (splitk > 1) __hip_atomic_store(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else *q = acc_out ? *q + v : v;
        };
        if (TRT) {
            // a thread summed the 8 rows of its chunk column mc = tid % (BM / 8) over its k rows: fold the NT / (BM / 8) threads of a column
            float* red = (float*)smem;                           // operand tiles are dead
            __syncthreads();
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid * 8 + j] = rs[j];
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float v = 0.f;
                for (int t = tid / 8; t < NT; t += BM / 8) v += red[t * 8 + (tid & 7)];
                put_rs(dst + m0 + tid, v);
            }
        } else if (AK) {
            static_assert(!AK || A_PER <= 8, "row sums of a k-major A: one accumulator per chunk slot");
#pragma unroll
            for (int q = 0; q < (AK ? A_PER : 0); ++q) {
                for (int o = 1; o < NCH; o <<= 1) rs[q] += __shfl_xor(rs[q], o, 64);
                const int id = tid + q * NT, r = id / NCH;
                if ((A_CH % NT == 0 || id < A_CH) && id % NCH == 0 && m0 + r < p.M) put_rs(dst + m0 + r, rs[q]);
            }
        } else if (MICRO && ADT != HULC_F32) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
                for (int o = 1; o < NCH; o <<= 1) rs[i] += __shfl_xor(rs[i], o, 64);
            if (tid < MTA && tid % NCH == 0 && m_inside) {
                const int r0 = m0 + (tid / NCH) * 8;
#pragma unroll
                for (int i = 0; i < 8; ++i) put_rs(dst + r0 + i, rs[i]);
            }
        } else if (MICRO) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                for (int o = 1; o < NCH; o <<= 1) rs[i] += __shfl_xor(rs[i], o, 64);       // the NCH k-groups of a row group are adjacent lanes
            if (tid < MTA && tid % NCH == 0 && m_inside) {
                const int r0 = m0 + (tid / NCH) * 4;
#pragma unroll
                for (int i = 0; i < 4; ++i) put_rs(dst + r0 + i, rs[i]);
            }
        } else {
            float* red = (float*)smem;                           // operand tiles are dead: NT floats of scratch
            __syncthreads();
            red[tid] = rs[0];
            __syncthreads();
            if (tid < BM && m0 + tid < p.M) {
                float v = 0.f;
                for (int t = tid; t < NT && t < A_CH; t += BM) v += red[t];
                put_rs(dst + m0 + tid, v);
            }
        }
    }

    if (splitk > 1) {
        // Raw partial sums go to this slice's slab; the workgroup that arrives LAST at the tile's counter sums all slabs in slice order
        // (the result does not depend on who that is) and runs the epilogue below — no second launch.  The slabs travel as agent-scope
        // atomic (write-through / L2-bypassing) accesses: they are written and read by different XCDs inside one launch.
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            if (n >= p.N) continue;
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int m = m0 + (wm * TM + i) * 32 + acc_row(e, lane);
                    if (m < p.M) __hip_atomic_store(slabs + ((long)blockIdx.z * p.M + m) * p.N + n, acc[i][j][e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
        }
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this thread's partials are acknowledged ...
        __syncthreads();                                          // ... and so are the workgroup's
        if (tid == 0) {
            unsigned* c = tile_ctr + blockIdx.y * gridDim.x + blockIdx.x;
            const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(splitk - 1);
            if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);       // zero again for the next launch
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
            const int nc = n < p.N ? n : 0;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
                for (int z = 0; z < splitk; ++z) {
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int m = m0 + (wm * TM + i) * 32 + acc_row(e, lane);
                        const int mc = m < p.M ? m : 0;
                        acc[i][j][e] += __hip_atomic_load(slabs + ((long)z * p.M + mc) * p.N + nc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
        }
        if (do_rowsum && tid < BM && m0 + tid < p.M) {
            const float* rsl = slabs + (long)splitk * p.M * p.N;
            float v = 0.f;
            for (int z = 0; z < splitk; ++z) v += __hip_atomic_load(rsl + (long)z * p.M + m0 + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            p.rowsum[m0 + tid] = p.rowsum_accumulate ? p.rowsum[m0 + tid] + v : v;
        }
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
                if (p.relu == 1) v = fmaxf(v, 0.f);
                else if (p.relu == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));   // exact (erf) GELU
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
    if (p.relu == 1) v = fmaxf(v, 0.f);
    else if (p.relu == 2) v = 0.5f * v * (1.f + erff(v * 0.70710678118654752f));
    if (p.mask) v = load_elem(p.mask, p.mask_dtype, (long)m * p.ld_mask + n) > 0.f ? v * p.mask_scale : 0.f;
    if (p.drop_p > 0.f) v *= dropout_scale(p.drop_seed ^ (p.seed_dev ? p.seed_dev[0] : 0ull), (uint64_t)m * (uint64_t)p.N + n, p.drop_p);
    const long ci = (long)m * p.ldc + n;
    if (p.accumulate) v += load_elem(p.C, p.c_dtype, ci);
    return v;
}

template <typename CT, int TM, bool AK, bool BK, int NW, int ADT, int BDT>
__global__ __launch_bounds__(NW * 64) void gemm_skinny_kernel(GemmP p, float* __restrict__ slabs, int splitk, int kw, unsigned* __restrict__ tile_ctr) {
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
            for (int i = 0; i < TM; ++i) load_operand_chunk_raw<AK, ADT>(ca[u][i], p.A, p.lda, p.M, kend, i * 32 + r, k);
            load_operand_chunk_raw<BK, BDT>(cb[u], p.B, p.ldb, p.N, kend, n0 + r, k);
        }
        // every load of the batch is issued before the first value is touched (no branch around a load, no conversion in between)
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int k = k0 + u * 16 + h * 8;
#pragma unroll
            for (int i = 0; i < TM; ++i) finish_operand_chunk<AK, ADT>(ca[u][i], kend, k);
            finish_operand_chunk<BK, BDT>(cb[u], kend, k);
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
        else __hip_atomic_store(slabs + ((long)blockIdx.y * p.M + m) * p.N + n, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (splitk == 1) return;
    // split K over workgroups: the LAST workgroup to arrive at this column strip's counter sums the slabs in slice order and runs the epilogue
    // (the protocol of gemm_kernel / wgrad_group: agent-scope slab accesses, self-resetting counter) — no second launch
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* c = tile_ctr + blockIdx.x;
        const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(splitk - 1);
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = last;
    }
    __syncthreads();
    if (!s_last) return;
    for (int idx = tid; idx < TM * 1024; idx += NW * 64) {
        const int i = idx >> 10, row = (idx >> 5) & 31, col = idx & 31;
        const int m = i * 32 + row, n = n0 + col;
        if (m >= p.M || n >= p.N) continue;
        float v = 0.f;
        for (int sl = 0; sl < splitk; ++sl) v += __hip_atomic_load(slabs + ((long)sl * p.M + m) * p.N + n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        store_elem(p.C, p.c_dtype, (long)m * p.ldc + n, gemm_epilogue(p, v, m, n));
    }
}

// K slices: waves first (up to 16 per workgroup: combined through LDS in the same launch, epilogue fused),
// then workgroups (fp32 slabs + epilogue kernel) until ~256 workgroups exist; every wave keeps >= 64 k.
constexpr long HULC_GEMM_CTR_BYTES = 16384;      // 4096 tile counters at the head of hulc_gemm_desc.ws

template <typename CT>
void launch_skinny(const GemmP& p, int ak, int bk, float* ws, long ws_bytes, hipStream_t s) {
    const int colblocks = (p.N + 31) / 32;
    const bool big = p.M <= 32 && p.K >= 1024;                  // 16 waves x TM=1: 66 KB of LDS for the reduction
    const bool big2 = !big && p.K >= 1024;                       // 33..64 rows: 8 waves x TM=2, same LDS footprint
    const int nw = big ? 16 : (big2 ? 8 : 4);
    int splitk = 1;
    const char* force = getenv("HULC_SKINNY_SPLITK");           // tuning knob for A/B runs
    static const int min_kw = getenv("HULC_SKINNY_MINKW") ? atoi(getenv("HULC_SKINNY_MINKW")) : 64;   // measured (tools/rnn_bench.py): more, shorter K slices win even with the extra epilogue launch
    while (colblocks * splitk < 200 && p.K / (nw * (splitk * 2)) >= min_kw) splitk *= 2;
    if (force) splitk = atoi(force);
    unsigned* ctr = (unsigned*)ws;                               // the head of the workspace: tile counters (zero between launches)
    ws = ws ? ws + HULC_GEMM_CTR_BYTES / 4 : ws;
    ws_bytes = ws_bytes > HULC_GEMM_CTR_BYTES ? ws_bytes - HULC_GEMM_CTR_BYTES : 0;
    while (splitk > 1 && ((long)splitk * p.M * p.N * 4 > ws_bytes || (long)colblocks * 4 > HULC_GEMM_CTR_BYTES)) splitk /= 2;
    int kw = (p.K + splitk * nw - 1) / (splitk * nw);
    kw = (kw + 15) / 16 * 16;
    dim3 grid(colblocks, splitk);
    // operand layouts and storage types are template parameters (no run-time branch around a load, see load_operand_chunk_raw)
#define HULC_SK_DT(TMv, AKv, BKv, NWv, ADTv, BDTv) gemm_skinny_kernel<CT, TMv, AKv, BKv, NWv, ADTv, BDTv><<<grid, NWv * 64, 0, s>>>(p, ws, splitk, kw, ctr)
#define HULC_SK(TMv, AKv, BKv, NWv)                                                                         \
    do {                                                                                                    \
        if (sizeof(CT) == 4 || (p.a_dtype == HULC_F32 && p.b_dtype == HULC_F32)) HULC_SK_DT(TMv, AKv, BKv, NWv, HULC_F32, HULC_F32); \
        else if (p.a_dtype == HULC_F32) HULC_SK_DT(TMv, AKv, BKv, NWv, HULC_F32, HULC_BF16);               \
        else if (p.b_dtype == HULC_F32) HULC_SK_DT(TMv, AKv, BKv, NWv, HULC_BF16, HULC_F32);               \
        else HULC_SK_DT(TMv, AKv, BKv, NWv, HULC_BF16, HULC_BF16);                                          \
    } while (0)
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
#undef HULC_SK_DT
#undef HULC_SK
    // (split K: summed and finished by the last workgroup of each column strip, inside the launch)
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
    // workspace: [HULC_GEMM_CTR_BYTES of tile counters (zero between launches) | split-K slabs + row-sum slabs]
    unsigned* ctr = (unsigned*)ws;
    ws = ws ? ws + HULC_GEMM_CTR_BYTES / 4 : ws;
    ws_bytes = ws_bytes > HULC_GEMM_CTR_BYTES ? ws_bytes - HULC_GEMM_CTR_BYTES : 0;
    while (splitk > 1 && ((long)splitk * p.M * (p.N + 1) * 4 > ws_bytes || (long)gx * gy * 4 > HULC_GEMM_CTR_BYTES)) splitk /= 2;
    dim3 grid(gx, gy, splitk), block(WM * WN * 64);
    // operand layouts, storage types and the staging mode are template parameters (no run-time branch around a load)
    const int adt = sizeof(CT) == 4 ? HULC_F32 : p.a_dtype, bdt = sizeof(CT) == 4 ? HULC_F32 : p.b_dtype;
    const int amr = adt == HULC_F32 ? 4 : 8, bmr = bdt == HULC_F32 ? 4 : 8;
    const bool micro = !ak && !bk && p.lda % amr == 0 && p.ldb % bmr == 0 && (uintptr_t)p.A % 16 == 0 && (uintptr_t)p.B % 16 == 0 &&
                       p.M % amr == 0 && p.N % bmr == 0 && p.M >= amr && p.N >= bmr &&
                       (sizeof(CT) == 4 || (p.M % 8 == 0 && p.N % 8 == 0));      // bf16 MFMA: 8-row chunks of the [k][rows] tiles
#define HULC_GK(AKv, BKv, ADTv, BDTv, MICROv) gemm_kernel<CT, TM, TN, WM, WN, AKv, BKv, ADTv, BDTv, MICROv><<<grid, block, 0, s>>>(p, ws, splitk, ctr)
#define HULC_GK_DT(AKv, BKv, MICROv)                                                              \
    do {                                                                                          \
        if (adt == HULC_F32 && bdt == HULC_F32) HULC_GK(AKv, BKv, HULC_F32, HULC_F32, MICROv);    \
        else if constexpr (sizeof(CT) == 2) {                                                     \
            if (adt == HULC_F32) HULC_GK(AKv, BKv, HULC_F32, HULC_BF16, MICROv);                  \
            else if (bdt == HULC_F32) HULC_GK(AKv, BKv, HULC_BF16, HULC_F32, MICROv);             \
            else HULC_GK(AKv, BKv, HULC_BF16, HULC_BF16, MICROv);                                 \
        }                                                                                         \
    } while (0)
    if (ak && bk) HULC_GK_DT(true, true, false);
    else if (ak && !bk) HULC_GK_DT(true, false, false);
    else if (!ak && bk) HULC_GK_DT(false, true, false);
    else if (micro) HULC_GK_DT(false, false, true);
    else HULC_GK_DT(false, false, false);
#undef HULC_GK_DT
#undef HULC_GK
    // (split K: the last workgroup to arrive at a tile sums the slabs and runs the epilogue inside the same launch)
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

int hulc_gemm_tn128_try(const hulc_gemm_desc* d, hipStream_t s);   // gemm_tn128.hip
int hulc_gemm_nt128_try(const hulc_gemm_desc* d, hipStream_t s);   // gemm_nt128.hip

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
    if (p.rowsum && d->M <= 64) return hulc_fail(-6, "hulc_gemm: rowsum_a needs M > 64 (tiled path)");
    hipStream_t s = (hipStream_t)stream;
    if (!getenv("HULC_NO_GEMM_TN128")) {          // large row-major x row-major bf16 products (the recurrent weight gradients): 128-deep k-steps
        const int took = hulc_gemm_tn128_try(d, s);
        if (took < 0) return took;
        if (took) return hulc_check_launch("hulc_gemm");
    }
    {                                             // large k-major x k-major bf16 products: 128 x 128 tiles, 64-deep k-steps, ds_read_b128 operands
        const int took = hulc_gemm_nt128_try(d, s);
        if (took < 0) return took;
        if (took) return hulc_check_launch("hulc_gemm");
    }
    if (d->M <= 64) {
        if (d->compute == HULC_F32) launch_skinny<float>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
        else launch_skinny<bf16_t>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
    } else if (d->compute == HULC_F32) launch_ct<float>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
    else launch_ct<bf16_t>(p, d->a_kmajor, d->b_kmajor, (float*)d->ws, d->ws ? d->ws_bytes : 0, s);
    return hulc_check_launch("hulc_gemm");
}
