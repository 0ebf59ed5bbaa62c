// wgrad_group.hip — every small weight-gradient product of a backward pass in ONE launch:  C_i[M][N] (+)= sum_k A_i[k][m] B_i[k][n]
// (k = token; A = the layer's output gradients, B = its inputs; both row-major with the token index outer), plus the bias gradient
// rowsum_i[m] (+)= sum_k A_i[k][m].  Reference: the weight / bias gradients autograd computes for every nn.Linear of
// plan_proposal_net.py:26-47, goal_encoders.py:21-34,53-71, plan_recognition_net.py:115-148, vision_network.py:43-47 and the decoder heads
// (logistic_decoder_rnn.py:81-84).
//
// Why one launch: a training step holds ~30 of these products (outputs 32 x 128 ... 2048 x 2048, K = 32 ... 3136).  As single GEMM launches each
// fills a fraction of the chip for 5-25 us and needs a second launch to sum its split-K slabs: 60 dispatches, 0.45 ms of a 4.2 ms step.
// Here the host cuts all of them into work items (one 64 x 64 output tile over a slice of K), 4 workgroups per CU work the list off side by
// side, and the slabs of a split tile are summed — in slice order, so the result does not depend on who finishes when — by whichever
// workgroup arrives last at the tile's counter (the counter goes back to zero for the next launch).
//
// Workgroup: 4 waves, each a 32 x 32 accumulator (mfma 32x32x16 bf16).  k-step 64: both operand tiles are copied as they lie in memory,
// [k][64 columns] (fp32 sources rounded to bf16 on the way, as hulc_gemm does), and the fragments come out of ds_read_b64_tr_b16.  Two LDS
// stages: the next k-step's loads are issued before the MFMAs of the current one and written to the other stage behind them.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <algorithm>
#include <vector>

namespace {

constexpr int T = 64;                     // tile edge and k-step
constexpr int RS = T * 2 + 16;            // LDS row stride of a [k][64] bf16 tile (144 B: the 4 k rows of a transposing read on distinct banks)
constexpr int TILE_B = T * RS;            // 9216 B
constexpr int MAX_ITEMS = 34;             // per launch (kernel-argument space: 34 x 112 + 34 x 4 bytes of the 4 KB)
constexpr int SLAB = T * T + T;           // floats per partial slab (even): the tile, then its 64 bias-gradient partials
constexpr int CTR_WORDS = 65536;          // counters at the head of the workspace

struct Item {
    const void* A; const void* B; float* C; float* rowsum;
    int M, N, K, lda, ldb, ldc;
    int flags;                            // bit 0 A fp32, bit 1 B fp32, bit 2 accumulate into C, bit 3 accumulate into rowsum
    int tn, ksplit, kper;                 // tiles along N, slices of K, k-steps per slice
    int ctr0;                             // first counter
    int perm;                             // > 0: product column n is stored at column (n % perm) * (N / perm) + n / perm
    int cmul;                             // otherwise: at column n * cmul
    int mstore;                           // rows m < mstore are stored (<= M)
    int taps_wp;                          // > 0: nine products in one — tap u = (dy, dx) reads B shifted by (dy * taps_wp + dx) rows and writes C + u
    long slab0;                           // first slab
};
struct GroupP { int first[MAX_ITEMS]; Item it[MAX_ITEMS]; int n, total; unsigned* ctr; float* slabs; };   // first[j]: first work item of problem j (INT_MAX: unused)

typedef short v4s __attribute__((ext_vector_type(4)));
typedef v4s __attribute__((address_space(3))) * lds_v4s;
HULC_DEVICE v4s tr_read(const char* q) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)q); }

// 16 consecutive columns of one k row, raw
template <bool F32> struct Raw;
template <> struct Raw<true>  { float4 v[4]; };
template <> struct Raw<false> { uint4 v[2]; };

template <bool F32>
HULC_DEVICE void raw_load(Raw<F32>& r, const void* base, long ld, int k, int c0, int kmax, int cols) {
    const int kk = k < kmax ? k : kmax - 1;
    const int c = c0 + 16 <= cols ? c0 : (c0 + 8 <= cols ? c0 : 0);     // clamped: out-of-range pieces are zeroed when they are stored
    const int c8 = c0 + 16 <= cols ? c0 + 8 : c;
    if constexpr (F32) {
        const float* p = (const float*)base + (long)kk * ld;
        r.v[0] = *(const float4*)(p + c); r.v[1] = *(const float4*)(p + c + 4);
        r.v[2] = *(const float4*)(p + c8); r.v[3] = *(const float4*)(p + c8 + 4);
    } else {
        const uint16_t* p = (const uint16_t*)base + (long)kk * ld;
        r.v[0] = *(const uint4*)(p + c); r.v[1] = *(const uint4*)(p + c8);
    }
}

// -> bf16 in LDS (zero where the piece lies outside the matrix); SUM && sum: add the fp32 values to the thread's 16 column sums (selects, no branches)
template <bool F32, bool SUM>
HULC_DEVICE void raw_store(const Raw<F32>& r, char* dst, bool sum, bool kvalid, int c0, int cols, float (&rs)[16]) {
    const bool v0 = kvalid && c0 + 8 <= cols, v1 = kvalid && c0 + 16 <= cols;
    const float s0 = (sum && v0) ? 1.f : 0.f, s1 = (sum && v1) ? 1.f : 0.f;
    uint4 o0, o1;
    if constexpr (F32) {
        o0 = make_uint4(pack_bf16x2(r.v[0].x, r.v[0].y), pack_bf16x2(r.v[0].z, r.v[0].w), pack_bf16x2(r.v[1].x, r.v[1].y), pack_bf16x2(r.v[1].z, r.v[1].w));
        o1 = make_uint4(pack_bf16x2(r.v[2].x, r.v[2].y), pack_bf16x2(r.v[2].z, r.v[2].w), pack_bf16x2(r.v[3].x, r.v[3].y), pack_bf16x2(r.v[3].z, r.v[3].w));
        if constexpr (SUM) {
            rs[0] += s0 * r.v[0].x; rs[1] += s0 * r.v[0].y; rs[2] += s0 * r.v[0].z; rs[3] += s0 * r.v[0].w;
            rs[4] += s0 * r.v[1].x; rs[5] += s0 * r.v[1].y; rs[6] += s0 * r.v[1].z; rs[7] += s0 * r.v[1].w;
            rs[8] += s1 * r.v[2].x; rs[9] += s1 * r.v[2].y; rs[10] += s1 * r.v[2].z; rs[11] += s1 * r.v[2].w;
            rs[12] += s1 * r.v[3].x; rs[13] += s1 * r.v[3].y; rs[14] += s1 * r.v[3].z; rs[15] += s1 * r.v[3].w;
        }
    } else {
        o0 = r.v[0]; o1 = r.v[1];
        if constexpr (SUM) {
#define WG_SUM1(w_, sc_, j_) rs[j_] += (sc_) * __uint_as_float((w_) << 16); rs[j_ + 1] += (sc_) * __uint_as_float((w_) & 0xffff0000u);
            WG_SUM1(o0.x, s0, 0) WG_SUM1(o0.y, s0, 2) WG_SUM1(o0.z, s0, 4) WG_SUM1(o0.w, s0, 6)
            WG_SUM1(o1.x, s1, 8) WG_SUM1(o1.y, s1, 10) WG_SUM1(o1.z, s1, 12) WG_SUM1(o1.w, s1, 14)
#undef WG_SUM1
        }
    }
    const uint32_t k0 = v0 ? 0xffffffffu : 0u, k1 = v1 ? 0xffffffffu : 0u;      // (a 16-byte select goes through a stack array)
    *(uint4*)dst = make_uint4(o0.x & k0, o0.y & k0, o0.z & k0, o0.w & k0);
    *(uint4*)(dst + 16) = make_uint4(o1.x & k1, o1.y & k1, o1.z & k1, o1.w & k1);
}

template <bool AF32, bool BF32>
HULC_DEVICE void run_item(const Item& it, const GroupP& p, int local, char* smem, int* s_last) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    int u = 0;
    if (it.taps_wp > 0) { u = local % 9; local /= 9; }                    // the nine taps of a (tile, slice) are neighbours: one XCD, one L2 copy of A and B
    const int ks_i = local % it.ksplit, t = local / it.ksplit, tn_i = t % it.tn, tm_i = t / it.tn;
    const int tslab = it.taps_wp > 0 ? t * 9 + u : t;
    float* const Cb = it.C + u;
    const int m0 = tm_i * T, n0 = tn_i * T;
    const int nsteps_all = (it.K + T - 1) / T;
    const int step0 = ks_i * it.kper, step1 = min(step0 + it.kper, nsteps_all);
    const int kr = tid >> 2, c0 = (tid & 3) * 16;                         // this thread's piece of a staged tile: k row, 16 columns
    const bool do_sum = it.rowsum != nullptr && tn_i == 0;
    const int colsA = it.M - m0, colsB = it.N - n0;                        // valid columns of the two tiles (may exceed 64)
    const char* Ab = (const char*)it.A + (long)m0 * (AF32 ? 4 : 2);
    const char* Bb = (const char*)it.B + ((long)n0 + (long)((u / 3 - 1) * it.taps_wp + (u % 3 - 1)) * it.ldb * (it.taps_wp > 0 ? 1 : 0)) * (BF32 ? 4 : 2);

    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float rs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) rs[j] = 0.f;

    Raw<AF32> ra; Raw<BF32> rb;
    raw_load<AF32>(ra, Ab, it.lda, step0 * T + kr, c0, it.K, colsA);
    raw_load<BF32>(rb, Bb, it.ldb, step0 * T + kr, c0, it.K, colsB);
    {
        const bool kv = step0 * T + kr < it.K;
        raw_store<AF32, true>(ra, smem + kr * RS + c0 * 2, kv && do_sum, kv, c0, colsA, rs);
        raw_store<BF32, false>(rb, smem + TILE_B + kr * RS + c0 * 2, false, kv, c0, colsB, rs);
    }
    __syncthreads();
    const int krow = (lane >> 5) * 8 + ((lane & 15) >> 2), col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
#define WG_MMA(cur_)                                                                                                   \
    {                                                                                                                  \
        const char* At = smem + (cur_) * 2 * TILE_B + wm * 64;                                                         \
        const char* Bt = smem + (cur_) * 2 * TILE_B + TILE_B + wn * 64;                                                \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                             \
            union { v4s v[2]; bf16x8_t f; } a, b;                                                                      \
            const char* qa = At + (ks * 16 + krow) * RS + col;                                                         \
            const char* qb = Bt + (ks * 16 + krow) * RS + col;                                                         \
            a.v[0] = tr_read(qa); a.v[1] = tr_read(qa + 4 * RS);                                                       \
            b.v[0] = tr_read(qb); b.v[1] = tr_read(qb + 4 * RS);                                                       \
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, b.f, acc, 0, 0, 0);                                     \
        }                                                                                                              \
    }
    // all trips but the last: next tile's loads in front of the MFMAs, its LDS writes behind them (straight-line code around the staging
    // registers — a branch there sends them to scratch); the last trip is peeled: no load to wait for
    for (int st = step0; st + 1 < step1; ++st) {
        const int cur = (st - step0) & 1;
        raw_load<AF32>(ra, Ab, it.lda, (st + 1) * T + kr, c0, it.K, colsA);
        raw_load<BF32>(rb, Bb, it.ldb, (st + 1) * T + kr, c0, it.K, colsB);
        __builtin_amdgcn_sched_barrier(0);
        WG_MMA(cur)
        __builtin_amdgcn_sched_barrier(0);
        char* dst = smem + (cur ^ 1) * 2 * TILE_B;
        const bool kv = (st + 1) * T + kr < it.K;
        raw_store<AF32, true>(ra, dst + kr * RS + c0 * 2, kv && do_sum, kv, c0, colsA, rs);
        raw_store<BF32, false>(rb, dst + TILE_B + kr * RS + c0 * 2, false, kv, c0, colsB, rs);
        __syncthreads();
    }
    WG_MMA((step1 - 1 - step0) & 1)
#undef WG_MMA
    __syncthreads();                                                       // (the stage buffers are reused below)
    // ---- bias-gradient partial of this slice: 64 k rows x 64 columns of thread sums -> column sums (fixed order)
    float* red = (float*)smem;                                             // [64 k rows][64 columns + 1], then [4][64] quarter sums
    float bsum = 0.f;
    if (do_sum) {
#pragma unroll
        for (int j = 0; j < 16; ++j) red[kr * (T + 1) + c0 + j] = rs[j];
        __syncthreads();
        float q = 0.f;                                                     // wave w: rows 16 w .. 16 w + 15 of column `lane`
#pragma unroll
        for (int r = 0; r < 16; ++r) q += red[(wave * 16 + r) * (T + 1) + lane];
        red[T * (T + 1) + wave * T + lane] = q;
        __syncthreads();
        if (tid < T) bsum = ((red[T * (T + 1) + tid] + red[T * (T + 1) + T + tid]) + red[T * (T + 1) + 2 * T + tid]) + red[T * (T + 1) + 3 * T + tid];
    }
    const int nloc = wn * 32 + (lane & 31);
    const int nn = n0 + nloc;
    const int ncol = it.perm > 0 ? (nn % it.perm) * (it.N / it.perm) + nn / it.perm : nn * it.cmul;   // (h, w, c) -> (c, h, w) weight columns of a flattened conv map
    if (it.ksplit == 1) {
        const bool accum = it.flags & 4;
        if (n0 + nloc < it.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 32 + acc_row(e, lane);
                if (m < it.mstore) {
                    float* dst = Cb + (long)m * it.ldc + ncol;
                    *dst = accum ? *dst + acc[e] : acc[e];
                }
            }
        }
        if (do_sum && tid < T && m0 + tid < it.mstore) {
            float* dst = it.rowsum + m0 + tid;
            *dst = (it.flags & 8) ? *dst + bsum : bsum;
        }
        return;
    }
    // ---- split tile: slab out, last arrival sums all slabs in slice order.  The slabs travel as agent-scope atomic (write-through / L2-bypassing)
    // 8-byte accesses: ordinary stores + a release / acquire fence pair would write back and invalidate the whole L2 of the XCD once per
    // work item (measured: 0.9 ms for the launch instead of tens of microseconds).
    unsigned long long* slab = (unsigned long long*)(p.slabs + (it.slab0 + (long)tslab * it.ksplit + ks_i) * SLAB);
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
        const unsigned long long bits = (unsigned long long)__float_as_uint(acc[e]) | ((unsigned long long)__float_as_uint(acc[e + 1]) << 32);
        __hip_atomic_store(slab + (wave * 8 + (e >> 1)) * 64 + lane, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (do_sum && tid < T) __hip_atomic_store((unsigned*)slab + T * T + tid, __float_as_uint(bsum), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // this thread's stores are acknowledged ...
    __syncthreads();                                                       // ... and so are the workgroup's
    if (tid == 0) {
        unsigned* c = p.ctr + it.ctr0 + tslab;
        const unsigned old = __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = old == (unsigned)(it.ksplit - 1);
        if (last) __hip_atomic_store(c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // ready for the next launch
        *s_last = last;
    }
    __syncthreads();
    if (!*s_last) return;
    const unsigned long long* s0 = (const unsigned long long*)(p.slabs + (it.slab0 + (long)tslab * it.ksplit) * SLAB);
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int s = 0; s < it.ksplit; ++s) {
        const unsigned long long* sl = s0 + (long)s * (SLAB / 2);
        unsigned long long v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = __hip_atomic_load(sl + (wave * 8 + e) * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int e = 0; e < 8; ++e) { acc[2 * e] += __uint_as_float((unsigned)v[e]); acc[2 * e + 1] += __uint_as_float((unsigned)(v[e] >> 32)); }
    }
    {
        const bool accum = it.flags & 4;
        if (n0 + nloc < it.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * 32 + acc_row(e, lane);
                if (m < it.mstore) {
                    float* dst = Cb + (long)m * it.ldc + ncol;
                    *dst = accum ? *dst + acc[e] : acc[e];
                }
            }
        }
    }
    if (do_sum && tid < T && m0 + tid < it.mstore) {
        float b = 0.f;
        for (int s = 0; s < it.ksplit; ++s)
            b += __uint_as_float(__hip_atomic_load((const unsigned*)(s0 + (long)s * (SLAB / 2)) + T * T + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        float* dst = it.rowsum + m0 + tid;
        *dst = (it.flags & 8) ? *dst + b : b;
    }
}

__global__ __launch_bounds__(256) void wgrad_group_kernel(GroupP p) {
    __shared__ __attribute__((aligned(16))) char smem[4 * TILE_B];
    __shared__ int s_last;
    // workgroup b runs on XCD b % 8: runs of 8 consecutive work items (the slices of one tile, neighbouring tiles of one problem) share an
    // XCD and its L2, successive runs go round the XCDs so that every XCD sees the same mix of problems (total is a multiple of 64)
    const int g = blockIdx.x >> 3;
    const int w = (g >> 3) * 64 + (blockIdx.x & 7) * 8 + (g & 7);
    int i = 0;
#pragma unroll
    for (int j = 1; j < MAX_ITEMS; ++j) i = w >= p.first[j] ? j : i;       // one wide scalar load of the table, then compares
    const Item& it = p.it[i];
    const int local = w - p.first[i];
    if (local >= it.tn * ((it.M + T - 1) / T) * it.ksplit * (it.taps_wp > 0 ? 9 : 1)) return;        // padding of the last problem
    switch (it.flags & 3) {
    case 0: run_item<false, false>(it, p, local, smem, &s_last); break;
    case 1: run_item<true, false>(it, p, local, smem, &s_last); break;
    case 2: run_item<false, true>(it, p, local, smem, &s_last); break;
    default: run_item<true, true>(it, p, local, smem, &s_last); break;
    }
}

struct Plan { int tm, tn, ksplit, kper, tiles; };
Plan plan_item(const hulc_wgrad_item& d) {
    Plan pl;
    pl.tm = (d.M + T - 1) / T; pl.tn = (d.N + T - 1) / T; pl.tiles = pl.tm * pl.tn * (d.conv_taps_wp > 0 ? 9 : 1);      // (tile, tap) pairs
    const int nsteps = (d.K + T - 1) / T;
    // slices of 32 k-steps (2048 tokens: the per-timestep layers of a 64-sequence step are NOT split), at most 256 slabs per problem.  Round 4: with
    // slices of 4 k-steps / 1024 slabs the launch moved 594 MB, two thirds of it partial slabs written through to memory and read back (a slab is
    // 16 KB per tile and slice), at 4.6 TB/s — bound by traffic it created itself; 0.115 -> 0.086 ms (HULC_WGG_SLICE / HULC_WGG_CAP: the sweep)
    static const int slice = getenv("HULC_WGG_SLICE") ? atoi(getenv("HULC_WGG_SLICE")) : 32;
    static const int maxslabs = getenv("HULC_WGG_CAP") ? atoi(getenv("HULC_WGG_CAP")) : 256;
    int want = (nsteps + slice - 1) / slice;
    const int cap = pl.tiles >= maxslabs ? 1 : maxslabs / pl.tiles;
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    pl.kper = (nsteps + want - 1) / want;
    pl.ksplit = (nsteps + pl.kper - 1) / pl.kper;
    return pl;
}

int check_item(const hulc_wgrad_item& d) {
    if (d.M <= 0 || d.N <= 0 || d.K <= 0 || d.K % 32) return hulc_fail(-2, "hulc_wgrad_group: K must be a positive multiple of 32");
    if (d.M % 8 || d.N % 8) return hulc_fail(-2, "hulc_wgrad_group: M and N must be multiples of 8");
    const int ea = d.a_dtype == HULC_F32 ? 4 : 8, eb = d.b_dtype == HULC_F32 ? 4 : 8;
    if ((d.a_dtype != HULC_F32 && d.a_dtype != HULC_BF16) || (d.b_dtype != HULC_F32 && d.b_dtype != HULC_BF16))
        return hulc_fail(-2, "hulc_wgrad_group: operands are fp32 or bf16");
    if (d.lda % ea || d.ldb % eb || ((uintptr_t)d.A | (uintptr_t)d.B) % 16) return hulc_fail(-2, "hulc_wgrad_group: operand rows must be 16-byte aligned");
    if (d.lda < d.M || d.ldb < d.N || d.ldc < (d.col_mul > 1 ? (d.N - 1) * d.col_mul + 1 : d.N) || !d.A || !d.B || !d.C) return hulc_fail(-2, "hulc_wgrad_group: bad leading dimension or null operand");
    if (d.conv_taps_wp > 0 && (d.rowsum || d.col_mul != 9 || d.col_perm)) return hulc_fail(-2, "hulc_wgrad_group: conv_taps_wp goes with col_mul = 9, no rowsum, no col_perm");
    if (d.col_perm < 0 || (d.col_perm > 0 && d.N % d.col_perm)) return hulc_fail(-2, "hulc_wgrad_group: col_perm must divide N");
    return 0;
}

}  // namespace

// conv_taps_wp items that wgrad_taps.hip takes run there (second launch, own slab region behind this file's); the rest here
static void split_items(const hulc_wgrad_item* items, int n, std::vector<hulc_wgrad_item>& rest, std::vector<const hulc_wgrad_item*>& taps) {
    for (int i = 0; i < n; ++i) {
        if (hulc_wgrad_taps_takes(&items[i])) taps.push_back(&items[i]);
        else rest.push_back(items[i]);
    }
}
static long group_slab_bytes(const std::vector<hulc_wgrad_item>& rest) {
    long slabs = 0;
    for (const hulc_wgrad_item& d : rest) {
        const Plan pl = plan_item(d);
        if (pl.ksplit > 1) slabs += (long)pl.tiles * pl.ksplit;
    }
    return slabs * SLAB * 4;
}

extern "C" long hulc_wgrad_group_workspace(const hulc_wgrad_item* items, int n) {
    std::vector<hulc_wgrad_item> rest; std::vector<const hulc_wgrad_item*> taps;
    split_items(items, n, rest, taps);
    return (long)CTR_WORDS * 4 + group_slab_bytes(rest) + hulc_wgrad_taps_workspace(taps.data(), (int)taps.size());
}

extern "C" int hulc_wgrad_group(const hulc_wgrad_item* all_items, int n_all, void* ws, long ws_bytes, void* stream) {
    if (n_all <= 0) return 0;
    if (!all_items || !ws) return hulc_fail(-2, "hulc_wgrad_group: null argument");
    for (int i = 0; i < n_all; ++i)
        if (int rc = check_item(all_items[i])) return rc;
    std::vector<hulc_wgrad_item> rest; std::vector<const hulc_wgrad_item*> taps;
    split_items(all_items, n_all, rest, taps);
    const long own = (long)CTR_WORDS * 4 + group_slab_bytes(rest);
    if (ws_bytes < own + hulc_wgrad_taps_workspace(taps.data(), (int)taps.size())) return hulc_fail(-3, "hulc_wgrad_group: workspace too small (hulc_wgrad_group_workspace)");
    if (!taps.empty())
        if (int rc = hulc_wgrad_taps_launch(taps.data(), (int)taps.size(), (char*)ws + own, (hipStream_t)stream)) return rc;
    const hulc_wgrad_item* items = rest.data();
    const int n = (int)rest.size();
    if (n == 0) return 0;
    // the problems whose work items run longest (most k-steps per slice) go first: the launch ends with the short ones
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return plan_item(items[a]).kper > plan_item(items[b]).kper; });
    long slab = 0;
    int ctr = 0;
    for (int base = 0; base < n; base += MAX_ITEMS) {
        GroupP p;
        p.n = n - base < MAX_ITEMS ? n - base : MAX_ITEMS;
        p.ctr = (unsigned*)ws; p.slabs = (float*)((char*)ws + (long)CTR_WORDS * 4);
        int first = 0;
        for (int j = 0; j < MAX_ITEMS; ++j) p.first[j] = 0x7fffffff;
        for (int j = 0; j < p.n; ++j) {
            const hulc_wgrad_item& d = items[order[base + j]];
            const Plan pl = plan_item(d);
            Item& it = p.it[j];
            it.A = d.A; it.B = d.B; it.C = d.C; it.rowsum = d.rowsum;
            it.M = d.M; it.N = d.N; it.K = d.K; it.lda = d.lda; it.ldb = d.ldb; it.ldc = d.ldc;
            it.flags = (d.a_dtype == HULC_F32 ? 1 : 0) | (d.b_dtype == HULC_F32 ? 2 : 0) | (d.accumulate ? 4 : 0) | (d.rowsum_accumulate ? 8 : 0);
            p.first[j] = first; it.tn = pl.tn; it.ksplit = pl.ksplit; it.kper = pl.kper;
            it.ctr0 = ctr; it.slab0 = slab; it.perm = d.col_perm; it.cmul = d.col_mul > 1 ? d.col_mul : 1; it.mstore = d.store_rows > 0 && d.store_rows < d.M ? d.store_rows : d.M; it.taps_wp = d.conv_taps_wp > 0 ? d.conv_taps_wp : 0;
            first += pl.tiles * pl.ksplit;
            if (pl.ksplit > 1) { ctr += pl.tiles; slab += (long)pl.tiles * pl.ksplit; }
        }
        if (ctr > CTR_WORDS) return hulc_fail(-3, "hulc_wgrad_group: more split tiles than counters");
        p.total = (first + 63) / 64 * 64;
        wgrad_group_kernel<<<p.total, 256, 0, (hipStream_t)stream>>>(p);
    }
    return hulc_check_launch("hulc_wgrad_group");
}
