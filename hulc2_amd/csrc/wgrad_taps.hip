// wgrad_taps.hip — the weight gradient of a 3 x 3 / padding-1 convolution on the padded grid, all nine taps from ONE pass over the operands:
//     dW[co][ci][u] (+)= sum_r dZ[r][co] X[r + dy wp + dx][ci],   u = 3 (dy + 1) + (dx + 1),  wp = W + 2,  r = grid rows
// Reference: the weight gradients autograd computes for Conv2dReLU's nn.Conv2d (hulc2/affordance/models/core/unet_decoder.py:6-28) in
// UnetLangFusionDecoder (unet_decoder.py:83-146) — ten convolutions per step, 443 GFLOP at 32 images.
//
// hulc_wgrad_group ran a conv_taps_wp item as nine independent 64 x 64 tiles per k-slice: every tap staged dZ and its own shifted X again
// (32 FLOP per staged byte, 4 LDS fragment reads per MFMA) — 1.49 ms per step, 300 TFLOP/s.  Here a work unit owns a 64 (co) x 64 (ci) tile of
// ALL nine taps over a slice of rows: per 64-row k-step it stages dZ once and three 66-row windows of X (one per dy; the dx = -1 / 0 / +1
// fragments are the same window one row apart: three transposing reads + v_alignbit give all three), 143 FLOP per staged byte, and a wave
// (32 co x 32 ci x 9 taps = 144 accumulator registers) reads one dZ fragment per nine MFMAs — 11 LDS fragment reads per nine MFMAs.  Tiles with 32 output or input channels (the last decoder block) let the idle waves take
// every second / fourth 16-row sub-step instead and add their accumulators through LDS at the end.
// Split tiles write fp32 slabs; a second launch sums them in a fixed order (16 slice groups, then the groups) — no atomics, bit-reproducible.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <algorithm>
#include <stdlib.h>
#include <vector>

namespace {

constexpr int T = 64;                        // tile edge and k-step
constexpr int RS = T * 2 + 16;               // LDS row stride (144 B: the 4 k rows of a transposing read on distinct banks)
constexpr int WROWS = T + 2;                 // rows of one dy window of X: k0 - 1 .. k0 + 64
constexpr int STAGE_ROWS = T + 3 * WROWS;    // dZ tile + three windows
constexpr int STAGE_B = STAGE_ROWS * RS;     // 37,728 B; two stages
constexpr int BCH = 3 * WROWS * 8;           // 16-byte chunks of the three windows (1584 = 6 x 256 + 48)
constexpr int MAXI = 16;                     // items per launch
constexpr int SLABF = 4 * 9 * 16 * 64;       // floats per slab: [block 2 bm + bn][tap][accumulator register][lane]

struct TItem {
    const uint16_t* A; const uint16_t* B; float* C;
    int M, N, K, lda, ldb, ldc, wp;
    int tn, ksplit, kper;                    // tiles along N, slices of K, k-steps per slice
    int mode;                                // bit 0: M == 32 (waves wm split the sub-steps), bit 1: N == 32 (waves wn do)
    int accumulate;
    int mstore;                              // rows m < mstore are stored (the one-channel head padded to 32 output channels)
    long slab0;                              // first slab (split items)
};
struct TapsP { int first[MAXI]; int rfirst[MAXI]; TItem it[MAXI]; int n; float* slabs; int dbg; };

typedef short v4s __attribute__((ext_vector_type(4)));
typedef v4s __attribute__((address_space(3))) * lds_v4s;
HULC_DEVICE v4s tr_read(const char* q) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)q); }

template <int KP>      // waves per output block = 16-row sub-step phases
HULC_DEVICE void taps_unit(const TItem& it, const TapsP& p, int local, char* smem) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    const int ks_i = local % it.ksplit, t = local / it.ksplit, tn_i = t % it.tn, tm_i = t / it.tn;
    const int m0 = tm_i * T, n0 = tn_i * T;
    const bool ms = it.mode & 1, ns = it.mode & 2;
    const int bm = ms ? 0 : wm, bn = ns ? 0 : wn;
    const int kq = (ms ? wm : 0) * (ns ? 2 : 1) + (ns ? wn : 0);
    const int nsteps = (it.K + T - 1) / T;
    const int step0 = ks_i * it.kper, step1 = min(step0 + it.kper, nsteps);

    f32x16_t acc[9];
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;

    // ---- staging: 16-byte chunks, chunk q of a tile = row q / 8, columns 8 (q % 8) .. + 7; thread `tid` takes chunks tid + 256 i.
    // dZ: 512 chunks (i = 0, 1).  X: the three windows are 198 consecutive LDS rows = 1584 chunks (i = 0 .. 5, and 48 threads' worth of i = 6:
    // the other threads repeat one of those 48 — same data to the same address)
    const int ch = tid & 7, rw = tid >> 3;
    const int ca = m0 + ch * 8 < it.M ? ch * 8 : 0, cb = n0 + ch * 8 < it.N ? ch * 8 : 0;      // (columns past the matrix: never used, read in range)
    const uint16_t* Ab = it.A + m0 + ca;
    const uint16_t* Bb = it.B + n0 + cb;
    const int q6 = 6 * 256 + tid % 48, j6 = (q6 >> 3) - 2 * WROWS;
    const uint16_t* Bb6 = it.B + n0 + (n0 + (q6 & 7) * 8 < it.N ? (q6 & 7) * 8 : 0);
    const int lds6 = (T + (q6 >> 3)) * RS + (q6 & 7) * 16;
    const int ldsA = rw * RS + ch * 16, ldsB = (T + rw) * RS + ch * 16;
    const int Km1 = it.K - 1;
    uint4 ra0, ra1, rb0, rb1, rb2, rb3, rb4, rb5, rb6;
#define TP_LDB(i_, reg_)                                                                                               \
    {                                                                                                                  \
        const int q_ = tid + 256 * (i_);                                                                               \
        const int d_ = (q_ >= 8 * WROWS) + (q_ >= 16 * WROWS);                                                         \
        const int j_ = (q_ >> 3) - d_ * WROWS;                                                                         \
        reg_ = *(const uint4*)(Bb + (long)(min(k0_ + j_ - 1, Km1) + (d_ - 1) * it.wp) * it.ldb);                       \
    }
#define TP_LOAD(step_)                                                                                                 \
    {                                                                                                                  \
        const int k0_ = (step_) * T;                                                                                   \
        ra0 = *(const uint4*)(Ab + (long)min(k0_ + rw, Km1) * it.lda);                                                 \
        ra1 = *(const uint4*)(Ab + (long)min(k0_ + rw + 32, Km1) * it.lda);                                            \
        TP_LDB(0, rb0) TP_LDB(1, rb1) TP_LDB(2, rb2) TP_LDB(3, rb3) TP_LDB(4, rb4) TP_LDB(5, rb5)                      \
        rb6 = *(const uint4*)(Bb6 + (long)(min(k0_ + j6 - 1, Km1) + it.wp) * it.ldb);                                  \
    }
#define TP_STORE(stage_, step_)                                                                                        \
    {                                                                                                                  \
        char* st_ = smem + (stage_) * STAGE_B;                                                                         \
        const int k0_ = (step_) * T;                                                                                   \
        const uint32_t v0_ = k0_ + rw <= Km1 ? 0xffffffffu : 0u, v1_ = k0_ + rw + 32 <= Km1 ? 0xffffffffu : 0u;       \
        *(uint4*)(st_ + ldsA) = make_uint4(ra0.x & v0_, ra0.y & v0_, ra0.z & v0_, ra0.w & v0_);                        \
        *(uint4*)(st_ + ldsA + 32 * RS) = make_uint4(ra1.x & v1_, ra1.y & v1_, ra1.z & v1_, ra1.w & v1_);              \
        *(uint4*)(st_ + ldsB) = rb0; *(uint4*)(st_ + ldsB + 32 * RS) = rb1; *(uint4*)(st_ + ldsB + 64 * RS) = rb2;     \
        *(uint4*)(st_ + ldsB + 96 * RS) = rb3; *(uint4*)(st_ + ldsB + 128 * RS) = rb4; *(uint4*)(st_ + ldsB + 160 * RS) = rb5;   \
        *(uint4*)(st_ + lds6) = rb6;                                                                                   \
    }
    const int krow = (lane >> 5) * 8 + ((lane & 15) >> 2), col = (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
    const int fa = krow * RS + col + bm * 64, fb = (T + krow) * RS + col + bn * 64;
    // the dx = -1 / 0 / +1 fragments of a window are the same 8 rows shifted by one: three transposing reads (rows 0..3, 4..7, 8..11 of the lane
    // group) give all three — x = 1 by v_alignbit across the six dwords, x = 2 is dwords 1..4 — instead of six: 11 LDS fragment reads per nine
    // MFMAs instead of 20 (rows 10, 11 are read and dropped; for the last sub-step they lie behind the window: two rows of LDS slack)
#define TP_MMA(stage_)                                                                                                 \
    {                                                                                                                  \
        const char* s_ = smem + (stage_) * STAGE_B;                                                                    \
        _Pragma("unroll") for (int ks = 0; ks < 4 / KP; ++ks) {                                                        \
            const int kk_ = (ks * KP + kq) * 16;                                                                       \
            union { v4s v[2]; bf16x8_t f; } a;                                                                         \
            a.v[0] = tr_read(s_ + fa + kk_ * RS); a.v[1] = tr_read(s_ + fa + (kk_ + 4) * RS);                          \
            _Pragma("unroll") for (int d = 0; d < 3; ++d) {                                                            \
                const char* qb = s_ + fb + (d * WROWS + kk_) * RS;                                                     \
                union { v4s v[3]; uint32_t w[6]; } t_;                                                                 \
                t_.v[0] = tr_read(qb); t_.v[1] = tr_read(qb + 4 * RS); t_.v[2] = tr_read(qb + 8 * RS);                 \
                union { uint32_t w[4]; bf16x8_t f; } b0, b1, b2;                                                       \
                b0.w[0] = t_.w[0]; b0.w[1] = t_.w[1]; b0.w[2] = t_.w[2]; b0.w[3] = t_.w[3];                            \
                b1.w[0] = __builtin_amdgcn_alignbit(t_.w[1], t_.w[0], 16); b1.w[1] = __builtin_amdgcn_alignbit(t_.w[2], t_.w[1], 16);   \
                b1.w[2] = __builtin_amdgcn_alignbit(t_.w[3], t_.w[2], 16); b1.w[3] = __builtin_amdgcn_alignbit(t_.w[4], t_.w[3], 16);   \
                b2.w[0] = t_.w[1]; b2.w[1] = t_.w[2]; b2.w[2] = t_.w[3]; b2.w[3] = t_.w[4];                            \
                acc[d * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, b0.f, acc[d * 3 + 0], 0, 0, 0);          \
                acc[d * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, b1.f, acc[d * 3 + 1], 0, 0, 0);          \
                acc[d * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a.f, b2.f, acc[d * 3 + 2], 0, 0, 0);          \
            }                                                                                                          \
        }                                                                                                              \
    }
    TP_LOAD(step0)
    TP_STORE(0, step0)
    __syncthreads();
    for (int st = step0; st + 1 < step1; ++st) {
        const int cur = (st - step0) & 1;
        if (!(p.dbg & 1)) TP_LOAD(st + 1)
        __builtin_amdgcn_sched_barrier(0);
        if (!(p.dbg & 2)) TP_MMA(cur)
        __builtin_amdgcn_sched_barrier(0);
        TP_STORE(cur ^ 1, st + 1)
        __syncthreads();
    }
    TP_MMA((step1 - 1 - step0) & 1)
#undef TP_LDB
#undef TP_LOAD
#undef TP_STORE
#undef TP_MMA
    // ---- the sub-step phases of a block add up through LDS, phase by phase (fixed order)
    if (KP > 1) {
        float* red = (float*)smem + (ms ? bn : bm) * (KP == 2 ? 9 * 16 * 64 : 0);   // (KP = 2: two blocks, 72 KB of the stages; KP = 4: one)
        for (int r = 1; r < KP; ++r) {
            __syncthreads();
            if (kq == r) {
#pragma unroll
                for (int u = 0; u < 9; ++u)
#pragma unroll
                    for (int e = 0; e < 16; ++e) red[(u * 16 + e) * 64 + lane] = acc[u][e];
            }
            __syncthreads();
            if (kq == 0) {
#pragma unroll
                for (int u = 0; u < 9; ++u)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[u][e] += red[(u * 16 + e) * 64 + lane];
            }
        }
        if (kq != 0) return;
    }
    const int mb = m0 + bm * 32, n = n0 + bn * 32 + (lane & 31);
    if (it.ksplit == 1) {
        if (n < it.N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = mb + acc_row(e, lane);
                if (m < it.mstore) {
                    float* dst = it.C + (long)m * it.ldc + (long)n * 9;
#pragma unroll
                    for (int u = 0; u < 9; ++u) dst[u] = it.accumulate ? dst[u] + acc[u][e] : acc[u][e];
                }
            }
        }
        return;
    }
    float* slab = p.slabs + (it.slab0 + (long)t * it.ksplit + ks_i) * SLABF + (bm * 2 + bn) * (9 * 16 * 64);
#pragma unroll
    for (int u = 0; u < 9; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) slab[(u * 16 + e) * 64 + lane] = acc[u][e];
}

template <int OCC>     // workgroups per CU the register budget allows: 2 -> at most 256 registers per lane
__global__ __launch_bounds__(256, OCC) void wgrad_taps_kernel(TapsP p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // runs of 8 consecutive units (slices of one tile: neighbouring rows, overlapping dy windows) share an XCD and its L2
    const int g = blockIdx.x >> 3;
    const int w = (g >> 3) * 64 + (blockIdx.x & 7) * 8 + (g & 7);
    int i = 0;
#pragma unroll
    for (int j = 1; j < MAXI; ++j) i = w >= p.first[j] ? j : i;
    const TItem& it = p.it[i];
    const int local = w - p.first[i];
    if (local >= it.tn * ((it.M + T - 1) / T) * it.ksplit) return;
    switch (it.mode) {
    case 0: taps_unit<1>(it, p, local, smem); break;
    case 3: taps_unit<4>(it, p, local, smem); break;
    default: taps_unit<2>(it, p, local, smem); break;
    }
}

// one workgroup per (split tile, block, tap): 16 waves take the slices s = wave, wave + 16, ... of the 1024 values, then 16 partial sums each
__global__ __launch_bounds__(1024) void wgrad_taps_reduce_kernel(TapsP p) {
    __shared__ float part[16 * 16 * 64];
    int i = 0;
#pragma unroll
    for (int j = 1; j < MAXI; ++j) i = (int)blockIdx.x >= p.rfirst[j] ? j : i;
    const TItem& it = p.it[i];
    const int local = blockIdx.x - p.rfirst[i];
    const int u = local % 9, b = (local / 9) & 3, t = local / 36;
    const int tn_i = t % it.tn, tm_i = t / it.tn;
    const int mb = tm_i * T + (b >> 1) * 32, nb = tn_i * T + (b & 1) * 32;
    if (mb >= it.M || nb >= it.N) return;
    const int tid = threadIdx.x, lane = tid & 63, sg = tid >> 6;
    const float* s0 = p.slabs + (it.slab0 + (long)t * it.ksplit) * SLABF + ((b * 9 + u) * 16) * 64 + lane;
    float a[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    for (int s = sg; s < it.ksplit; s += 16) {
        const float* sl = s0 + (long)s * SLABF;
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = sl[e * 64];
#pragma unroll
        for (int e = 0; e < 16; ++e) a[e] += v[e];
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) part[(sg * 16 + e) * 64 + lane] = a[e];
    __syncthreads();
    const int e = sg;
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += part[(g * 16 + e) * 64 + lane];
    const int m = mb + acc_row(e, lane), n = nb + (lane & 31);
    if (m >= it.mstore) return;
    float* dst = it.C + (long)m * it.ldc + (long)n * 9 + u;
    *dst = it.accumulate ? *dst + v : v;
}

struct TPlan { int tm, tn, tiles, mode, kp, ksplit, kper; float cost; };
TPlan plan_taps(const hulc_wgrad_item& d) {
    TPlan pl;
    pl.tm = (d.M + T - 1) / T; pl.tn = (d.N + T - 1) / T; pl.tiles = pl.tm * pl.tn;
    pl.mode = (d.M == 32 ? 1 : 0) | (d.N == 32 ? 2 : 0);
    pl.kp = pl.mode == 0 ? 1 : (pl.mode == 3 ? 4 : 2);
    const int nsteps = (d.K + T - 1) / T;
    static const int per = getenv("HULC_WGRAD_TAPS_KPER") ? atoi(getenv("HULC_WGRAD_TAPS_KPER")) : 64;
    int want = (nsteps + per * pl.kp - 1) / (per * pl.kp);                 // ~64 k-steps of nine-tap MFMAs per wave and unit
    if (nsteps <= 160 && pl.tiles >= 32) want = 1;                         // many tiles, few rows: no slabs at all
    if (want > 128) want = 128;
    if (want < 1) want = 1;
    pl.kper = (nsteps + want - 1) / want;
    pl.ksplit = (nsteps + pl.kper - 1) / pl.kper;
    pl.cost = (float)pl.kper * (pl.kp == 1 ? 1.f : pl.kp == 2 ? 0.6f : 0.4f);
    return pl;
}

}  // namespace

// ---- internal interface of hulc_wgrad_group (wgrad_group.hip): the conv_taps_wp items this file takes
int hulc_wgrad_taps_takes(const hulc_wgrad_item* d) {
    static const bool off = getenv("HULC_NO_WGRAD_TAPS") != nullptr;
    if (off || d->conv_taps_wp <= 0) return 0;
    if (d->a_dtype != HULC_BF16 || d->b_dtype != HULC_BF16 || d->rowsum || d->col_perm || d->col_mul != 9) return 0;
    if (!((d->M % T == 0) || d->M == 32) || !((d->N % T == 0) || d->N == 32) || d->K % 32 || d->K < T) return 0;
    if (d->lda % 8 || d->ldb % 8 || ((uintptr_t)d->A | (uintptr_t)d->B) % 16) return 0;
    return 1;
}

long hulc_wgrad_taps_workspace(const hulc_wgrad_item* const* items, int n) {
    long slabs = 0;
    for (int i = 0; i < n; ++i) {
        const TPlan pl = plan_taps(*items[i]);
        if (pl.ksplit > 1) slabs += (long)pl.tiles * pl.ksplit;
    }
    return slabs * SLABF * 4;
}

int hulc_wgrad_taps_launch(const hulc_wgrad_item* const* items, int n, void* slabs, hipStream_t s) {
    static const int occ = getenv("HULC_WGRAD_TAPS_OCC") ? atoi(getenv("HULC_WGRAD_TAPS_OCC")) : 2;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)wgrad_taps_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B + 2 * RS) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_taps_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE_B + 2 * RS) != hipSuccess)
            return hulc_fail(-8, "hulc_wgrad_group: could not raise the dynamic LDS limit (taps)");
        attr = true;
    }
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return plan_taps(*items[a]).cost > plan_taps(*items[b]).cost; });   // long units first
    long slab = 0;
    for (int base = 0; base < n; base += MAXI) {
        TapsP p;
        p.n = n - base < MAXI ? n - base : MAXI;
        p.slabs = (float*)slabs;
        p.dbg = getenv("HULC_TAPS_DBG") ? atoi(getenv("HULC_TAPS_DBG")) : 0;      // timing probes: 1 no global loads, 2 no MFMAs (results are wrong)
        int first = 0, rfirst = 0;
        for (int j = 0; j < MAXI; ++j) p.first[j] = p.rfirst[j] = 0x7fffffff;
        for (int j = 0; j < p.n; ++j) {
            const hulc_wgrad_item& d = *items[order[base + j]];
            const TPlan pl = plan_taps(d);
            TItem& it = p.it[j];
            it.A = (const uint16_t*)d.A; it.B = (const uint16_t*)d.B; it.C = d.C;
            it.M = d.M; it.N = d.N; it.K = d.K; it.lda = d.lda; it.ldb = d.ldb; it.ldc = d.ldc; it.wp = d.conv_taps_wp;
            it.tn = pl.tn; it.ksplit = pl.ksplit; it.kper = pl.kper; it.mode = pl.mode; it.accumulate = d.accumulate; it.slab0 = slab;
            it.mstore = d.store_rows > 0 && d.store_rows < d.M ? d.store_rows : d.M;
            p.first[j] = first; first += pl.tiles * pl.ksplit;
            if (pl.ksplit > 1) { p.rfirst[j] = rfirst; rfirst += pl.tiles * 36; slab += (long)pl.tiles * pl.ksplit; }
        }
        // the reduce table must be monotone for the lookup: unsplit items keep INT_MAX only behind the split ones of larger index — give them
        // the running value instead (zero workgroups of their own)
        {
            int run = rfirst;
            for (int j = p.n - 1; j >= 0; --j) { if (p.rfirst[j] == 0x7fffffff) p.rfirst[j] = run; else run = p.rfirst[j]; }
        }
        const int total = (first + 63) / 64 * 64;
        if (occ == 1) wgrad_taps_kernel<1><<<total, 256, 2 * STAGE_B + 2 * RS, s>>>(p);
        else wgrad_taps_kernel<2><<<total, 256, 2 * STAGE_B + 2 * RS, s>>>(p);
        if (rfirst > 0) wgrad_taps_reduce_kernel<<<rfirst, 1024, 0, s>>>(p);
    }
    return hulc_check_launch("hulc_wgrad_group (taps)");
}
