// gridconv.hip — the zero-padded 3 x 3, stride-1 convolutions of the affordance model's U-Net decoder as a shifted GEMM
// on a PADDED GRID.  Reference: Conv2dReLU / DecoderBlock of hulc2/affordance/models/core/unet_decoder.py:6-80 (nn.Conv2d(k = 3, padding = 1,
// bias = False) + BatchNorm2d + ReLU), the segmentation head of visual_lang_encoders/r3m_rn18.py:64-69, and autograd's data gradient of both.
//
// Layout: an activation map (N, H, W, C) lives as rows of C bf16 channels on the grid (N, H + 2, W + 2) — pixel (n, y, x) is grid row
// n (H + 2)(W + 2) + (y + 1)(W + 2) + (x + 1), the border rows are ZERO, and W + 3 zero guard rows precede and follow the tensor.  Then
//     Y[r][co] = sum over taps t = (dy, dx) and ci of  X[r + dy (W + 2) + dx][ci] * Wt[co][t * Cin + ci]
// for every interior row r with no bounds test at all: a tap is a constant row offset, the padding is the zero border, and the same kernel
// computes the data gradient (flipped taps, transposed weights: prepared by the host).  The weight gradient is nine products
// dW_t = dY^T X[. + off_t] over the grid rows — items of the grouped weight-gradient launch (wgrad_group.hip), the zero border of dY silencing
// the rows that are not pixels.  The price is (H + 2)(W + 2) / (H W) - 1 extra rows: 31 % at 14 x 14, 7 % at 56 x 56, 2 % at 224 x 224.
//
// Kernel: workgroup tile 128 grid rows x BN output channels (BN = 128 / 64 / 32), 4 waves, 32 x 32 x 16 bf16 MFMA, k-steps of 32 channels of
// one tap, operand tiles [row][32 k] in LDS with 80-byte rows (conflict-free ds_read_b128, as gemm.hip), register prefetch of the next k-step
// behind the MFMAs, two LDS stages.  Epilogue: bf16 store with the border rows forced to zero (the output is a grid tensor again), optional
// per-channel partial sums of y and y^2 over the interior rows from the fp32 accumulators (BatchNorm batch statistics, reduced in a fixed
// order by hulc_grid_bn_finalize), optional fp32 copy of channel 0 + bias (the one-channel segmentation head).
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

constexpr int GC_BM = 128;

struct GcP {
    const uint16_t* X; long ldx;
    const uint16_t* Wt; long ldw;
    uint16_t* Y; long ldy;
    float* out0; const float* bias;
    float* stats;
    int R, H, W, Cin, Cout;
    int flip;                       // k-block t reads the rows of tap 8 - t: the data gradient on UNflipped weights [ci][t][co]
    const float* cbias;             // optional per-channel bias (a folded BatchNorm's shift), then ...
    const uint16_t* add; long ldadd;   // ... an optional residual branch (grid tensor), then ...
    int relu;                       // ... an optional ReLU — the frozen ResNet trunk's BasicBlock (hulc_gridconv3x3_fused)
};

// one k-step of KCH x 32 channels for a wave's TM x TN accumulators; operand rows RS bytes apart (KCH = 1: 80, KCH = 2: 144 — both keep the 16
// lanes of a ds_read_b128 group on distinct banks)
template <int TM, int TN, int KCH, int RS>
HULC_DEVICE void gc_mma(const char* a_rows, const char* b_rows, f32x16_t (&acc)[TM][TN], int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 2 * KCH; ++ks) {
        bf16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8_t*)(a_rows + (i * 32 + r) * RS + (ks * 2 + h) * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8_t*)(b_rows + (j * 32 + r) * RS + (ks * 2 + h) * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <int WN, int TN, int KCH>
__global__ __launch_bounds__(256) void gridconv_kernel(GcP p) {
    constexpr int WM = 4 / WN, TM = GC_BM / (WM * 32), BN = WN * TN * 32;
    constexpr int RS = KCH * 64 + 16, CPR = KCH * 4;               // LDS row stride, 16-byte chunks per row of a k-step
    constexpr int A_PER = GC_BM * CPR / 256, B_PER = (BN * CPR + 255) / 256;      // chunks per thread
    constexpr int RPP = 256 / CPR;                                 // rows covered by one pass of the 256 threads
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [stage][A tile | B tile]
    __shared__ float sred[2][WM][BN];
    __shared__ unsigned char rowok[GC_BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r0 = blockIdx.x * GC_BM, n0 = blockIdx.y * BN;
    const int Wp = p.W + 2, PP = (p.H + 2) * Wp;
    if (tid < GC_BM) {
        const int r = r0 + tid, rem = r % PP, yy = rem / Wp, xx = rem - yy * Wp;
        rowok[tid] = (r < p.R && yy >= 1 && yy <= p.H && xx >= 1 && xx <= p.W) ? 1 : 0;
    }
    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    const int kpt = p.Cin / (32 * KCH), nk = 9 * kpt;              // k-steps per tap, in all
    // this thread's chunks: rows (tid / CPR) + q RPP, chunk tid % CPR — the same pattern for the A and the B tile
    const int ch = tid % CPR, rw = tid / CPR;
    // staging registers are NAMED values in straight-line code (arrays of them end up in scratch memory: DESIGN.md rule 6); the unused ones of a
    // configuration are compile-time dead
    const int ar0 = min(r0 + rw, p.R - 1), ar1 = min(r0 + rw + RPP, p.R - 1), ar2 = min(r0 + rw + 2 * RPP, p.R - 1), ar3 = min(r0 + rw + 3 * RPP, p.R - 1);
    const int ar4 = min(r0 + rw + 4 * RPP, p.R - 1), ar5 = min(r0 + rw + 5 * RPP, p.R - 1), ar6 = min(r0 + rw + 6 * RPP, p.R - 1), ar7 = min(r0 + rw + 7 * RPP, p.R - 1);
    uint4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3, rb4, rb5, rb6, rb7;
    ra0 = ra1 = ra2 = ra3 = ra4 = ra5 = ra6 = ra7 = rb0 = rb1 = rb2 = rb3 = rb4 = rb5 = rb6 = rb7 = make_uint4(0u, 0u, 0u, 0u);
    auto tap_off = [&](int t) { const int u = p.flip ? 8 - t : t; return (u / 3 - 1) * Wp + (u % 3 - 1); };
    constexpr bool BFULL = BN * CPR % 256 == 0;                    // every thread holds B chunks in every pass
#define GC_LA(q) if (q < A_PER) ra##q = *(const uint4*)(p.X + (long)(ar##q + off_) * p.ldx + c0_);
#define GC_LB(q) if (q < B_PER && (BFULL || rw + q * RPP < BN)) rb##q = *(const uint4*)(p.Wt + (long)(n0 + rw + q * RPP) * p.ldw + kb_);
#define GC_LOAD(ks_)                                                                                                   \
    {                                                                                                                  \
        const int t_ = (ks_) / kpt, c0_ = ((ks_) - t_ * kpt) * 32 * KCH + ch * 8, off_ = tap_off(t_);                  \
        const long kb_ = (long)t_ * p.Cin + c0_;                                                                       \
        GC_LA(0) GC_LA(1) GC_LA(2) GC_LA(3) GC_LA(4) GC_LA(5) GC_LA(6) GC_LA(7)                                        \
        GC_LB(0) GC_LB(1) GC_LB(2) GC_LB(3) GC_LB(4) GC_LB(5) GC_LB(6) GC_LB(7)                                        \
    }
#define GC_SA(q) if (q < A_PER) *(uint4*)(As_ + (rw + q * RPP) * RS + ch * 16) = ra##q;
#define GC_SB(q) if (q < B_PER && (BFULL || rw + q * RPP < BN)) *(uint4*)(Bs_ + (rw + q * RPP) * RS + ch * 16) = rb##q;
#define GC_STORE(stage_)                                                                                               \
    {                                                                                                                  \
        char* As_ = smem + (stage_) * (GC_BM + BN) * RS;                                                               \
        char* Bs_ = As_ + GC_BM * RS;                                                                                  \
        GC_SA(0) GC_SA(1) GC_SA(2) GC_SA(3) GC_SA(4) GC_SA(5) GC_SA(6) GC_SA(7)                                        \
        GC_SB(0) GC_SB(1) GC_SB(2) GC_SB(3) GC_SB(4) GC_SB(5) GC_SB(6) GC_SB(7)                                        \
    }
    GC_LOAD(0)
    GC_STORE(0)
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        const int nx = ks + 1 < nk ? ks + 1 : ks;                  // last trip reloads its own tile into the other stage: nobody reads it
        GC_LOAD(nx)
        __builtin_amdgcn_sched_barrier(0);
        const char* As = smem + cur * (GC_BM + BN) * RS;
        const char* Bs = As + GC_BM * RS;
        gc_mma<TM, TN, KCH, RS>(As + wm * TM * 32 * RS, Bs + wn * TN * 32 * RS, acc, lane);
        __builtin_amdgcn_sched_barrier(0);
        GC_STORE(cur ^ 1)
        __syncthreads();
    }
#undef GC_LOAD
#undef GC_STORE
#undef GC_LA
#undef GC_LB
#undef GC_SA
#undef GC_SB
    // ---- epilogue
    float s1[TN], s2[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j) { s1[j] = 0.f; s2[j] = 0.f; }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = (wm * TM + i) * 32 + acc_row(e, lane);
                const int r = r0 + row;
                const bool ok = rowok[row] != 0;
                float v = 0.f;
                if (ok) {
                    v = acc[i][j][e];
                    if (p.cbias) v += p.cbias[n];
                    if (p.add) v += bf16_bits_to_f32(p.add[(long)r * p.ldadd + n]);
                    if (p.relu) v = fmaxf(v, 0.f);
                }
                s1[j] += v; s2[j] += v * v;
                if (r < p.R) {
                    if (p.Y) p.Y[(long)r * p.ldy + n] = f32_to_bf16_bits(v);
                    if (p.out0 && n == 0) p.out0[r] = ok ? v + (p.bias ? p.bias[0] : 0.f) : 0.f;
                }
            }
    }
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            s1[j] += __shfl_xor(s1[j], 32, 64); s2[j] += __shfl_xor(s2[j], 32, 64);
            if (lane < 32) { sred[0][wm][(wn * TN + j) * 32 + lane] = s1[j]; sred[1][wm][(wn * TN + j) * 32 + lane] = s2[j]; }
        }
        __syncthreads();
        if (tid < BN) {
            float a = 0.f, b = 0.f;
#pragma unroll
            for (int w = 0; w < WM; ++w) { a += sred[0][w][tid]; b += sred[1][w][tid]; }
            p.stats[((long)blockIdx.x * 2) * p.Cout + n0 + tid] = a;
            p.stats[((long)blockIdx.x * 2 + 1) * p.Cout + n0 + tid] = b;
        }
    }
}

// ---- thin layers (Cin = 32 / 64: the last decoder block, the head and their data gradients — 1.6 M grid rows at 32 images) --------------------
// The kernel above stages 128 rows x 32 channels per (tap, k-step): nine L2 -> LDS passes over the input, 32-64 FLOP per staged byte, and the
// layer takes 120-210 us where its HBM traffic is 40-60.  Here a PERSISTENT workgroup keeps the whole filter of its 32 / 64 output channels in
// registers (9 taps x Cin / 16 fragments) and walks row tiles: per tile it stages three windows of 130 rows x Cin (one per dy; the dx taps are
// the same window read one row apart) — three passes instead of nine, no weight traffic per tile, one LDS fragment read per MFMA.  The output
// tile goes through LDS and leaves as 16-byte chunks (a tile of a 32-channel map is one contiguous 8 KB block).  Tiles are dealt so that
// the workgroups of an XCD work on neighbouring tiles (the dy windows of a tile are the rows of its neighbours: L2 hits).
template <int CIN, int NB, int OCC>
__global__ __launch_bounds__(256, OCC) void gridconv_thin_kernel(GcP p, int ntiles) {
    constexpr int CPR = CIN / 8, RS = CIN * 2 + 16, KS = CIN / 16;
    constexpr int WR = GC_BM + 2;                  // rows of one dy window: r0 - 1 .. r0 + 128
    constexpr int NCHK = 3 * WR * CPR;             // 16-byte chunks per tile: 3120 (Cin 64) / 1560 (Cin 32)
    constexpr int FULL = NCHK / 256, REM = NCHK % 256;     // 12 + 48 / 6 + 24
    constexpr int RPP = 256 / CPR;                 // rows per pass of the 256 threads
    constexpr int ORS = NB * 64 + 16;              // output tile rows in LDS
    constexpr int BN = NB * 32;
    static_assert(REM > 0 && FULL <= 12, "chunk plan");
    extern __shared__ __attribute__((aligned(16))) char smem[];    // [3 WR rows][RS] | output tile [128][ORS]
    char* const outs = smem + 3 * WR * RS;
    __shared__ float sred[2][4][BN];
    __shared__ unsigned char rowok[GC_BM];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int Wp = p.W + 2, PP = (p.H + 2) * Wp;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, nslots = gridDim.x >> 3;
    const int per = (ntiles + 7) / 8, tend = min((xcd + 1) * per, ntiles);
    int tile = xcd * per + slot;
    if (tile >= tend) return;
    // the filter: fragment (u, ks, nb) = k-block of the tap that reads offset u (flip: the data gradient walks the taps backwards)
    bf16x8_t wf[9][KS][NB];
#pragma unroll
    for (int u = 0; u < 9; ++u) {
        const int t = p.flip ? 8 - u : u;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
            for (int nb = 0; nb < NB; ++nb)
                wf[u][ks][nb] = *(const bf16x8_t*)(p.Wt + (long)(n0 + nb * 32 + r) * p.ldw + (long)t * p.Cin + ks * 16 + h * 8);
    }
    const int ch = tid % CPR, rw = tid / CPR;
    const int rwL = (FULL * 256 + tid % REM) / CPR, chL = (tid % REM) % CPR;       // the last, partial pass (the other threads repeat one of its chunks)
    const int Rm1 = p.R - 1;
    uint4 g0, g1, g2, g3, g4, g5, g6, g7, g8, g9, g10, g11, gL;
    g0 = g1 = g2 = g3 = g4 = g5 = g6 = g7 = g8 = g9 = g10 = g11 = make_uint4(0u, 0u, 0u, 0u);
#define GT_ROW(row_) ({ const int d_ = ((row_) >= WR) + ((row_) >= 2 * WR); (long)(min(r0_ + (row_) - d_ * WR - 1, Rm1) + (d_ - 1) * Wp); })
#define GT_L(q) if (q < FULL) g##q = *(const uint4*)(p.X + GT_ROW(rw + q * RPP) * p.ldx + ch * 8);
#define GT_LOAD(tile_)                                                                                                 \
    {                                                                                                                  \
        const int r0_ = (tile_) * GC_BM;                                                                               \
        GT_L(0) GT_L(1) GT_L(2) GT_L(3) GT_L(4) GT_L(5) GT_L(6) GT_L(7) GT_L(8) GT_L(9) GT_L(10) GT_L(11)              \
        gL = *(const uint4*)(p.X + GT_ROW(rwL) * p.ldx + chL * 8);                                                     \
    }
#define GT_S(q) if (q < FULL) *(uint4*)(smem + (rw + q * RPP) * RS + ch * 16) = g##q;
#define GT_STORE()                                                                                                     \
    {                                                                                                                  \
        GT_S(0) GT_S(1) GT_S(2) GT_S(3) GT_S(4) GT_S(5) GT_S(6) GT_S(7) GT_S(8) GT_S(9) GT_S(10) GT_S(11)              \
        *(uint4*)(smem + rwL * RS + chL * 16) = gL;                                                                    \
    }
    GT_LOAD(tile)
    for (;;) {
        const int r0 = tile * GC_BM;
        GT_STORE()
        const int next = tile + nslots;
        const bool more = next < tend;
        GT_LOAD(more ? next : tile)                                // (the last trip reloads its own tile: nobody uses it)
        if (tid < GC_BM) {
            const int rr = r0 + tid, rem = rr % PP, yy = rem / Wp, xx = rem - yy * Wp;
            rowok[tid] = (rr < p.R && yy >= 1 && yy <= p.H && xx >= 1 && xx <= p.W) ? 1 : 0;
        }
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        f32x16_t acc[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[nb][e] = 0.f;
        const char* arow = smem + (wave * 32 + r) * RS + h * 16;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int x = 0; x < 3; ++x)
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) {
                    const bf16x8_t a = *(const bf16x8_t*)(arow + (d * WR + x) * RS + ks * 32);
#pragma unroll
                    for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, wf[d * 3 + x][ks][nb], acc[nb], 0, 0, 0);
                }
        __builtin_amdgcn_sched_barrier(0);
        // ---- epilogue: border rows to zero, statistics partials, bf16 tile through LDS
        float s1[NB], s2[NB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            s1[nb] = 0.f; s2[nb] = 0.f;
            const int n = n0 + nb * 32 + r;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = wave * 32 + acc_row(e, lane);
                const bool ok = rowok[row] != 0;
                float v = 0.f;
                if (ok) {
                    v = acc[nb][e];
                    if (p.cbias) v += p.cbias[n];
                    if (p.add) v += bf16_bits_to_f32(p.add[(long)(r0 + row) * p.ldadd + n]);
                    if (p.relu) v = fmaxf(v, 0.f);
                }
                s1[nb] += v; s2[nb] += v * v;
                *(uint16_t*)(outs + row * ORS + (nb * 32 + r) * 2) = f32_to_bf16_bits(v);
                if (p.out0 && n == 0 && r0 + row < p.R) p.out0[r0 + row] = ok ? v + (p.bias ? p.bias[0] : 0.f) : 0.f;
            }
        }
        if (p.stats) {
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                s1[nb] += __shfl_xor(s1[nb], 32, 64); s2[nb] += __shfl_xor(s2[nb], 32, 64);
                if (lane < 32) { sred[0][wave][nb * 32 + lane] = s1[nb]; sred[1][wave][nb * 32 + lane] = s2[nb]; }
            }
        }
        __syncthreads();
        if (p.Y) {
#pragma unroll
            for (int k = 0; k < NB * 2; ++k) {
                const int c = tid + 256 * k, row = c / (NB * 4), cc = c % (NB * 4);
                if (r0 + row < p.R) *(uint4*)(p.Y + (long)(r0 + row) * p.ldy + n0 + cc * 8) = *(const uint4*)(outs + row * ORS + cc * 16);
            }
        }
        if (p.stats && tid < BN) {
            const float a = ((sred[0][0][tid] + sred[0][1][tid]) + sred[0][2][tid]) + sred[0][3][tid];
            const float b = ((sred[1][0][tid] + sred[1][1][tid]) + sred[1][2][tid]) + sred[1][3][tid];
            p.stats[((long)tile * 2) * p.Cout + n0 + tid] = a;
            p.stats[((long)tile * 2 + 1) * p.Cout + n0 + tid] = b;
        }
        __syncthreads();                                           // windows, rowok, outs and sred are free again
        if (!more) break;
        tile = next;
    }
#undef GT_ROW
#undef GT_L
#undef GT_LOAD
#undef GT_S
#undef GT_STORE
}

}  // namespace

// see include/hulc2_amd.h
extern "C" long hulc_gridconv_stats_bytes(int N, int H, int W, int Cout) {
    const long R = (long)N * (H + 2) * (W + 2);
    return ((R + GC_BM - 1) / GC_BM) * 2 * Cout * (long)sizeof(float);
}

static int gridconv_launch(const void* x, long ldx, const void* wt, void* y, long ldy, int N, int H, int W, int Cin, int Cout, int flip_taps,
                           float* stats, float* out0, const float* bias0, const float* cbias, const void* add, long ldadd, int relu, void* stream);

extern "C" int hulc_gridconv3x3(const void* x, long ldx, const void* wt, void* y, long ldy, int N, int H, int W, int Cin, int Cout, int flip_taps,
                                float* stats, float* out0, const float* bias0, void* stream) {
    return gridconv_launch(x, ldx, wt, y, ldy, N, H, W, Cin, Cout, flip_taps, stats, out0, bias0, nullptr, nullptr, 0, 0, stream);
}

extern "C" int hulc_gridconv3x3_fused(const void* x, long ldx, const void* wt, void* y, long ldy, int N, int H, int W, int Cin, int Cout, const float* bias,
                                      const void* add, long ldadd, int relu, void* stream) {
    if (!y) return hulc_fail(-1, "hulc_gridconv3x3_fused: null pointer");
    if (add && (ldadd < Cout)) return hulc_fail(-3, "hulc_gridconv3x3_fused: residual rows shorter than Cout");
    return gridconv_launch(x, ldx, wt, y, ldy, N, H, W, Cin, Cout, 0, nullptr, nullptr, nullptr, bias, add, ldadd, relu, stream);
}

static int gridconv_launch(const void* x, long ldx, const void* wt, void* y, long ldy, int N, int H, int W, int Cin, int Cout, int flip_taps,
                           float* stats, float* out0, const float* bias0, const float* cbias, const void* add, long ldadd, int relu, void* stream) {
    if (!x || !wt || (!y && !out0)) return hulc_fail(-1, "hulc_gridconv3x3: null pointer");
    if (N <= 0 || H <= 0 || W <= 0 || Cin % 32 || Cin <= 0 || Cout % 32 || Cout <= 0) return hulc_fail(-2, "hulc_gridconv3x3: Cin and Cout must be positive multiples of 32");
    if (ldx % 8 || ldx < Cin || (y && (ldy < Cout)) || (uintptr_t)x % 16 || (uintptr_t)wt % 16) return hulc_fail(-3, "hulc_gridconv3x3: rows must be 16-byte aligned");
    const long R = (long)N * (H + 2) * (W + 2);
    if (R >= (1L << 31) / 2) return hulc_fail(-2, "hulc_gridconv3x3: grid too large");
    GcP p;
    p.X = (const uint16_t*)x; p.ldx = ldx; p.Wt = (const uint16_t*)wt; p.ldw = 9L * Cin; p.Y = (uint16_t*)y; p.ldy = ldy;
    p.out0 = out0; p.bias = bias0; p.cbias = cbias; p.add = (const uint16_t*)add; p.ldadd = ldadd; p.relu = relu; p.stats = stats; p.R = (int)R; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout; p.flip = flip_taps ? 1 : 0;
    const unsigned gx = (unsigned)((R + GC_BM - 1) / GC_BM);
    hipStream_t s = (hipStream_t)stream;
    // thin layers: persistent workgroups with the filter in registers (gridconv_thin_kernel)
    static const int thin = getenv("HULC_GRIDCONV_THIN") ? atoi(getenv("HULC_GRIDCONV_THIN")) : 1;
    if (thin && ldy % 8 == 0 && (!y || (uintptr_t)y % 16 == 0) && ((Cin == 32 && (Cout == 32 || Cout == 64)) || (Cin == 64 && (Cout == 32 || (thin > 1 && Cout == 64))))) {
        static int ncu_cached = 0;                                  // (one device model per process: the count is asked once)
        if (!ncu_cached) {
            int dev = 0, n = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
            ncu_cached = n;
        }
        static const int occ3 = getenv("HULC_GRIDCONV_THIN_OCC3") ? atoi(getenv("HULC_GRIDCONV_THIN_OCC3")) : 1;
#define GT_LAUNCH(CINv, NBv, GYv, OCCv)                                                                                \
        {                                                                                                              \
            auto kern = gridconv_thin_kernel<CINv, NBv, OCCv>;                                                         \
            const int wgs = ((OCCv * ncu_cached + 7) / 8) * 8;     /* OCC per CU, a multiple of the 8 XCDs */          \
            const size_t lds = (size_t)3 * (GC_BM + 2) * (CINv * 2 + 16) + (size_t)GC_BM * (NBv * 64 + 16);            \
            static bool attr = false;                                                                                  \
            if (!attr) {                                                                                               \
                if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)    \
                    return hulc_fail(-8, "hulc_gridconv3x3: could not raise the dynamic LDS limit (thin)");            \
                attr = true;                                                                                           \
            }                                                                                                          \
            kern<<<dim3(wgs, GYv), 256, lds, s>>>(p, (int)gx);                                                         \
        }
        if (Cin == 32 && Cout == 32) { if (occ3) GT_LAUNCH(32, 1, 1, 3) else GT_LAUNCH(32, 1, 1, 2) }
        else if (Cin == 32) GT_LAUNCH(32, 2, 1, 2)
        else if (Cout == 32) GT_LAUNCH(64, 1, 1, 2)
        else GT_LAUNCH(64, 1, 2, 2)
#undef GT_LAUNCH
        return hulc_check_launch("hulc_gridconv3x3 (thin)");
    }
    // k-steps of 64 channels (half the barriers per MFMA) where the channel count allows and the layer is not a thin HBM-bound one
    static const int k64 = getenv("HULC_GRIDCONV_K64") ? atoi(getenv("HULC_GRIDCONV_K64")) : 1;
    const bool wide = k64 && Cin % 64 == 0 && Cin >= 128;
#define GC_LAUNCH(WNv, TNv, KCHv, BNv)                                                                                 \
    {                                                                                                                  \
        auto kern = gridconv_kernel<WNv, TNv, KCHv>;                                                                   \
        const size_t lds = (size_t)2 * (GC_BM + BNv) * (KCHv * 64 + 16);                                               \
        static bool attr = false;                                                                                      \
        if (!attr && lds > 48 * 1024) {                                                                                \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)    \
                return hulc_fail(-8, "hulc_gridconv3x3: could not raise the dynamic LDS limit");                       \
            attr = true;                                                                                               \
        }                                                                                                              \
        kern<<<dim3(gx, Cout / BNv), 256, lds, s>>>(p);                                                                \
    }
    // 128-channel k-steps for the layers that put at most one 128 x 128 workgroup on a CU (512-channel maps at 14 x 14): nothing overlaps a
    // workgroup's barriers there, so fewer, longer k-steps
    static const int k128 = getenv("HULC_GRIDCONV_K128") ? atoi(getenv("HULC_GRIDCONV_K128")) : 1;
    if (k128 && Cout % 128 == 0 && Cin % 128 == 0 && Cin >= 256 && (long)gx * (Cout / 128) <= 320) GC_LAUNCH(2, 2, 4, 128)
    else if (Cout % 128 == 0) { if (wide) GC_LAUNCH(2, 2, 2, 128) else GC_LAUNCH(2, 2, 1, 128) }
    else if (Cout % 64 == 0) { if (wide) GC_LAUNCH(1, 2, 2, 64) else GC_LAUNCH(1, 2, 1, 64) }
    else { if (wide) GC_LAUNCH(1, 1, 2, 32) else GC_LAUNCH(1, 1, 1, 32) }
#undef GC_LAUNCH
    return hulc_check_launch("hulc_gridconv3x3");
}
