// conv.hip — the un-padded conv stacks of both camera encoders as MFMA implicit GEMMs.
//
// reference: hulc2/models/perceptual_encoders/vision_network.py:36-47 (static 200x200) and
//            vision_network_gripper.py:11-20 (gripper 84x84): 8x8 s4 (3->32), 4x4 s2 (32->64),
//            3x3 s1 (64->64), ReLU after each; backward = what autograd derives for nn.Conv2d.
//
// Data layout in HBM: conv1 reads the batch as the reference delivers it (NCHW fp32, k = (c,kh,kw));
// every intermediate activation is NHWC so an im2col row is made of long contiguous channel runs and
// an 8-element k-chunk (the MFMA operand unit) is one aligned 16/32-byte load.
//
//   forward / data-gradient : gather kernel.  M = output pixels, N = output channels, K = taps x inner.
//       A[m][k] is gathered on the fly: address = pixel_base(m) + tap_off[k / inner] + k % inner with a
//       bounds check per tap (zero fill = padding, needed by the data gradient only).
//       The data gradient of a stride-s conv is run as s*s parity classes, each a dense stride-1
//       correlation of dY with the matching weight taps (no multiplications by structural zeros).
//   weight-gradient         : reduction over pixels.  dW[co][k] = sum_m dY[m][co] * A[m][k]; both
//       operands have the reduction index outermost in memory, so they are transposed while staged
//       (two pixels per thread -> one packed LDS write per k element).  Pixel range split over
//       workgroups, fp32 partial slabs summed by a second deterministic pass; the bias gradient
//       (column sums of dY) rides along.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

// LDS-band kernel (conv_band.hip): 0 = launched, 1 = geometry not covered (use the gather kernel), < 0 = error
int hulc_conv_band_dispatch(int C, int NSET, int TH, int TW, int S, const void* x, int x_dtype, int N, int H, int W, int pad_y, int pad_x,
                            long x_sn, long x_sy, long x_sx, void* y, int y_dtype, long y_sn, long y_sy, long y_sx, const void* wt,
                            int w_dtype, long ldw, const float* bias, const void* mask, int mask_dtype, int relu, int ncls,
                            const int* cls_OH, const int* cls_OW, const long* cls_yoff, const int* cls_cobase, const long* cls_wrow0,
                            const long* cls_wtap, const void* add, unsigned* bits_out, const unsigned* bits_in, int bits_channels, void* y16,
                            hipStream_t s);
// LDS-band conv1 forward (conv1_band.hip): NCHW fp32 frames, 3 -> 32 channels, 8x8 stride 4; same return convention
int hulc_conv1_band_dispatch(const float* x, const void* w, int w_dtype, long ldw, const float* bias, void* y, int y_dtype, int relu,
                             int N, int H, int W, int u8, int pad, const int* shift, const int* fidx, unsigned* relu_bits, const void* w_lo,
                             const void* x2, int n_split, const void* x_slot, const void* x2_slot, hipStream_t s);
// LDS-band weight gradient (conv_wgrad_band.hip): same return convention
int hulc_conv_wgrad_band_dispatch(int nchw, int Cin, int Cout, int KH, int KW, int S, const void* x, int x_dtype, const void* dy, int dy_dtype,
                                  int N, int H, int W, float* dw, float* db, void* ws, long ws_bytes, int dw_oihw, int accumulate, int u8, int pad,
                                  const int* shift, const int* fidx, const void* x2, int n_split, const void* x_slot, const void* x2_slot, hipStream_t s);

namespace {

#define HULC_MAX_TAPS 64

struct GatherP {
    const void* X; void* Y; const void* Wt; const float* bias; const void* mask; const void* add;
    int x_dtype, y_dtype, w_dtype, mask_dtype, add_dtype;
    int Nimg, OH, OW, Cout, H, W;
    long x_sn, x_sy, x_sx;
    long y_sn, y_sy, y_sx;
    long ldw;
    int stride, ntaps, inner_log2, check_bounds;
    int stride_x;                // output-x step in input positions (= stride except for the packed ResNet stem)
    int grid_kw, grid_pad;       // > 0: the taps are a dense KH x KW grid, tap t = (t / grid_kw - grid_pad, t % grid_kw - grid_pad), weights k-contiguous
    int relu; float mask_scale;
    int tap_dy[HULC_MAX_TAPS], tap_dx[HULC_MAX_TAPS];
    long tap_off[HULC_MAX_TAPS], w_tap_off[HULC_MAX_TAPS];
};

// one gathered 8-element chunk of the implicit im2col row of a pixel
HULC_DEVICE void gather_chunk(Chunk8& c, const GatherP& p, long pix_base, int iy0, int ix0, int k0, int K) {
    bool keep = k0 < K;
    const int kc = keep ? k0 : 0;
    const int t = kc >> p.inner_log2, j = kc & ((1 << p.inner_log2) - 1);
    if (p.check_bounds) {
        const int iy = iy0 + p.tap_dy[t], ix = ix0 + p.tap_dx[t];
        keep = keep && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
    }
    // unconditional load from a clamped address + select (no branch around the load: see chunk_keep_if)
    const long off = keep ? pix_base + p.tap_off[t] + j : 0;
    chunk_load_contig(c, p.X, p.x_dtype, off);
    chunk_keep_if(c, keep);
}

template <typename CT, int TM, int TN, int WM, int WN>
__global__ __launch_bounds__(WM* WN * 64) void conv_gather_kernel(GatherP p) {
    using T = MmaTraits<CT>;
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int KT = T::KT, NCH = T::NCH, CHB = T::CHB;
    constexpr int A_CH = BM * NCH, B_CH = BN * NCH;
    constexpr int A_PER = (A_CH + NT - 1) / NT, B_PER = (B_CH + NT - 1) / NT;
    static_assert(A_CH % NT == 0, "A tile must divide evenly");

    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * HULC_ROWB];
    __shared__ long out_off[BM];      // output element offset of each tile row, -1 = out of range

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long Mtot = (long)p.Nimg * p.OH * p.OW;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int K = p.ntaps << p.inner_log2;
    const int nkt = (K + KT - 1) / KT;

    if (tid < BM) {
        long m = m0 + tid;
        if (m < Mtot) {
            int ox = (int)(m % p.OW); long r = m / p.OW; int oy = (int)(r % p.OH); long n = r / p.OH;
            out_off[tid] = n * p.y_sn + (long)oy * p.y_sy + (long)ox * p.y_sx;
        } else out_off[tid] = -1;
    }

    // per-thread gather coordinates of the A rows this thread stages (fixed over the k loop)
    long pbase[A_PER]; int iy0[A_PER], ix0[A_PER];
#pragma unroll
    for (int q = 0; q < A_PER; ++q) {
        int r = (tid + q * NT) / NCH;
        long m = m0 + r; if (m >= Mtot) m = Mtot - 1;
        int ox = (int)(m % p.OW); long rr = m / p.OW; int oy = (int)(rr % p.OH); long n = rr / p.OH;
        iy0[q] = oy * p.stride; ix0[q] = ox * p.stride_x;
        pbase[q] = n * p.x_sn + (long)iy0[q] * p.x_sy + (long)ix0[q] * p.x_sx;
    }

    f32x16_t acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    Chunk8 ra[A_PER], rb[B_PER];
    auto load_tiles = [&](int kt) {
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            int ch = (tid + q * NT) % NCH;
            gather_chunk(ra[q], p, pbase[q], iy0[q], ix0[q], kt * KT + ch * 8, K);
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            int id = tid + q * NT;
            if (B_CH % NT == 0 || id < B_CH) {
                int r = id / NCH, ch = id % NCH, k0 = kt * KT + ch * 8;
                int n = n0 + r; n = n < p.Cout ? n : p.Cout - 1;
                const bool keep = k0 < K;
                const int kc = keep ? k0 : 0;
                const int t = kc >> p.inner_log2, j = kc & ((1 << p.inner_log2) - 1);
                chunk_load_contig(rb[q], p.Wt, p.w_dtype, (long)n * p.ldw + p.w_tap_off[t] + j);
                chunk_keep_if(rb[q], keep);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        char* As = smem + buf * (BM + BN) * HULC_ROWB;
        char* Bs = As + BM * HULC_ROWB;
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            int id = tid + q * NT;
            chunk_store_lds<CT>(As + (id / NCH) * HULC_ROWB + (id % NCH) * CHB, ra[q]);
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) {
            int id = tid + q * NT;
            if (B_CH % NT == 0 || id < B_CH) chunk_store_lds<CT>(Bs + (id / NCH) * HULC_ROWB + (id % NCH) * CHB, rb[q]);
        }
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles(kt + 1);
        const char* As = smem + buf * (BM + BN) * HULC_ROWB;
        const char* Bs = As + BM * HULC_ROWB;
        MmaTile<CT, TM, TN>::run(As + wm * TM * 32 * HULC_ROWB, Bs + wn * TN * 32 * HULC_ROWB, acc, lane);
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }

#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + (lane & 31);
        if (n >= p.Cout) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long off = out_off[(wm * TM + i) * 32 + acc_row(e, lane)];
                if (off < 0) continue;
                float v = acc[i][j][e] + bv;
                if (p.add) v += load_elem(p.add, p.add_dtype, off + n);       // residual branch of a ResNet block
                if (p.relu) v = fmaxf(v, 0.f);
                if (p.mask) v = load_elem(p.mask, p.mask_dtype, off + n) > 0.f ? v * p.mask_scale : 0.f;
                store_elem(p.Y, p.y_dtype, off + n, v);
            }
    }
}

// bf16-only variant for taps of >= 32 contiguous elements (every NHWC layer with Cin >= 32): the whole 32-wide k tile lies inside ONE
// tap, so the tap's offsets are wave-uniform scalar loads, and the gathered 16-byte pieces stay untouched in registers between the global
// load and the LDS write (zero padding is a select at store time) — the loads of tile kt+1 really are in flight during the MFMAs of
// tile kt.  The generic kernel above converts every chunk to fp32 as it arrives, which puts the memory wait in front of the MFMAs.
// UNI = false: taps narrower than the k tile (the stem: 8 channels per tap, 4 taps per tile) — the tap of a thread's piece is computed from the
// dense KH x KW grid (p.grid_kw / p.grid_pad) instead of being looked up, and K may end inside a tile.
// KT: k-tile width.  Measured on the ResNet trunk's 128..512-channel layers (450-590 TFLOP/s with KT = 32, 64 x 64 per wave): KT = 64
// is 7 % slower (73 KB of LDS per block), 128 x 64 per wave 35 % slower — 32 / <2,2,2,2> stays.
template <int TA, int TB, int ROWB, int KS>
HULC_DEVICE void mma_rows_bf16(const char* a_rows, const char* b_rows, f32x16_t (&acc)[TA][TB], int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        bf16x8_t a[TA], b[TB];
#pragma unroll
        for (int i = 0; i < TA; ++i) a[i] = *(const bf16x8_t*)(a_rows + (i * 32 + r) * ROWB + (ks * 2 + h) * 16);
#pragma unroll
        for (int j = 0; j < TB; ++j) b[j] = *(const bf16x8_t*)(b_rows + (j * 32 + r) * ROWB + (ks * 2 + h) * 16);
#pragma unroll
        for (int i = 0; i < TA; ++i)
#pragma unroll
            for (int j = 0; j < TB; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <int TM, int TN, int WM, int WN, bool UNI, int KT = 32>
__global__ __launch_bounds__(WM* WN * 64) void conv_gather_bf16_kernel(GatherP p) {
    constexpr int NT = WM * WN * 64;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int NCH = KT / 8, ROWB = KT * 2 + 16;
    constexpr int A_PER = BM * NCH / NT, B_PER = BN * NCH / NT;
    static_assert((BM * NCH) % NT == 0 && (BN * NCH) % NT == 0 && NT % NCH == 0, "tiles must divide evenly");

    __shared__ __attribute__((aligned(16))) char smem[2 * (BM + BN) * ROWB];
    __shared__ long out_off[BM];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const long Mtot = (long)p.Nimg * p.OH * p.OW;
    const long m0 = (long)blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int K = p.ntaps << p.inner_log2;
    const int nkt = (K + KT - 1) / KT;
    const int ch = tid % NCH, row0 = tid / NCH;                 // this thread's 16-byte piece of a tile row, rows row0 + q * NT / NCH

    if (tid < BM) {
        long m = m0 + tid;
        if (m < Mtot) {
            int ox = (int)(m % p.OW); long r = m / p.OW; int oy = (int)(r % p.OH); long n = r / p.OH;
            out_off[tid] = n * p.y_sn + (long)oy * p.y_sy + (long)ox * p.y_sx;
        } else out_off[tid] = -1;
    }

    long pbase[A_PER]; int iy0[A_PER], ix0[A_PER];
#pragma unroll
    for (int q = 0; q < A_PER; ++q) {
        long m = m0 + row0 + q * (NT / NCH); if (m >= Mtot) m = Mtot - 1;
        int ox = (int)(m % p.OW); long rr = m / p.OW; int oy = (int)(rr % p.OH); long n = rr / p.OH;
        iy0[q] = oy * p.stride; ix0[q] = ox * p.stride_x;
        pbase[q] = n * p.x_sn + (long)iy0[q] * p.x_sy + (long)ix0[q] * p.x_sx;
    }
    long wbase[B_PER];
#pragma unroll
    for (int q = 0; q < B_PER; ++q) {
        int n = n0 + row0 + q * (NT / NCH); n = n < p.Cout ? n : p.Cout - 1;
        wbase[q] = (long)n * p.ldw;
    }

    // MFMA roles swapped (A = weights, B = pixels): D[channel][pixel] leaves a lane with ONE pixel and groups of four consecutive
    // channels, so the epilogue is 8-byte (bf16) / 16-byte (fp32) accesses with one output offset per lane
    f32x16_t acc[TN][TM];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[j][i][e] = 0.f;

    uint4 ra[A_PER], rb[B_PER];
    unsigned keep = 0;
    bool kin = true;                                                 // UNI = false: this thread's piece of the tile is below K
    const int inner_mask = (1 << p.inner_log2) - 1;
    auto load_tiles = [&](int kt) {
        long toff, woff; int dy, dx;
        if (UNI) {
            const int k0 = kt * KT;
            const int t = k0 >> p.inner_log2, j0 = (k0 & inner_mask) + ch * 8;   // t is wave-uniform: scalar loads of the tap's offsets
            toff = p.tap_off[t] + j0; woff = p.w_tap_off[t] + j0;
            dy = p.tap_dy[t]; dx = p.tap_dx[t];
        } else {
            const int kc = kt * KT + ch * 8;
            kin = kc < K;
            const int kk = kin ? kc : 0;
            const int t = kk >> p.inner_log2, kh = t / p.grid_kw;
            dy = kh - p.grid_pad; dx = t - kh * p.grid_kw - p.grid_pad;
            toff = (long)dy * p.x_sy + (long)dx * p.x_sx + (kk & inner_mask); woff = kk;
        }
        keep = 0;
#pragma unroll
        for (int q = 0; q < A_PER; ++q) {
            const int iy = iy0[q] + dy, ix = ix0[q] + dx;
            const bool in = kin && (!p.check_bounds || (iy >= 0 && iy < p.H && ix >= 0 && ix < p.W));
            ra[q] = *(const uint4*)((const uint16_t*)p.X + (in ? pbase[q] + toff : 0));
            keep |= (in ? 1u : 0u) << q;
        }
#pragma unroll
        for (int q = 0; q < B_PER; ++q) rb[q] = *(const uint4*)((const uint16_t*)p.Wt + wbase[q] + woff);
    };
    auto store_tiles = [&](int buf) {
        char* As = smem + buf * (BM + BN) * ROWB;
        char* Bs = As + BM * ROWB;
#pragma unroll
        for (int q = 0; q < A_PER; ++q)
            *(uint4*)(As + (row0 + q * (NT / NCH)) * ROWB + ch * 16) = ((keep >> q) & 1u) ? ra[q] : make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < B_PER; ++q) *(uint4*)(Bs + (row0 + q * (NT / NCH)) * ROWB + ch * 16) = kin ? rb[q] : make_uint4(0, 0, 0, 0);
    };

    load_tiles(0);
    store_tiles(0);
    __syncthreads();
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nkt) load_tiles(kt + 1);
        const char* As = smem + buf * (BM + BN) * ROWB;
        const char* Bs = As + BM * ROWB;
        mma_rows_bf16<TN, TM, ROWB, KT / 16>(Bs + wn * TN * 32 * ROWB, As + wm * TM * 32 * ROWB, acc, lane);
        if (kt + 1 < nkt) store_tiles(buf ^ 1);
        __syncthreads();
    }

    const int hh = lane >> 5;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const long off = out_off[(wm * TM + i) * 32 + (lane & 31)];
        if (off < 0) continue;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int nb = n0 + (wn * TN + j) * 32;              // Cout % 32 == 0: a 32-channel block is inside or outside as a whole
            if (nb >= p.Cout) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = nb + 8 * g + 4 * hh;               // registers 4g..4g+3 = channels c..c+3
                float v[4] = {acc[j][i][4 * g], acc[j][i][4 * g + 1], acc[j][i][4 * g + 2], acc[j][i][4 * g + 3]};
                if (p.bias) { const float4 b = *(const float4*)(p.bias + c); v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w; }
                if (p.add) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] += load_elem(p.add, p.add_dtype, off + c + e);
                }
                if (p.relu) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                }
                if (p.mask) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = load_elem(p.mask, p.mask_dtype, off + c + e) > 0.f ? v[e] * p.mask_scale : 0.f;
                }
                if (p.y_dtype == HULC_BF16) *(uint2*)((uint16_t*)p.Y + off + c) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                else *(float4*)((float*)p.Y + off + c) = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
    }
}

template <typename CT> bool launch_gather_raw(const GatherP&, hipStream_t) { return false; }
template <> bool launch_gather_raw<bf16_t>(const GatherP& p, hipStream_t s) {
    if (p.x_dtype != HULC_BF16 || p.w_dtype != HULC_BF16 || p.Cout <= 32 || getenv("HULC_GATHER_GENERIC")) return false;
    const long Mtot = (long)p.Nimg * p.OH * p.OW;
    const bool wide = p.Cout % 128 == 0;
    // every wave owns 64 pixels x 64 channels (8 MFMAs per 32-wide k tile): 128 x 128 tiles for wide layers, 256 pixels x 64 channels else
    const bool tall = !wide && p.Cout % 64 == 0 && Mtot >= 256 * 512 && !getenv("HULC_GATHER_M128");
    dim3 grid((unsigned)((Mtot + (tall ? 255 : 127)) / (tall ? 256 : 128)), wide ? p.Cout / 128 : (p.Cout + 63) / 64);
    // few 128 x 128 tiles (the trunk's 128 / 256 / 512-channel stages at 32 images: 196, 98 and 52 workgroups on 256 CUs; measured 5.02 -> 4.80 ms per affordance step): 64 x 64 tiles, four times the
    // workgroups — twice the LDS bytes per MFMA, but the chip is filled
    static const long small_thr = getenv("HULC_GATHER_SMALL") ? atol(getenv("HULC_GATHER_SMALL")) : 256;
    if (wide && p.inner_log2 >= 5 && (long)grid.x * grid.y < small_thr) {
        dim3 g64((unsigned)((Mtot + 63) / 64), p.Cout / 64);
        // 128- / 64-wide k tiles where a tap holds them: these launches are a chain of (load, LDS write, barrier, 2 MFMAs per wave) steps — a quarter /
        // half as many (affordance step, 32 images: 4.81 ms with 32-wide tiles, 4.67 with 64, 4.57 with 128)
        static const int kt64 = getenv("HULC_GATHER_SMALL_KT64") ? atoi(getenv("HULC_GATHER_SMALL_KT64")) : 2;
        if (kt64 >= 2 && p.inner_log2 >= 7) conv_gather_bf16_kernel<1, 1, 2, 2, true, 128><<<g64, 256, 0, s>>>(p);
        else if (kt64 && p.inner_log2 >= 6) conv_gather_bf16_kernel<1, 1, 2, 2, true, 64><<<g64, 256, 0, s>>>(p);
        else conv_gather_bf16_kernel<1, 1, 2, 2, true><<<g64, 256, 0, s>>>(p);
        return true;
    }
    if (p.inner_log2 >= 5) {
        if (wide) conv_gather_bf16_kernel<2, 2, 2, 2, true><<<grid, 256, 0, s>>>(p);
        else if (tall) conv_gather_bf16_kernel<2, 2, 4, 1, true><<<grid, 256, 0, s>>>(p);
        else conv_gather_bf16_kernel<2, 1, 2, 2, true><<<grid, 256, 0, s>>>(p);
    } else if (p.grid_kw > 0) {
        if (wide) conv_gather_bf16_kernel<2, 2, 2, 2, false><<<grid, 256, 0, s>>>(p);
        else if (tall) conv_gather_bf16_kernel<2, 2, 4, 1, false><<<grid, 256, 0, s>>>(p);
        else conv_gather_bf16_kernel<2, 1, 2, 2, false><<<grid, 256, 0, s>>>(p);
    } else return false;
    return true;
}

template <typename CT>
void launch_gather(const GatherP& p, hipStream_t s) {
    if (launch_gather_raw<CT>(p, s)) return;
    const long Mtot = (long)p.Nimg * p.OH * p.OW;
    if (p.Cout <= 32) {
        dim3 grid((unsigned)((Mtot + 127) / 128), 1);
        conv_gather_kernel<CT, 1, 1, 4, 1><<<grid, 256, 0, s>>>(p);
    } else if (p.Cout % 128 == 0 && sizeof(CT) == 2 && !getenv("HULC_GATHER_N64")) {
        // the ResNet trunk's wide layers: a 128 x 128 tile halves the LDS bytes per MFMA (each wave owns 64 x 64)
        dim3 grid((unsigned)((Mtot + 127) / 128), p.Cout / 128);
        conv_gather_kernel<CT, 2, 2, 2, 2><<<grid, 256, 0, s>>>(p);
    } else {
        dim3 grid((unsigned)((Mtot + 127) / 128), (p.Cout + 63) / 64);
        conv_gather_kernel<CT, 2, 1, 2, 2><<<grid, 256, 0, s>>>(p);
    }
}

int log2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return (1 << l) == v ? l : -1;
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradP {
    GatherP g;                 // gather description of the forward input (X, taps, strides, OH/OW...)
    const void* dY; int dy_dtype; long dy_sn, dy_sy, dy_sx;   // upstream gradient, channels contiguous
    float* partial_w;          // [P][Cout][K]
    float* partial_b;          // [P][Cout] or null
    long pix_per_block;        // pixels per workgroup (multiple of the reduction tile)
    int pair_fastest;          // staging item order: consecutive lanes walk pixel pairs (1) or k-chunks (0)
};

// write two pixels' worth of one 8-wide chunk transposed into the LDS tile: element j of the chunk
// goes to row (row0 + j), column pair (m, m+1) of the reduction tile.
template <typename CT> HULC_DEVICE void store_pair_transposed(char* tile, int row0, int m, const Chunk8& a, const Chunk8& b);
template <> HULC_DEVICE void store_pair_transposed<bf16_t>(char* tile, int row0, int m, const Chunk8& a, const Chunk8& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *(uint32_t*)(tile + (row0 + j) * HULC_ROWB + m * 2) = pack_bf16x2(a.v[j], b.v[j]);
}
template <> HULC_DEVICE void store_pair_transposed<float>(char* tile, int row0, int m, const Chunk8& a, const Chunk8& b) {
#pragma unroll
    for (int j = 0; j < 8; ++j) *(float2*)(tile + (row0 + j) * HULC_ROWB + m * 4) = make_float2(a.v[j], b.v[j]);
}

// running (image, row, col) of an output pixel; advanced by a fixed step per reduction tile instead of
// re-dividing the flat pixel index every tile
struct PixIter {
    int n, oy, ox;
    HULC_DEVICE void init(long m, int OH, int OW) { ox = (int)(m % OW); long r = m / OW; oy = (int)(r % OH); n = (int)(r / OH); }
    HULC_DEVICE void advance(int step, int OH, int OW) {
        ox += step;
        while (ox >= OW) { ox -= OW; if (++oy >= OH) { oy = 0; ++n; } }
    }
};

// TMC = Cout / 32 (1 or 2), TNW = 32-wide k tiles per wave (workgroup k slice = 128 * TNW).
// Each workgroup: 4 waves; wave w owns k tiles [w*TNW, (w+1)*TNW) of the slice and all output channels.
template <typename CT, int TMC, int TNW>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradP p) {
    using T = MmaTraits<CT>;
    constexpr int KT = T::KT;                 // pixels per reduction tile (32 bf16 / 16 f32)
    constexpr int CO = TMC * 32, KS = 128 * TNW;
    constexpr int PAIRS = KT / 2;             // pixel pairs per tile
    constexpr int DY_ITEMS = PAIRS * (CO / 8), X_ITEMS = PAIRS * (KS / 8);
    constexpr int X_PER = (X_ITEMS + 255) / 256;
    static_assert(DY_ITEMS <= 256, "dY staging fits one pass");

    __shared__ __attribute__((aligned(16))) char smem[2 * (CO + KS) * HULC_ROWB];
    __shared__ float bpart[DY_ITEMS][8];   // per-thread bias partials, summed in a fixed order

    const GatherP& g = p.g;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long Mtot = (long)g.Nimg * g.OH * g.OW;
    const long mb = (long)blockIdx.x * p.pix_per_block;
    long me = mb + p.pix_per_block; if (me > Mtot) me = Mtot;
    const int K = g.ntaps << g.inner_log2;
    const int ks0 = blockIdx.y * KS;
    const int ntile = (int)((me - mb + KT - 1) / KT);

    f32x16_t acc[TMC][TNW];
#pragma unroll
    for (int i = 0; i < TMC; ++i)
#pragma unroll
        for (int j = 0; j < TNW; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    float bacc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bacc[j] = 0.f;

    // item -> (pixel pair, chunk): consecutive threads take consecutive chunks of one pixel pair so a
    // wave's loads cover whole contiguous channel runs.  Pixel coordinates are carried incrementally.
    const int dy_pr = p.pair_fastest ? tid % PAIRS : tid / (CO / 8), dy_ch = p.pair_fastest ? tid / PAIRS : tid % (CO / 8);
    PixIter dyit; dyit.init(mb + 2 * dy_pr < Mtot ? mb + 2 * dy_pr : Mtot - 1, g.OH, g.OW);
    PixIter xit[X_PER]; int x_pr[X_PER], x_ch[X_PER];
#pragma unroll
    for (int q = 0; q < X_PER; ++q) {
        const int id = tid + q * 256;
        x_pr[q] = p.pair_fastest ? id % PAIRS : id / (KS / 8); x_ch[q] = p.pair_fastest ? id / PAIRS : id % (KS / 8);
        const long m = mb + 2 * x_pr[q];
        xit[q].init(m < Mtot ? m : Mtot - 1, g.OH, g.OW);
    }
    auto x_base = [&](const PixIter& it, int& iy0, int& ix0) -> long {
        iy0 = it.oy * g.stride; ix0 = it.ox * g.stride;
        return (long)it.n * g.x_sn + (long)iy0 * g.x_sy + (long)ix0 * g.x_sx;
    };
    auto dy_base = [&](const PixIter& it) -> long { return (long)it.n * p.dy_sn + (long)it.oy * p.dy_sy + (long)it.ox * p.dy_sx; };

    Chunk8 dya, dyb, xa[X_PER], xb[X_PER];
    auto load_tiles = [&](int t) {
        const long mt = mb + (long)t * KT;
        if (tid < DY_ITEMS) {
            const long m = mt + 2 * dy_pr;
            PixIter nx = dyit; nx.advance(1, g.OH, g.OW);
            chunk_load_contig(dya, p.dY, p.dy_dtype, m < me ? dy_base(dyit) + dy_ch * 8 : 0);
            chunk_keep_if(dya, m < me);
            chunk_load_contig(dyb, p.dY, p.dy_dtype, m + 1 < me ? dy_base(nx) + dy_ch * 8 : 0);
            chunk_keep_if(dyb, m + 1 < me);
            dyit.advance(KT, g.OH, g.OW);
        }
#pragma unroll
        for (int q = 0; q < X_PER; ++q) {
            if (X_ITEMS % 256 == 0 || tid + q * 256 < X_ITEMS) {
                const int k0 = ks0 + x_ch[q] * 8;
                const long m = mt + 2 * x_pr[q];
                int iy0, ix0;
                PixIter nx = xit[q]; nx.advance(1, g.OH, g.OW);
                const long b0 = x_base(xit[q], iy0, ix0);
                gather_chunk(xa[q], g, b0, iy0, ix0, m < me ? k0 : K, K);           // k0 = K -> zero chunk, still one unconditional load
                const long b1 = x_base(nx, iy0, ix0);
                gather_chunk(xb[q], g, b1, iy0, ix0, m + 1 < me ? k0 : K, K);
                xit[q].advance(KT, g.OH, g.OW);
            }
        }
    };
    auto store_tiles = [&](int buf) {
        char* As = smem + buf * (CO + KS) * HULC_ROWB;   // dY^T : rows = output channel
        char* Bs = As + CO * HULC_ROWB;                   // X^T  : rows = k inside the slice
        if (tid < DY_ITEMS) {
            store_pair_transposed<CT>(As, dy_ch * 8, 2 * dy_pr, dya, dyb);
#pragma unroll
            for (int j = 0; j < 8; ++j) bacc[j] += dya.v[j] + dyb.v[j];
        }
#pragma unroll
        for (int q = 0; q < X_PER; ++q)
            if (X_ITEMS % 256 == 0 || tid + q * 256 < X_ITEMS) store_pair_transposed<CT>(Bs, x_ch[q] * 8, 2 * x_pr[q], xa[q], xb[q]);
    };

    if (ntile > 0) {
        load_tiles(0);
        store_tiles(0);
    }
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) load_tiles(t + 1);
        const char* As = smem + buf * (CO + KS) * HULC_ROWB;
        const char* Bs = As + CO * HULC_ROWB;
        MmaTile<CT, TMC, TNW>::run(As, Bs + wave * TNW * 32 * HULC_ROWB, acc, lane);
        if (t + 1 < ntile) store_tiles(buf ^ 1);
        __syncthreads();
    }

    // partial dW slab: rows = co, cols = k
    float* pw = p.partial_w + (long)blockIdx.x * CO * K;
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int k = ks0 + (wave * TNW + j) * 32 + (lane & 31);
        if (k < K) {
#pragma unroll
            for (int i = 0; i < TMC; ++i)
#pragma unroll
                for (int e = 0; e < 16; ++e) pw[(long)(i * 32 + acc_row(e, lane)) * K + k] = acc[i][j][e];
        }
    }
    if (p.partial_b && blockIdx.y == 0) {
        if (tid < DY_ITEMS) {
#pragma unroll
            for (int j = 0; j < 8; ++j) bpart[tid][j] = bacc[j];
        }
        __syncthreads();
        if (tid < CO) {
            float sacc = 0.f;
            for (int pr = 0; pr < PAIRS; ++pr) sacc += bpart[p.pair_fastest ? (tid / 8) * PAIRS + pr : pr * (CO / 8) + tid / 8][tid % 8];
            p.partial_b[(long)blockIdx.x * CO + tid] = sacc;
        }
    }
}

// out[r] = sum_p partial[p][r]: workgroup = 64 outputs x 16 P-slices (fixed slice boundaries and a fixed
// combine order -> deterministic), so the P-long loop is 16x shorter and the grid is R/64 workgroups of 1024.
// perm_c > 0: r = co*K + tap*perm_c + c is written at co*K + c*perm_taps + tap (forward k order -> the parameter's OIHW order)
__global__ __launch_bounds__(1024) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, long R, int accumulate,
                                                               int perm_c, int perm_taps) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long r = (long)blockIdx.x * 64 + lane;
    const int per = (P + 15) / 16;
    const int q0 = sl * per, q1 = q0 + per < P ? q0 + per : P;
    float s = 0.f;
    if (r < R)
        for (int q = q0; q < q1; ++q) s += partial[(long)q * R + r];
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && r < R) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        long o = r;
        if (perm_c > 0) { const long K = (long)perm_c * perm_taps, co = r / K, k = r % K; o = co * K + (k % perm_c) * perm_taps + k / perm_c; }
        out[o] = accumulate ? out[o] + t : t;
    }
}

void fill_gather(GatherP& g, const hulc_conv_desc* d) {
    g.x_dtype = d->x_dtype; g.w_dtype = d->w_dtype;
    g.Nimg = d->N; g.H = d->H; g.W = d->W;
    g.OH = (d->H - d->KH) / d->stride + 1; g.OW = (d->W - d->KW) / d->stride + 1;
    g.Cout = d->Cout; g.stride = d->stride; g.check_bounds = 0; g.grid_kw = 0; g.grid_pad = 0; g.stride_x = d->stride;
    if (d->x_nchw) {   // k = (c, kh, kw): one tap per (c, kh), inner run = KW along x
        g.x_sn = (long)d->Cin * d->H * d->W; g.x_sy = d->W; g.x_sx = 1;
        g.ntaps = d->Cin * d->KH; g.inner_log2 = log2_exact(d->KW);
        for (int c = 0; c < d->Cin; ++c)
            for (int kh = 0; kh < d->KH; ++kh) {
                int t = c * d->KH + kh;
                g.tap_dy[t] = kh; g.tap_dx[t] = 0;
                g.tap_off[t] = (long)c * d->H * d->W + (long)kh * d->W;
                g.w_tap_off[t] = (long)t * d->KW;
            }
    } else {           // NHWC, k = (kh, kw, c): one tap per (kh, kw), inner run = Cin
        g.x_sn = (long)d->H * d->W * d->Cin; g.x_sy = (long)d->W * d->Cin; g.x_sx = d->Cin;
        g.ntaps = d->KH * d->KW; g.inner_log2 = log2_exact(d->Cin);
        for (int kh = 0; kh < d->KH; ++kh)
            for (int kw = 0; kw < d->KW; ++kw) {
                int t = kh * d->KW + kw;
                g.tap_dy[t] = kh; g.tap_dx[t] = kw;
                g.tap_off[t] = (long)kh * g.x_sy + (long)kw * g.x_sx;
                g.w_tap_off[t] = (long)t * d->Cin;
            }
    }
    g.ldw = (long)g.ntaps << g.inner_log2;
}

int validate(const hulc_conv_desc* d, const char* who) {
    if (!d) return hulc_fail(-1, "conv: null descriptor");
    if (d->N <= 0 || d->H < d->KH || d->W < d->KW || d->stride <= 0) return hulc_fail(-2, "conv: bad geometry");
    const int taps = d->x_nchw ? d->Cin * d->KH : d->KH * d->KW;
    const int inner = d->x_nchw ? d->KW : d->Cin;
    if (taps > HULC_MAX_TAPS) return hulc_fail(-3, "conv: too many taps");
    if (log2_exact(inner) < 3) return hulc_fail(-4, "conv: inner run (KW for NCHW input, Cin for NHWC) must be a power of two >= 8");
    if (d->Cout != 32 && d->Cout != 64) return hulc_fail(-5, "conv: Cout must be 32 or 64");
    if (d->compute == HULC_F32 && (d->x_dtype != HULC_F32 || d->w_dtype != HULC_F32)) return hulc_fail(-6, "conv: f32 compute requires f32 operands");
    if (d->frame_index && !d->x_u8_nhwc) return hulc_fail(-7, "conv: frame_index addresses a uint8 NHWC episode store (x_u8_nhwc = 1)");
    (void)who;
    return 0;
}

}  // namespace

// ReLU sign planes of a stored activation y [npix][C] (C % 32 == 0): planes [C / 32][npix], bit c % 32 of plane c / 32 = (y > 0) — the pass
// behind kernels that do not write the planes from their epilogue
__global__ __launch_bounds__(256) void relu_bits_kernel(const void* __restrict__ y, int y_dtype, long npix, int nw, unsigned* __restrict__ bits) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // word i of pixel-major order: pixel i / nw, plane i % nw
    if (i >= npix * nw) return;
    unsigned m = 0;
    if (y_dtype == HULC_BF16) {
        const uint4* q = (const uint4*)((const uint16_t*)y + i * 32);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint4 v = q[j];
            const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                m |= (((w[e] & 0xffffu) - 1u < 0x7fffu ? 1u : 0u) | ((w[e] >> 16) - 1u < 0x7fffu ? 2u : 0u)) << (8 * j + 2 * e);
        }
    } else {
        const float* q = (const float*)y + i * 32;
#pragma unroll
        for (int c = 0; c < 32; ++c) m |= (q[c] > 0.f ? 1u : 0u) << c;
    }
    bits[(i % nw) * npix + i / nw] = m;
}
static void launch_relu_bits(const hulc_conv_desc* d, const void* y, hipStream_t s) {
    const int OH = (d->H - d->KH) / d->stride + 1, OW = (d->W - d->KW) / d->stride + 1;
    const long npix = (long)d->N * OH * OW;
    const int nw = d->Cout / 32;
    relu_bits_kernel<<<(unsigned)((npix * nw + 255) / 256), 256, 0, s>>>(y, d->y_dtype, npix, nw, (unsigned*)d->relu_bits);
}

extern "C" int hulc_conv2d_fwd(const hulc_conv_desc* d, const void* x, const void* w, const float* bias, void* y, void* stream) {
    int rc = validate(d, "fwd"); if (rc) return rc;
    if (!x || !w || !y) return hulc_fail(-1, "hulc_conv2d_fwd: null pointer");
    GatherP g; fill_gather(g, d);
    bool bits_pending = false;
    if (d->relu_bits && (d->Cout % 32 || !d->relu)) return hulc_fail(-4, "hulc_conv2d_fwd: relu_bits needs relu and Cout % 32 == 0");
    if (d->y_bf16 && ((d->y_dtype != HULC_F32 && d->y_dtype != HULC_F16) || d->relu_bits || ((uintptr_t)d->y_bf16 % 16)))
        return hulc_fail(-4, "hulc_conv2d_fwd: y_bf16 goes with an fp32 / fp16 output, no sign planes, 16-byte aligned");
    if (d->y_dtype == HULC_F16 && !d->y_bf16) return hulc_fail(-4, "hulc_conv2d_fwd: an fp16 output is the twin of a bf16 map (y_bf16)");
    g.X = x; g.Wt = w; g.bias = bias; g.Y = y; g.mask = nullptr; g.mask_dtype = HULC_F32; g.mask_scale = 1.f; g.add = nullptr; g.add_dtype = HULC_F32;
    g.y_dtype = d->y_dtype; g.relu = d->relu;
    g.y_sn = (long)g.OH * g.OW * d->Cout; g.y_sy = (long)g.OW * d->Cout; g.y_sx = d->Cout;
    if (d->compute == HULC_BF16 && !d->x_nchw && d->KH * d->KW <= 16 && d->Cout % 32 == 0 && d->Cout / 32 <= 4) {
        const int nset = d->Cout / 32;
        int cOH[4], cOW[4], cco[4]; long cyo[4], cw0[4], ctap[4 * 16];
        for (int c = 0; c < nset; ++c) {
            cOH[c] = g.OH; cOW[c] = g.OW; cyo[c] = 0; cco[c] = 32 * c; cw0[c] = 32 * c;
            for (int t = 0; t < 16; ++t) ctap[c * 16 + t] = t < d->KH * d->KW ? g.w_tap_off[t] : 0;
        }
        rc = hulc_conv_band_dispatch(d->Cin, nset, d->KH, d->KW, d->stride, x, d->x_dtype, d->N, d->H, d->W, 0, 0, g.x_sn, g.x_sy, g.x_sx,
                                     y, d->y_dtype, g.y_sn, g.y_sy, g.y_sx, w, d->w_dtype, g.ldw, bias, nullptr, HULC_F32, d->relu, nset,
                                     cOH, cOW, cyo, cco, cw0, ctap, nullptr, (d->relu && d->y_dtype == HULC_BF16) ? (unsigned*)d->relu_bits : nullptr, nullptr,
                                     d->Cout, (d->y_dtype != HULC_BF16 ? d->y_bf16 : nullptr), (hipStream_t)stream);
        if (rc == 1 && d->relu_bits)            // (the planes' preconditions failed, not the geometry: the same launch without them)
            rc = hulc_conv_band_dispatch(d->Cin, nset, d->KH, d->KW, d->stride, x, d->x_dtype, d->N, d->H, d->W, 0, 0, g.x_sn, g.x_sy, g.x_sx,
                                         y, d->y_dtype, g.y_sn, g.y_sy, g.y_sx, w, d->w_dtype, g.ldw, bias, nullptr, HULC_F32, d->relu, nset,
                                         cOH, cOW, cyo, cco, cw0, ctap, nullptr, nullptr, nullptr, 0, (d->y_dtype != HULC_BF16 ? d->y_bf16 : nullptr), (hipStream_t)stream), bits_pending = d->relu_bits != nullptr;
        else bits_pending = false;
        if (rc < 0) return rc;
        if (rc == 0) { if (bits_pending) launch_relu_bits(d, y, (hipStream_t)stream); return hulc_check_launch("hulc_conv2d_fwd(band)"); }
        bits_pending = d->relu_bits != nullptr;
    }
    if (d->y_dtype == HULC_F16)
        return hulc_fail(-6, "hulc_conv2d_fwd: the fp16 twin is stored by the direct-to-LDS band kernel only (bf16 compute, NHWC 23 x 23 x 64 -> 64, 3 x 3)");
    if (d->compute == HULC_BF16 && d->x_nchw && (d->x_dtype == HULC_F32 || d->x_u8_nhwc) && d->Cin == 3 && d->Cout == 32 && d->KH == 8 && d->KW == 8 &&
        d->stride == 4) {
        unsigned* planes = (d->relu && d->y_dtype == HULC_BF16) ? (unsigned*)d->relu_bits : nullptr;
        rc = hulc_conv1_band_dispatch((const float*)x, w, d->w_dtype, g.ldw, bias, y, d->y_dtype, d->relu, d->N, d->H, d->W, d->x_u8_nhwc, d->aug_pad,
                                      d->aug_shift, d->frame_index, planes, d->w_lo, d->x2, d->n_split, d->x_slot, d->x2_slot, (hipStream_t)stream);
        if (rc < 0) return rc;
        if (rc != 0 && d->x2) return hulc_fail(-6, "hulc_conv2d_fwd: a second frame tensor (x2) is taken by the conv1 band kernel only");
        if (rc != 0 && (d->x_slot || d->x2_slot)) return hulc_fail(-6, "hulc_conv2d_fwd: frame slots are taken by the conv1 band kernel only");
        if (rc != 0 && d->w_lo) return hulc_fail(-6, "hulc_conv2d_fwd: split operands (w_lo) are taken by the conv1 band kernel only");
        if (rc == 0) { if (d->relu_bits && !planes) launch_relu_bits(d, y, (hipStream_t)stream); return hulc_check_launch("hulc_conv2d_fwd(conv1 band)"); }
    }
    if (d->x_u8_nhwc) return hulc_fail(-6, "hulc_conv2d_fwd: uint8 frames are consumed by the conv1 band kernel only (bf16 compute, 3 -> 32, 8x8 stride 4, W % 4 == 0)");
    if (d->x_slot || d->x2_slot) return hulc_fail(-6, "hulc_conv2d_fwd: frame slots are taken by the conv1 band kernel only");
    if (d->compute == HULC_F32) launch_gather<float>(g, (hipStream_t)stream); else launch_gather<bf16_t>(g, (hipStream_t)stream);
    if (d->relu_bits) launch_relu_bits(d, y, (hipStream_t)stream);       // kernels without the epilogue: the planes from a second pass over y
    if (d->y_bf16) {                                                     // (the band kernels store the copy themselves: here a cast launch follows)
        rc = hulc_cast_f32_to_bf16((const float*)y, d->y_bf16, (long)d->N * g.OH * g.OW * d->Cout, stream);
        if (rc) return rc;
    }
    return hulc_check_launch("hulc_conv2d_fwd");
}

// Zero-padded convolution of the frozen ResNet trunk (VisionR3M): NHWC x, OHWI weights with the BatchNorm scale
// folded in, bias = the folded shift, optional residual `add` (same shape / dtype as y) summed before the ReLU.
extern "C" int hulc_conv2d_padded_fwd(const hulc_conv_desc* d, int pad, const void* x, const void* w, const float* bias, const void* add, void* y,
                                      void* stream) {
    if (!d || !x || !w || !y) return hulc_fail(-1, "hulc_conv2d_padded_fwd: null pointer");
    if (d->x_nchw || d->x_u8_nhwc) return hulc_fail(-7, "hulc_conv2d_padded_fwd: NHWC activations only");
    if (d->N <= 0 || d->stride <= 0 || pad < 0 || d->H + 2 * pad < d->KH || d->W + 2 * pad < d->KW) return hulc_fail(-2, "hulc_conv2d_padded_fwd: bad geometry");
    if (d->KH * d->KW > HULC_MAX_TAPS) return hulc_fail(-3, "hulc_conv2d_padded_fwd: too many taps");
    if (log2_exact(d->Cin) < 3) return hulc_fail(-4, "hulc_conv2d_padded_fwd: Cin must be a power of two >= 8");
    if (d->Cout % 32) return hulc_fail(-5, "hulc_conv2d_padded_fwd: Cout must be a multiple of 32");
    if (d->compute == HULC_F32 && (d->x_dtype != HULC_F32 || d->w_dtype != HULC_F32)) return hulc_fail(-6, "hulc_conv2d_padded_fwd: f32 compute requires f32 operands");
    GatherP g;
    g.X = x; g.Wt = w; g.bias = bias; g.Y = y; g.mask = nullptr; g.mask_dtype = HULC_F32; g.mask_scale = 1.f; g.add = add; g.add_dtype = d->y_dtype;
    g.x_dtype = d->x_dtype; g.w_dtype = d->w_dtype; g.y_dtype = d->y_dtype; g.relu = d->relu;
    g.Nimg = d->N; g.H = d->H; g.W = d->W; g.Cout = d->Cout; g.stride = d->stride; g.check_bounds = pad > 0;
    g.grid_kw = d->KW; g.grid_pad = pad; g.stride_x = d->stride;
    g.OH = (d->H + 2 * pad - d->KH) / d->stride + 1; g.OW = (d->W + 2 * pad - d->KW) / d->stride + 1;
    g.x_sn = (long)d->H * d->W * d->Cin; g.x_sy = (long)d->W * d->Cin; g.x_sx = d->Cin;
    g.y_sn = (long)g.OH * g.OW * d->Cout; g.y_sy = (long)g.OW * d->Cout; g.y_sx = d->Cout;
    g.ntaps = d->KH * d->KW; g.inner_log2 = log2_exact(d->Cin);
    for (int kh = 0; kh < d->KH; ++kh)
        for (int kw = 0; kw < d->KW; ++kw) {
            const int t = kh * d->KW + kw;
            g.tap_dy[t] = kh - pad; g.tap_dx[t] = kw - pad;
            g.tap_off[t] = (long)(kh - pad) * g.x_sy + (long)(kw - pad) * g.x_sx;
            g.w_tap_off[t] = (long)t * d->Cin;
        }
    g.ldw = (long)g.ntaps << g.inner_log2;
    // 64 -> 64, 3 x 3, stride 1 (ResNet layer1): the LDS-band kernel stages every input row once instead of gathering it nine times
    if (d->compute == HULC_BF16 && d->Cin == 64 && d->Cout == 64 && d->KH == 3 && d->KW == 3 && d->stride == 1 && d->x_dtype == HULC_BF16 &&
        d->y_dtype == HULC_BF16 && d->w_dtype == HULC_BF16 && !getenv("HULC_NO_BAND_PADDED")) {
        int cOH[4], cOW[4], cco[4]; long cyo[4], cw0[4], ctap[4 * 16];
        for (int c = 0; c < 2; ++c) {
            cOH[c] = g.OH; cOW[c] = g.OW; cyo[c] = 0; cco[c] = 32 * c; cw0[c] = 32 * c;
            for (int t = 0; t < 16; ++t) ctap[c * 16 + t] = t < 9 ? g.w_tap_off[t] : 0;
        }
        const int rc = hulc_conv_band_dispatch(64, 2, 3, 3, 1, x, d->x_dtype, d->N, d->H, d->W, pad, pad, g.x_sn, g.x_sy, g.x_sx, y, d->y_dtype, g.y_sn,
                                               g.y_sy, g.y_sx, w, d->w_dtype, g.ldw, bias, nullptr, HULC_BF16, d->relu, 2, cOH, cOW, cyo, cco, cw0, ctap,
                                               add, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
        if (rc < 0) return rc;
        if (rc == 0) return hulc_check_launch("hulc_conv2d_padded_fwd(band)");
    }
    if (d->compute == HULC_F32) launch_gather<float>(g, (hipStream_t)stream); else launch_gather<bf16_t>(g, (hipStream_t)stream);
    return hulc_check_launch("hulc_conv2d_padded_fwd");
}

// The ResNet stem (7 x 7, stride 2, padding 3, 3 input channels) on the packed input hulc_r3m_normalize_packed writes: pixels of 4 bf16
// channels (RGB + 0) inside a zero border, so TWO neighbouring pixels are one aligned 16-byte piece.  Seen as "double pixels" the stem is
// a 7 x 4 convolution with stride (2, 1) and no bounds checks: K = 7 * 4 * 8 = 224 (seven 32-wide k tiles exactly) instead of the
// 7 * 7 * 8 = 392 of the channel-padded NHWC8 form, and a pixel costs 8 gathered bytes instead of 16.
extern "C" int hulc_r3m_packed_width(int W) { return 2 * ((W + 6 - 7) / 2) + 8; }

extern "C" int hulc_r3m_stem_fwd(const void* xp, const void* w, const float* bias, void* y, int y_dtype, int N, int H, int W, int Cout, int relu,
                                 void* stream) {
    if (!xp || !w || !y) return hulc_fail(-1, "hulc_r3m_stem_fwd: null pointer");
    if (N <= 0 || H < 1 || W < 1 || Cout % 64) return hulc_fail(-2, "hulc_r3m_stem_fwd: bad geometry (Cout must be a multiple of 64)");
    const int Wp = hulc_r3m_packed_width(W), Hp = H + 6;
    GatherP g;
    g.X = xp; g.Wt = w; g.bias = bias; g.Y = y; g.mask = nullptr; g.mask_dtype = HULC_F32; g.mask_scale = 1.f; g.add = nullptr; g.add_dtype = HULC_F32;
    g.x_dtype = HULC_BF16; g.w_dtype = HULC_BF16; g.y_dtype = y_dtype; g.relu = relu;
    g.Nimg = N; g.H = Hp; g.W = Wp / 2; g.Cout = Cout; g.stride = 2; g.stride_x = 1; g.check_bounds = 0;
    g.OH = (H + 6 - 7) / 2 + 1; g.OW = (W + 6 - 7) / 2 + 1;
    g.x_sn = (long)Hp * Wp * 4; g.x_sy = (long)Wp * 4; g.x_sx = 8;
    g.y_sn = (long)g.OH * g.OW * Cout; g.y_sy = (long)g.OW * Cout; g.y_sx = Cout;
    g.ntaps = 28; g.inner_log2 = 3; g.grid_kw = 4; g.grid_pad = 0;
    for (int t = 0; t < 28; ++t) {
        g.tap_dy[t] = t / 4; g.tap_dx[t] = t % 4;
        g.tap_off[t] = (long)(t / 4) * g.x_sy + (long)(t % 4) * g.x_sx;
        g.w_tap_off[t] = (long)t * 8;
    }
    g.ldw = 224;
    if (!launch_gather_raw<bf16_t>(g, (hipStream_t)stream)) launch_gather<bf16_t>(g, (hipStream_t)stream);
    return hulc_check_launch("hulc_r3m_stem_fwd");
}

// dX (NHWC [N][H][W][Cin]) from dY (NHWC [N][OH][OW][Cout]); wt = weights permuted to [Cin][KH][KW][Cout].
// relu_src (optional, same shape as dX): dX *= (relu_src > 0)  — the ReLU that produced this conv's input.
extern "C" int hulc_conv2d_bwd_data(const hulc_conv_desc* d, const void* dy, const void* wt, void* dx, const void* relu_src, void* stream) {
    if (!d || !dy || !wt || !dx) return hulc_fail(-1, "hulc_conv2d_bwd_data: null pointer");
    if (d->x_nchw) return hulc_fail(-7, "hulc_conv2d_bwd_data: only NHWC activations have a data gradient on this path");
    if (log2_exact(d->Cout) < 3) return hulc_fail(-4, "conv bwd_data: Cout must be a power of two >= 8");
    if (d->Cin != 32 && d->Cin != 64) return hulc_fail(-5, "conv bwd_data: Cin must be 32 or 64");
    const int OH = (d->H - d->KH) / d->stride + 1, OW = (d->W - d->KW) / d->stride + 1, s = d->stride;
    const int xsz = d->x_dtype == HULC_F32 ? 4 : 2;
    // LDS-band kernel: all stride^2 parity classes x Cin/32 channel tiles in ONE launch sharing one staged dY band
    if (d->compute == HULC_BF16 && !getenv("HULC_NO_BAND_DGRAD") && d->Cin % 32 == 0 && d->KH % s == 0 && d->KW % s == 0) {
        const int U = d->KH / s, V = d->KW / s, tiles = d->Cin / 32, nset = s * s * tiles;
        if (nset <= 4 && U * V <= 16) {
            int cOH[4], cOW[4], cco[4]; long cyo[4], cw0[4], ctap[4 * 16];
            int c = 0;
            for (int py = 0; py < s; ++py)
                for (int px = 0; px < s; ++px)
                    for (int j = 0; j < tiles; ++j, ++c) {
                        cOH[c] = (d->H - py + s - 1) / s; cOW[c] = (d->W - px + s - 1) / s;
                        cyo[c] = ((long)py * d->W + px) * d->Cin; cco[c] = 32 * j; cw0[c] = 32 * j;
                        for (int t = 0; t < 16; ++t) ctap[c * 16 + t] = 0;
                        for (int ty = 0; ty < U; ++ty)
                            for (int tx = 0; tx < V; ++tx)
                                ctap[c * 16 + ty * V + tx] = ((long)(py + s * (U - 1 - ty)) * d->KW + (px + s * (V - 1 - tx))) * d->Cout;
                    }
            int brc = hulc_conv_band_dispatch(d->Cout, nset, U, V, 1, dy, d->y_dtype, d->N, OH, OW, U - 1, V - 1, (long)OH * OW * d->Cout,
                                                    (long)OW * d->Cout, d->Cout, dx, d->x_dtype, (long)d->H * d->W * d->Cin,
                                                    (long)s * d->W * d->Cin, (long)s * d->Cin, wt, d->w_dtype, (long)d->KH * d->KW * d->Cout,
                                                    nullptr, relu_src, d->x_dtype, 0, nset, cOH, cOW, cyo, cco, cw0, ctap, nullptr, nullptr,
                                                    (const unsigned*)d->relu_bits, d->Cin, nullptr, (hipStream_t)stream);
            if (brc == 1 && d->relu_bits)       // the planes' preconditions failed: the same launch with the activation as the mask
                brc = hulc_conv_band_dispatch(d->Cout, nset, U, V, 1, dy, d->y_dtype, d->N, OH, OW, U - 1, V - 1, (long)OH * OW * d->Cout,
                                              (long)OW * d->Cout, d->Cout, dx, d->x_dtype, (long)d->H * d->W * d->Cin,
                                              (long)s * d->W * d->Cin, (long)s * d->Cin, wt, d->w_dtype, (long)d->KH * d->KW * d->Cout,
                                              nullptr, relu_src, d->x_dtype, 0, nset, cOH, cOW, cyo, cco, cw0, ctap, nullptr, nullptr, nullptr, 0, nullptr, (hipStream_t)stream);
            if (brc < 0) return brc;
            if (brc == 0) return hulc_check_launch("hulc_conv2d_bwd_data(band)");
        }
    }
    for (int py = 0; py < s; ++py)
        for (int px = 0; px < s; ++px) {
            GatherP g;
            g.X = dy; g.x_dtype = d->y_dtype; g.Wt = wt; g.w_dtype = d->w_dtype; g.bias = nullptr; g.add = nullptr; g.add_dtype = HULC_F32;
            g.Nimg = d->N; g.H = OH; g.W = OW;                       // the gathered tensor is dY
            g.OH = (d->H - py + s - 1) / s; g.OW = (d->W - px + s - 1) / s;   // this class' sub-grid of dX
            if (g.OH <= 0 || g.OW <= 0) continue;
            g.Cout = d->Cin; g.stride = 1; g.check_bounds = 1; g.grid_kw = 0; g.grid_pad = 0; g.stride_x = 1;
            g.x_sn = (long)OH * OW * d->Cout; g.x_sy = (long)OW * d->Cout; g.x_sx = d->Cout;
            g.inner_log2 = log2_exact(d->Cout);
            int t = 0;
            for (int kh = py; kh < d->KH; kh += s)
                for (int kw = px; kw < d->KW; kw += s) {
                    if (t >= HULC_MAX_TAPS) return hulc_fail(-3, "conv bwd_data: too many taps");
                    const int u = (kh - py) / s, v = (kw - px) / s;
                    g.tap_dy[t] = -u; g.tap_dx[t] = -v;
                    g.tap_off[t] = -(long)u * g.x_sy - (long)v * g.x_sx;
                    g.w_tap_off[t] = ((long)kh * d->KW + kw) * d->Cout;
                    ++t;
                }
            if (t == 0) continue;
            g.ntaps = t;
            g.ldw = (long)d->KH * d->KW * d->Cout;
            const long yoff = ((long)py * d->W + px) * d->Cin;
            g.Y = (char*)dx + yoff * xsz; g.y_dtype = d->x_dtype;
            g.y_sn = (long)d->H * d->W * d->Cin; g.y_sy = (long)s * d->W * d->Cin; g.y_sx = (long)s * d->Cin;
            g.mask = relu_src ? (const char*)relu_src + yoff * xsz : nullptr; g.mask_dtype = d->x_dtype; g.mask_scale = 1.f;
            g.relu = 0;
            if (d->compute == HULC_F32) {
                if (d->y_dtype != HULC_F32 || d->w_dtype != HULC_F32) return hulc_fail(-6, "conv bwd_data: f32 compute requires f32 operands");
                launch_gather<float>(g, (hipStream_t)stream);
            } else launch_gather<bf16_t>(g, (hipStream_t)stream);
        }
    return hulc_check_launch("hulc_conv2d_bwd_data");
}

// pixel split: ~1024 workgroups in total (4 per CU) with at least 256 pixels each
static void wgrad_split(const hulc_conv_desc* d, long& P, long& ppb, long& K, long& Mtot) {
    const int OH = (d->H - d->KH) / d->stride + 1, OW = (d->W - d->KW) / d->stride + 1;
    K = (long)d->Cin * d->KH * d->KW;
    Mtot = (long)d->N * OH * OW;
    const long kslices = (K + 255) / 256;
    ppb = (Mtot * kslices + 1023) / 1024;
    if (ppb < 256) ppb = 256;
    ppb = (ppb + 31) / 32 * 32;
    P = (Mtot + ppb - 1) / ppb;
}

extern "C" long hulc_conv2d_bwd_weight_workspace(const hulc_conv_desc* d) {
    if (!d) return -1;
    long P, ppb, K, Mtot;
    wgrad_split(d, P, ppb, K, Mtot);
    if (P < 512) P = 512;                                    // the band kernel writes one slab per persistent workgroup (<= 512)
    return P * d->Cout * (K + 1) * (long)sizeof(float);
}

// dW ([Cout][K] in the forward k order) and db ([Cout]) from x and dY; ws = workspace of
// hulc_conv2d_bwd_weight_workspace() bytes.
extern "C" int hulc_conv2d_bwd_weight(const hulc_conv_desc* d, const void* x, const void* dy, float* dw, float* db, void* ws, void* stream) {
    int rc = validate(d, "bwd_weight"); if (rc) return rc;
    if (!x || !dy || !dw || !ws) return hulc_fail(-1, "hulc_conv2d_bwd_weight: null pointer");
    if (d->compute == HULC_BF16) {                           // LDS-band kernel (conv_wgrad_band.hip) for the geometries it covers
        const long wsb = hulc_conv2d_bwd_weight_workspace(d);
        const int brc = hulc_conv_wgrad_band_dispatch(d->x_nchw, d->Cin, d->Cout, d->KH, d->KW, d->stride, x, d->x_dtype, dy, d->y_dtype,
                                                      d->N, d->H, d->W, dw, db, ws, wsb, d->dw_oihw, d->dw_accumulate, d->x_u8_nhwc, d->aug_pad,
                                                      d->aug_shift, d->frame_index, d->x2, d->n_split, d->x_slot, d->x2_slot, (hipStream_t)stream);
        if (brc < 0) return brc;
        if (brc == 0) return hulc_check_launch("hulc_conv2d_bwd_weight(band)");
    }
    if (d->x2) return hulc_fail(-6, "hulc_conv2d_bwd_weight: a second frame tensor (x2) is taken by the conv1 band kernel only");
    if (d->x_slot || d->x2_slot) return hulc_fail(-6, "hulc_conv2d_bwd_weight: frame slots are taken by the conv1 band kernel only");
    if (d->x_u8_nhwc) return hulc_fail(-6, "hulc_conv2d_bwd_weight: uint8 frames are consumed by the conv1 band kernel only");
    WgradP p; fill_gather(p.g, d);
    p.g.X = x; p.g.Wt = nullptr; p.g.bias = nullptr; p.g.Y = nullptr; p.g.mask = nullptr;
    p.dY = dy; p.dy_dtype = d->y_dtype;
    p.dy_sn = (long)p.g.OH * p.g.OW * d->Cout; p.dy_sy = (long)p.g.OW * d->Cout; p.dy_sx = d->Cout;
    long P, ppb, K, Mtot;
    wgrad_split(d, P, ppb, K, Mtot);
    p.pix_per_block = ppb;
    {
        // staging order: consecutive lanes walk pixel pairs (conflict-light transposed LDS writes; measured faster than
        // chunk-fastest on all six layers: tools/wgrad_bench.py).  HULC_WGRAD_PAIR_FASTEST=0 keeps the other order for A/B runs.
        const char* e = getenv("HULC_WGRAD_PAIR_FASTEST");
        p.pair_fastest = e ? atoi(e) : 1;
    }
    p.partial_w = (float*)ws;
    p.partial_b = db ? p.partial_w + P * d->Cout * K : nullptr;
    dim3 grid((unsigned)P, (unsigned)((K + 255) / 256));
    hipStream_t s = (hipStream_t)stream;
    if (d->compute == HULC_F32) {
        if (d->y_dtype != HULC_F32) return hulc_fail(-6, "conv bwd_weight: f32 compute requires f32 operands");
        if (d->Cout == 32) conv_wgrad_kernel<float, 1, 2><<<grid, 256, 0, s>>>(p); else conv_wgrad_kernel<float, 2, 2><<<grid, 256, 0, s>>>(p);
    } else {
        if (d->Cout == 32) conv_wgrad_kernel<bf16_t, 1, 2><<<grid, 256, 0, s>>>(p); else conv_wgrad_kernel<bf16_t, 2, 2><<<grid, 256, 0, s>>>(p);
    }
    const long R = (long)d->Cout * K;
    const int perm_c = (d->dw_oihw && !d->x_nchw) ? d->Cin : 0;
    reduce_partials_kernel<<<(unsigned)((R + 63) / 64), 1024, 0, s>>>(p.partial_w, dw, (int)P, R, d->dw_accumulate, perm_c, d->KH * d->KW);
    if (db) reduce_partials_kernel<<<1, 1024, 0, s>>>(p.partial_b, db, (int)P, d->Cout, d->dw_accumulate, 0, 0);
    return hulc_check_launch("hulc_conv2d_bwd_weight");
}
