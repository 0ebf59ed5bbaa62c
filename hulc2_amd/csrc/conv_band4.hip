// conv_band4.hip — the LDS-band convolution on a dataflow below 1 KB of LDS per MFMA (round 6).
//
// reference arithmetic: nn.Conv2d(+ReLU) of hulc2/models/perceptual_encoders/vision_network.py:41-46 (conv2 / conv3 of the static camera)
// and autograd's conv2d input gradient of the same layers.
//
// conv_band.hip gives each of 8 waves ONE 32-channel weight tile: every pixel fragment is pulled out of the LDS once per weight set, one
// ds_read_b128 (1 KB) per MFMA, two 256-register waves per SIMD.  Here a workgroup is FOUR waves, one per SIMD, 512 registers each:
//   * every wave keeps ALL weight sets of the launch in registers — most of them in the accumulation half of the unified file, read from LDS
//     straight into AGPRs and taken by the MFMAs as A operands where they lie: conv3 forward / data gradient 2 x 36 k-steps = 288 registers,
//     conv2 forward 2 x 32 = 256, conv2 data gradient 4 classes x 16 = 256;
//   * the waves split the band's PIXEL tiles; a pixel fragment read once feeds one MFMA per weight set — 2 (forward, conv3 data gradient)
//     or 4 (the four parity classes of conv2's data gradient read the same 2 x 2 neighbourhood of dY) independent accumulator chains;
//   * the sets of a launch therefore share ONE pixel enumeration: the largest class grid (OHmax x OWmax); a class with a shorter grid
//     computes the positions it does not own and drops them at the store (conv2's data gradient: 625 instead of 600 positions per frame);
//   * the band width is a template parameter: every fragment read is lane base + IMMEDIATE (no vector add per k-step in the MFMA loop);
//   * stride-2 bands are stored as two column-parity planes per row: the 16 lanes of a ds_read_b128 group then step through the LDS by ONE
//     80-byte pixel (20 dwords: sixteen distinct 16-byte bank columns) instead of two (40 dwords: two lanes per column);
//   * with one wave per SIMD nothing hides a wave's own address arithmetic, so the staging plan runs over the SOURCE: the frame rows a band
//     needs are one contiguous piece of memory, chunk ci of it is at piece + 16 ci (1 KB of consecutive bytes per wave instruction, no
//     bounds logic), zero padding is written once per launch.
// Register prefetch of the next band behind the MFMA loops, single LDS band, two barriers per unit, the weight prologue through LDS, the
// packed-word epilogue and the sign planes are conv_band.hip's.  bf16 in / out, dense NHWC input, one frame (or band of rows) per unit, no
// residual, no activation-tensor mask: everything else stays on conv_band.hip / the gather kernel.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "conv_band.h"
#include <stdio.h>
#include <stdlib.h>
#include <utility>

using namespace hulc_band;

namespace {

// weight fragments that live in the ACCUMULATION half of the register file for the whole launch: an "=a" result is an AGPR-class value for
// the register allocator (it cannot be traded against the staging registers, and an MFMA takes it as its A operand where it lies).  The
// compiler does not know this is an LDS read: the caller closes a batch with lds_wait().
template <int OFF>
HULC_DEVICE bf16x8_t lds_read_to_agpr(unsigned lds_addr) {
    bf16x8_t v;
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=a"(v) : "v"(lds_addr), "n"(OFF));
    return v;
}
HULC_DEVICE void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

template <int NV, int KSTEPS, int WS, int N, int... I>
HULC_DEVICE void load_agpr_weights(bf16x8_t (&wa)[N], unsigned base, std::integer_sequence<int, I...>) {
    // entry I of wa = fragment (set, k-step) = divmod(NV + I, KSTEPS): LDS row block set * 32, k offset 32 bytes per step
    ((wa[I] = lds_read_to_agpr<((NV + I) / KSTEPS) * 32 * WS + ((NV + I) % KSTEPS) * 32>(base)), ...);
}

// C: input channels, NSET: weight sets, TH x TW taps, S: input stride, WB: band columns (compile time), MAXCH: 16-byte chunks of a band's
// source rows per thread, NV: k-steps of set 0 whose fragments stay in VGPRs (the rest of set 0 and all other sets live in AGPRs),
// BITS: 0 none / 1 sign planes written / 2 sign planes read as the ReLU mask
template <int C, int NSET, int TH, int TW, int S, int WB, int MAXCH, int NV, int BITS>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void conv_band4_kernel(BandP p) {
    constexpr int NT = 256, NWAVE = 4;
    constexpr int K = TH * TW * C, KSTEPS = K / 16;
    constexpr int PS = C * 2 + 16;          // band pixel stride (bytes)
    constexpr int CPP = C / 8;              // 16-byte chunks per pixel
    constexpr int WH = S == 2 ? (WB + 1) / 2 : WB;          // columns of one parity plane
    constexpr int ROWB = (S == 2 ? 2 * WH : WB) * PS;       // LDS bytes of one band row
    constexpr int NA = NSET * KSTEPS - NV;                  // fragments in AGPRs
    static_assert(NV <= KSTEPS && NA * 4 + NSET * 16 <= 256, "the AGPR half holds the resident fragments and the accumulators");
    extern __shared__ __attribute__((aligned(16))) char band0[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const float inv_W = __builtin_amdgcn_rcpf((float)p.W), inv_OW = __builtin_amdgcn_rcpf((float)p.OWmax);
    const int bands = (p.OHmax + p.R - 1) / p.R;
    const int nunits = p.Nimg * bands;

    // the class fields the tile loop uses, in SGPRs (a scalar load inside the loop shares its counter with the LDS fragment reads)
    int c_OH[NSET], c_OW[NSET], c_co[NSET];
    long c_yoff[NSET];
#pragma unroll
    for (int s = 0; s < NSET; ++s) {
        c_OH[s] = __builtin_amdgcn_readfirstlane(p.cls[s].OH); c_OW[s] = __builtin_amdgcn_readfirstlane(p.cls[s].OW);
        c_co[s] = __builtin_amdgcn_readfirstlane(p.cls[s].co_base);
        c_yoff[s] = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)p.cls[s].y_off >> 32)) << 32) |
                           (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)p.cls[s].y_off));
    }

    // ---- band staging over SOURCE chunks (see the header)
    uint4 pre[MAXCH];
    auto band_rows = [&](int unit, int& n, int& r0, int& R, int& sy0, int& nchunk, int& brow0) {
        n = unit / bands; const int b = unit - n * bands; r0 = b * p.R; R = (r0 + p.R <= p.OHmax) ? p.R : p.OHmax - r0;
        const int rows = (R - 1) * S + TH, iy0 = r0 * S - p.pad_y;
        sy0 = iy0 > 0 ? iy0 : 0;
        const int sy1 = iy0 + rows < p.H ? iy0 + rows : p.H;
        nchunk = (sy1 - sy0) * p.W * CPP;
        brow0 = sy0 - iy0;
    };
    auto stage_load = [&](int unit) {
        int n, r0, R, sy0, nchunk, brow0; band_rows(unit, n, r0, R, sy0, nchunk, brow0);
        const long piece = (long)n * p.x_sn + (long)sy0 * p.x_sy;       // (elements; uniform)
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int ci = tid + j * NT;
            // (by value through band_load_bits: `pre[j] = pointer[i]` on HIP's uint4 is an aggregate copy that keeps the array in scratch)
            pre[j] = band_load_bits(p.X, piece + (ci < nchunk ? ci : 0) * 8);   // untouched until stage_store: an ALU use here is a wait in front of the MFMA loop
        }
    };
    auto stage_store = [&](int unit) {
        int n, r0, R, sy0, nchunk, brow0; band_rows(unit, n, r0, R, sy0, nchunk, brow0);
        int t2 = tid;
        asm volatile("" : "+v"(t2));                         // (opaque: the MAXCH addresses are not kept in registers across the MFMA loops)
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int ci = t2 + j * NT;
            const int spx = ci / CPP, c = ci % CPP;
            const int row = fast_div(spx, inv_W), col = spx - row * p.W;
            const int br = row + brow0, bc = col + p.pad_x;
            const int pos = S == 2 ? br * ROWB + ((bc & 1) * WH + (bc >> 1)) * PS : br * ROWB + bc * PS;
            if (ci < nchunk && bc < WB) *(uint4*)(band0 + pos + c * 16) = pre[j];
        }
    };

    // ---- prologue: the NSET x 32 x K weights once, coalesced, through the (still empty) band area; every wave then takes ALL of them
    constexpr int RPI = 64 / CPP;                            // weight rows one load instruction covers
    constexpr int WITEMS = (32 / RPI) * TH * TW;             // load instructions per weight set
    constexpr int NW = (NSET * WITEMS + NWAVE - 1) / NWAVE;  // ... per wave
    constexpr int WS = K * 2 + 16;                           // LDS bytes per weight row
    {
        uint4 wtmp[NW];
        const int wrow = lane / CPP, wc = lane % CPP;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int it = wave + i * NWAVE;                 // item = (set, row group, tap); wave-uniform
            const int itc = it < NSET * WITEMS ? it : NSET * WITEMS - 1;
            const int set = itc / WITEMS, rem = itc % WITEMS, rg = rem / (TH * TW), t = rem % (TH * TW);
            wtmp[i] = band_load_bits(p.Wt, (p.cls[set].w_row0 + rg * RPI + wrow) * p.ldw + p.cls[set].w_tap_off[t] + wc * 8);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int it = wave + i * NWAVE;
            if (it < NSET * WITEMS) {
                const int set = it / WITEMS, rem = it % WITEMS, rg = rem / (TH * TW), t = rem % (TH * TW);
                *(uint4*)(band0 + (set * 32 + rg * RPI + wrow) * WS + (t * C + wc * 8) * 2) = wtmp[i];
            }
        }
    }
    __shared__ float sbias[BAND_MAXCLS * 32];
    if (tid < NSET * 32) sbias[tid] = p.bias ? p.bias[p.cls[tid >> 5].co_base + (tid & 31)] : 0.f;
    __syncthreads();
    bf16x8_t wv[NV > 0 ? NV : 1];                            // set 0, k-steps [0, NV): VGPRs
    bf16x8_t wa[NA];                                         // everything else: AGPRs
#pragma unroll
    for (int ks = 0; ks < NV; ++ks) wv[ks] = *(const bf16x8_t*)(band0 + r * WS + (ks * 16 + h * 8) * 2);
    load_agpr_weights<NV, KSTEPS, WS>(wa, (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)band0 + r * WS + h * 16,
                                      std::make_integer_sequence<int, NA>{});
    lds_wait();
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the resident operands are complete before the band area is overwritten
    __syncthreads();
    int unit = blockIdx.x;
    if (unit < nunits) stage_load(unit);
    if (p.pad_y | p.pad_x) {                                 // zero padding: once (the launcher admits padded bands as whole frames only)
        for (int o = tid * 16; o < p.lds_band; o += NT * 16) *(uint4*)(band0 + o) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }
    if (unit < nunits) stage_store(unit);
    __syncthreads();

    auto wfrag = [&](int s, int ks) -> bf16x8_t { return (s == 0 && ks < NV) ? wv[ks < NV ? ks : 0] : wa[s * KSTEPS + ks - NV < 0 ? 0 : s * KSTEPS + ks - NV]; };

    // ---- the tile loop as ONE stream of MFMA blocks.  With a single wave per SIMD nothing else hides a wave's own epilogue, so the weight sets
    // are two GROUPs that alternate: block (tile, group g) = KSTEPS x GROUP MFMAs on group g's accumulators, and between its MFMAs — one slice per
    // k-step, SKEW k-steps behind the previous block's last MFMA — the wave converts, masks and stores the accumulators of the OTHER group (the
    // previous block: same tile for g = 1, the previous tile for g = 0; across unit boundaries too: the outputs do not depend on the band) and
    // re-initialises them with the bias for the block after this one.
    constexpr int GROUP = NSET / 2, NG = 2;
    constexpr int NSTEP = GROUP * 7, SKEW = 2;               // slices per pending block: 4 conversions, 2 stores, 1 re-initialisation per set
    static_assert(NSET % 2 == 0 && NSTEP + SKEW <= KSTEPS, "the other group's epilogue fits between the MFMAs of a block");
    f32x16_t acc[NSET];
    constexpr bool HASBIAS = BITS != 2;                      // (the data gradients carry no bias: their accumulators start from zero)
    auto acc_init = [&](int s) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (HASBIAS) b4 = *(const float4*)(sbias + s * 32 + 8 * g + 4 * h);      // (zeros when the launch has no bias)
            acc[s][4 * g] = b4.x; acc[s][4 * g + 1] = b4.y; acc[s][4 * g + 2] = b4.z; acc[s][4 * g + 3] = b4.w;
        }
    };
#pragma unroll
    for (int s = 0; s < NSET; ++s) acc_init(s);
    // context of the pending block's sets: output offset, does the lane own the position, sign-plane word
    long cur_off[NSET], prev_off[GROUP];
    bool cur_lv[NSET], prev_lv[GROUP];
    unsigned cur_mb[NSET], prev_mb[GROUP];
#pragma unroll
    for (int i = 0; i < GROUP; ++i) { prev_off[i] = 0; prev_lv[i] = false; prev_mb[i] = 0; }   // (nothing is pending in front of the first block: its slices run on unowned positions)
    uint2 pk[GROUP][4];
    unsigned mb_out[GROUP];
    const uint32_t floor2 = p.relu ? 0u : 0x80008000u;

    // slice m of the epilogue of the sets sb .. sb + GROUP - 1 with context (off, lv, mb): compile-time m, sb
    auto epi_slice = [&](int m, int sb, const long* off, const bool* lv, const unsigned* mbi) {
        const int i = m / 7, j = m % 7, s = sb + i;
        if (j < 4) {                                         // registers 4j..4j+3 = channels co_base + 8j + 4h + {0..3}
            pk[i][j] = make_uint2(max_s16x2(pack_bf16x2(acc[s][4 * j], acc[s][4 * j + 1]), floor2),
                                  max_s16x2(pack_bf16x2(acc[s][4 * j + 2], acc[s][4 * j + 3]), floor2));
        } else if (j < 6) {
            const int gp = j - 4;
            const auto sx = __builtin_amdgcn_permlane32_swap(pk[i][2 * gp].x, pk[i][2 * gp + 1].x, false, false);
            const auto sy = __builtin_amdgcn_permlane32_swap(pk[i][2 * gp].y, pk[i][2 * gp + 1].y, false, false);
            uint32_t o[4] = {sx[0], sy[0], sx[1], sy[1]};                 // channels co_base + 16 gp + 8 h + {0..7}
            if (BITS == 2) {
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = keep_u16x2(o[e], (mbi[i] >> (16 * gp + 8 * h + 2 * e)) & 3u);
            }
            if (BITS == 1) {
                const unsigned a = nonzero_u16x2(o[0]) | (nonzero_u16x2(o[1]) << 2) | (nonzero_u16x2(o[2]) << 4) | (nonzero_u16x2(o[3]) << 6);
                const unsigned w = ((a | (a >> 15)) & 0xffu) << (16 * gp);
                mb_out[i] = gp == 0 ? w : (mb_out[i] | w);
            }
            if (lv[i] && !(p.dbg & 2)) *(uint4*)((uint16_t*)p.Y + off[i] + 16 * gp + 8 * h) = make_uint4(o[0], o[1], o[2], o[3]);
            if (BITS == 1 && gp == 1) {
                unsigned w = mb_out[i] << (8 * h);
                const auto other = __builtin_amdgcn_permlane32_swap(w, w, false, false);
                w |= other[1];
                if (lv[i] && h == 0) p.bits_out[(long)(c_co[s] >> 5) * p.bplane + ((off[i] - c_co[s]) >> p.bshift)] = w;
            }
        } else {                                             // the set's accumulators start the block after this one from the bias
            acc_init(s);
        }
    };

    for (; unit < nunits; unit += gridDim.x) {
        const int next = unit + gridDim.x;
        const bool have_next = next < nunits && !(p.dbg & 4);
        if (have_next && !(p.dbg & 64)) stage_load(next);    // in flight during the MFMA blocks below

        int n, r0, R, sy0_, nchunk_, brow0_; band_rows(unit, n, r0, R, sy0_, nchunk_, brow0_);
        const int npix = R * p.OWmax;
        const int ntile = (npix + 31) / 32;
        for (int tile = wave; tile < ntile; tile += NWAVE) {
            int q = tile * 32 + r;
            const bool live = q < npix;
            if (!live) q = npix - 1;
            const int oy = fast_div(q, inv_OW), ox = q - oy * p.OWmax;
            const char* a0 = band0 + (oy * S) * ROWB + ox * PS + h * 16;        // (stride 2: parity plane 0, column ox = band column 2 ox)
#pragma unroll
            for (int s = 0; s < NSET; ++s) {
                cur_lv[s] = live && r0 + oy < c_OH[s] && ox < c_OW[s];
                const int oyc = cur_lv[s] ? r0 + oy : 0, oxc = cur_lv[s] ? ox : 0;   // (a position the class does not own: any valid address)
                cur_off[s] = c_yoff[s] + (long)n * p.y_sn + (long)oyc * p.y_sy + (long)oxc * p.y_sx + c_co[s];
                cur_mb[s] = 0;
                if (BITS == 2) cur_mb[s] = p.bits_in[(long)(c_co[s] >> 5) * p.bplane + ((cur_off[s] - c_co[s]) >> p.bshift)];
            }
            constexpr int RD = (KSTEPS > 32 || NSET > 2) ? 6 : 8;   // fragment reads in flight (the 288-register weight sets leave room for six)
            auto frag = [&](int ks) {
                const int k0 = ks * 16;
                const int t = k0 / C, c0 = k0 % C;
                const int ty = t / TW, tx = t % TW;
                const int toff = S == 2 ? ty * ROWB + ((tx & 1) * WH + (tx >> 1)) * PS : ty * ROWB + tx * PS;
                return *(const bf16x8_t*)(a0 + toff + c0 * 2);
            };
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                bf16x8_t pf[RD];
#pragma unroll
                for (int i = 0; i < RD; ++i) pf[i] = frag(i);
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ++ks) {
                    const bf16x8_t px = pf[ks % RD];
#pragma unroll
                    for (int i = 0; i < GROUP; ++i)
                        acc[g * GROUP + i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag(g * GROUP + i, ks), px, acc[g * GROUP + i], 0, 0, 0);   // D[channel][pixel]
                    __builtin_amdgcn_sched_barrier(0);
                    if (ks + RD < KSTEPS) pf[ks % RD] = frag(ks + RD);
                    if (ks >= SKEW && ks - SKEW < NSTEP) {
                        if (g == 0) epi_slice(ks - SKEW, (NG - 1) * GROUP, prev_off, prev_lv, prev_mb);               // the previous tile's last group
                        else epi_slice(ks - SKEW, (g - 1) * GROUP, cur_off + (g - 1) * GROUP, cur_lv + (g - 1) * GROUP, cur_mb + (g - 1) * GROUP);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
#pragma unroll
            for (int i = 0; i < GROUP; ++i) {
                prev_off[i] = cur_off[(NG - 1) * GROUP + i]; prev_lv[i] = cur_lv[(NG - 1) * GROUP + i]; prev_mb[i] = cur_mb[(NG - 1) * GROUP + i];
            }
        }
        __syncthreads();                                     // every wave is done reading this band
        if (have_next && !(p.dbg & 32)) stage_store(next);
        if (p.dbg & 32) {                                    // (time split: the loads are waited for, nothing is written)
#pragma unroll
            for (int j = 0; j < MAXCH; ++j) asm volatile("" :: "v"(pre[j].x));
        }
        __syncthreads();
    }
    // drain: the last block's epilogue
#pragma unroll
    for (int m = 0; m < NSTEP; ++m)
        if (m % 7 != 6) epi_slice(m, (NG - 1) * GROUP, prev_off, prev_lv, prev_mb);
}

template <int C, int NSET, int TH, int TW, int S, int WB, int MAXCH, int NV, int BITS>
int launch4(BandP& p, hipStream_t s) {
    constexpr int PS = C * 2 + 16, CPP = C / 8, K = TH * TW * C;
    constexpr int Wrow = S == 2 ? 2 * ((WB + 1) / 2) : WB;   // pixels of one LDS band row
    if ((p.OWmax - 1) * S + TW != WB) return -1;
    if (BITS == 2 && p.bias) return -1;                      // (the sign-plane-masked instances are the data gradients: no bias)
    // dense NHWC rows: a band's source rows are one contiguous piece
    if (p.x_sx != C || p.x_sy != (long)p.W * C || ((uintptr_t)p.X % 16) != 0 || (p.x_sn * 2) % 16 != 0) return -1;
    const long budget = (160 * 1024 - 1024) / 16 * 16;       // (512 B of static LDS: the bias table)
    auto rows_of = [&](int rr) { return (long)(rr - 1) * S + TH; };
    auto chunks_of = [&](int rr) { const long rows = rows_of(rr) < p.H ? rows_of(rr) : p.H; return rows * p.W * CPP; };
    int R = p.OHmax;
    while (R > 1 && (rows_of(R) * Wrow * PS > budget || chunks_of(R) > (long)MAXCH * 256)) --R;
    if (rows_of(R) * Wrow * PS > budget || chunks_of(R) > (long)MAXCH * 256) return -1;
    const int bands = (p.OHmax + R - 1) / R;
    if (bands > 1 && (p.pad_y || p.pad_x)) return -1;        // (zero padding is written once per launch: padded bands are whole frames)
    R = (p.OHmax + bands - 1) / bands;                       // equal-ish bands
    p.R = R; p.F = 1;
    if ((long)R * p.OWmax < 64) return -1;                   // a unit that cannot feed 4 waves
    const int nunits = p.Nimg * bands;
    size_t lds = (size_t)rows_of(R) * Wrow * PS;
    const size_t wbytes = (size_t)NSET * 32 * (K * 2 + 16);  // the prologue parks the weights in the band area
    p.lds_band = (int)((lds + 15) / 16 * 16);                // (what the padding pass zeroes)
    if (lds < wbytes) lds = wbytes;
    lds = (lds + 15) / 16 * 16;
    const int per = (nunits + 255) / 256, grid = (nunits + per - 1) / per;
    auto kern = conv_band4_kernel<C, NSET, TH, TW, S, WB, MAXCH, NV, BITS>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -2;
        attr_set = true;
    }
    kern<<<grid, 256, lds, s>>>(p);
    return 0;
}

}  // namespace

namespace hulc_band {

int launch_band4(BandP& p, int C, int NSET, int TH, int TW, int S, hipStream_t s) {
    // covered: bf16 tensors, no residual, no activation-tensor mask (sign planes only), the static camera's frame-sized maps (band width is a
    // template parameter; the small gripper maps pack several frames into a unit on conv_band.hip)
    if (p.x_dtype != HULC_BF16 || p.y_dtype != HULC_BF16 || p.w_dtype != HULC_BF16 || p.add || (p.mask && !p.bits_in) || p.dbg) return -1;
    { static const char* e = getenv("HULC_BAND4_DBG"); p.dbg = e ? atoi(e) : 0; }     // time splits: 2 no output stores, 4 no staging of later units, 32 loads waited for but not written to LDS, 64 LDS writes without loads; 8 say which launches were taken
    if (p.dbg & 8) fprintf(stderr, "[band4] C=%d NSET=%d %dx%d S=%d pad=%d,%d OHmax=%d OWmax=%d H=%d W=%d bits=%d\n", C, NSET, TH, TW, S, p.pad_y, p.pad_x, p.OHmax, p.OWmax, p.H, p.W, p.bits_out ? 1 : (p.bits_in ? 2 : 0));
    const int bits = p.bits_out ? 1 : (p.bits_in ? 2 : 0);
    if (C == 32 && NSET == 2 && TH == 4 && TW == 4 && S == 2 && !p.pad_y && !p.pad_x) {          // conv2 forward: 49 x 49 x 32 -> 23 x 23 x 64
        if (bits == 1) return launch4<32, 2, 4, 4, 2, 48, 20, 16, 1>(p, s);
        if (bits == 0) return launch4<32, 2, 4, 4, 2, 48, 20, 16, 0>(p, s);
        return -1;
    }
    if (C == 64 && NSET == 2 && TH == 3 && TW == 3 && S == 1) {
        if (bits == 0 && !p.pad_y && !p.pad_x) return launch4<64, 2, 3, 3, 1, 23, 17, 18, 0>(p, s);                  // conv3 forward: 23 x 23 -> 21 x 21
        if (bits == 2 && p.pad_y == 2 && p.pad_x == 2) return launch4<64, 2, 3, 3, 1, 25, 14, 18, 2>(p, s);          // conv3 data gradient: 21 x 21 -> 23 x 23
        return -1;
    }
    if (C == 64 && NSET == 4 && TH == 2 && TW == 2 && S == 1 && p.pad_y == 1 && p.pad_x == 1) {   // conv2 data gradient: 23 x 23 x 64 -> 4 classes of 25 x 25 x 32
        if (bits == 2) return launch4<64, 4, 2, 2, 1, 26, 17, 16, 2>(p, s);
        return -1;
    }
    return -1;
}

}  // namespace hulc_band
