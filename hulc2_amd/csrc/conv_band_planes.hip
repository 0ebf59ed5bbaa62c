// conv_band_planes.hip — the direct-to-LDS band convolution with a CHUNK-MAJOR band (round 6).
//
// reference arithmetic: nn.Conv2d(+ReLU) of hulc2/models/perceptual_encoders/vision_network.py:41-46 (conv3 of the static camera) and
// autograd's conv2d input gradient of conv3 / conv2.
//
// conv_band_glds_kernel (conv_band.hip) fills its band pixel-major (128-byte pixels, no padding possible: a direct load writes 1 KB of
// consecutive LDS) and spreads the fragment reads over the banks with an XOR swizzle on the source address — which every fragment read
// then has to undo: ~3 vector instructions per MFMA in a loop whose MFMAs and VALU share the SIMD's issue slot (252 MFMAs + ~1640 VALU per
// SIMD and frame in conv3's forward: 4.4 + 3 us of the 8.8 us a unit takes).  tools/probe/mfma_lds_probe.hip: the LDS itself is not what
// such a loop waits for.  First attempt (eight whole-frame planes, one per 16-byte chunk: fragment reads = base + immediate, ZERO VALU in the
// loop): no faster — a direct load then takes 64 pixels of one chunk, 64 cache lines per instruction, and the phase stamps (HULC_BAND_STAMPS)
// showed the second wave of every SIMD spending 6 300 of a unit's 15 400 cycles ISSUING its nine loads (address processing, 64 lines each).
// What ships in this file: blocks of 8 pixels, chunk-major INSIDE the 1 KB block —
//   * a direct load = the 1 KB of 8 consecutive pixels, 8 cache lines, as in the pixel-major kernel;
//   * slot s of block b holds chunk s ^ (b & 1) (picked on the source side): the 16 lanes of a read group — two blocks, one chunk — use both
//     halves of the bank row: conflict-free;
//   * the address of chunk 2 kc + h at tap (ty, tx) = per-lane TAP base + kc * 256 (immediate): ~6 VALU per tap and tile instead of ~5 per tap
//     + 2 per k-step;
//   * zero padding (data gradients) = ONE zero pixel behind the frame; a tap outside the frame is redirected when its base is formed.
// Everything else — two static LDS bands, one raw barrier per unit, weights parked through the band arrays in the prologue, sign planes fetched
// a unit ahead, packed-word epilogue — is conv_band_glds_kernel's.  bf16 in / out, whole frames per unit, 64 input channels.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "conv_band.h"
#include <stdlib.h>

using namespace hulc_band;

namespace {

constexpr int PL_BAND_BYTES = 69 * 1024;       // 8 planes x (23 x 23 + 1) pixels x 16 B = 67 840 B, rounded up

// NSET weight sets (32 output channels each), TH x TW taps, HIN x WIN input frame (compile time), BITS = 2: sign-plane words mask the result,
// PAD: taps may fall outside the frame (data gradients)
template <int NSET, int TH, int TW, int HIN, int WIN, int BITS, bool PAD, bool STAMP = false>
__global__ __launch_bounds__(512) void conv_band_planes_kernel(BandP p, unsigned long long* stamps = nullptr) {
    constexpr int C = 64, CPP = 8;
    constexpr int K = TH * TW * C, KSTEPS = K / 16, WPS = 8 / NSET;
    constexpr int SPA = NSET / 2;                           // weight sets parked per band array in the prologue
    constexpr int P = HIN * WIN;                            // pixels of a frame; pixel P is the zero pixel
    constexpr int NBLK = (P + 1 + 7) / 8;                   // 1 KB blocks of 8 pixels (the zero pixel included) = direct-load instructions per band
    static_assert(NBLK * 1024 <= PL_BAND_BYTES, "a frame fits a band");
    __shared__ __attribute__((aligned(2048))) float sbias[BAND_MAXCLS * 32 + 384];
    __shared__ __attribute__((aligned(256))) char bandA_[PL_BAND_BYTES];
    __shared__ __attribute__((aligned(256))) char bandB_[PL_BAND_BYTES];
    constexpr int MAXTW = BITS == 2 ? (NSET == 2 ? 6 : 12) : 1;             // tiles of one wave per unit that carry a sign-plane word
    __shared__ unsigned smask[8][MAXTW][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int set = wave % NSET, part = wave / NSET;
    const BandCls& cl = p.cls[set];
    const int cl_OH = cl.OH, cl_OW = cl.OW, cl_co = cl.co_base;
    const long cl_yoff = cl.y_off;
    const float inv_OW = __builtin_amdgcn_rcpf((float)cl_OW);
    const int nunits = p.Nimg;
    const int npix = cl_OH * cl_OW, ntile = (npix + 31) / 32;

    // ---- prologue: weights once through LDS (as conv_band_kernel), SPA sets per (still empty) band array
    constexpr int RPI = 64 / CPP, WITEMS = (32 / RPI) * TH * TW, NW = (WITEMS + WPS - 1) / WPS, WS = K * 2 + 16;
    static_assert((long)SPA * 32 * WS <= PL_BAND_BYTES, "the parked weights fit a band array");
    char* const wpark = ((set / SPA) ? bandB_ : bandA_) + (set % SPA) * 32 * WS;
    uint4 wtmp[NW];
    {
        const int wrow = lane / CPP, wc = lane % CPP;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int it = part + i * WPS, itc = it < WITEMS ? it : WITEMS - 1;
            const int rg = itc / (TH * TW), t = itc % (TH * TW);
            wtmp[i] = band_load_bits(p.Wt, (cl.w_row0 + rg * RPI + wrow) * p.ldw + cl.w_tap_off[t] + wc * 8);
        }
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int it = part + i * WPS;
            if (it < WITEMS) {
                const int rg = it / (TH * TW), t = it % (TH * TW);
                *(uint4*)(wpark + (rg * RPI + wrow) * WS + (t * C + wc * 8) * 2) = wtmp[i];
            }
        }
    }
    if (tid < NSET * 32) sbias[tid] = p.bias ? p.bias[p.cls[tid >> 5].co_base + (tid & 31)] : 0.f;
    __syncthreads();
    bf16x8_t wfrag[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) wfrag[ks] = *(const bf16x8_t*)(wpark + r * WS + (ks * 16 + h * 8) * 2);
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
    __syncthreads();
    // band layout: blocks of 8 pixels, 1 KB each; inside a block CHUNK-major — slot s (128 bytes) holds, for the block's 8 pixels, 16-byte chunk
    // s ^ (block & 1) — so that (a) one direct-load instruction = the 1 KB of 8 consecutive pixels in memory (fully coalesced: 8 cache lines) and
    // (b) the 16 lanes of a fragment-read group (two blocks, same chunk) fall on the two different halves of the 256-byte bank row.
    // chunk c = 2 kc + h of pixel q:  (q >> 3) * 1024 + ((c ^ ((q >> 3) & 1)) << 7) + (q & 7) * 16  =  tap base (per lane and tap) + kc * 256
    if (PAD && tid < 16) {                                   // the zero pixels (never written again: the last block's load stops in front of them)
        char* b = tid < 8 ? bandA_ : bandB_;
        *(uint4*)(b + (P >> 3) * 1024 + ((tid & 7) << 7) + (P & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);
    }

    // direct loads of one frame: block i by wave i % 8; lane = (slot, pixel of the block), the slot's chunk picked on the SOURCE side
    auto glds_band = [&](int unit, char* band) {
        const uint16_t* frame = (const uint16_t*)p.X + (long)unit * p.x_sn;
        const int slot = lane >> 3, pj = lane & 7;
        for (int i = wave; i < NBLK; i += 8) {
            const int q = i * 8 + pj;
            if (q < P)
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(frame + (long)q * C + ((slot ^ (i & 1)) << 3)),
                                                 (void __attribute__((address_space(3)))*)(band + i * 1024), 16, 0, 0);
        }
    };
    // one of them (block i) — for the variant that spreads a wave's loads over its tile loop (SPREAD)
    auto glds_one = [&](int unit, char* band, int i) {
        const uint16_t* frame = (const uint16_t*)p.X + (long)unit * p.x_sn;
        const int slot = lane >> 3, pj = lane & 7, q = i * 8 + pj;
        if (i < NBLK && q < P)
            __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(frame + (long)q * C + ((slot ^ (i & 1)) << 3)),
                                             (void __attribute__((address_space(3)))*)(band + i * 1024), 16, 0, 0);
    };
    unsigned mbn[MAXTW];
    auto mask_park = [&]() {
        if (BITS == 2 && lane < 32) {
#pragma unroll
            for (int i = 0; i < MAXTW; ++i) smask[wave][i][lane] = mbn[i];
        }
    };
    // sign-plane word of a tile's pixel: plane base + (pixel offset inside the frame, the same for every unit: formed ONCE) + the unit's frame offset
    int moff[MAXTW];
    if (BITS == 2) {
#pragma unroll
        for (int i = 0; i < MAXTW; ++i) {
            const int tile = part + i * WPS;
            int q = tile * 32 + r;
            q = q < npix ? q : npix - 1;
            if (tile >= ntile) q = 0;
            const int oy = fast_div(q, inv_OW), ox = q - oy * cl_OW;
            moff[i] = (int)((cl_yoff + (long)oy * p.y_sy + (long)ox * p.y_sx) >> p.bshift);
        }
    }
    const unsigned* const mplane = BITS == 2 ? p.bits_in + (long)(cl_co >> 5) * p.bplane : nullptr;
    const long unit_words = p.y_sn >> p.bshift;              // (the launcher's caller checked y_sn % channels == 0)
    auto mask_fetch = [&](int unit) {
        const unsigned* mp = mplane + (long)unit * unit_words;
#pragma unroll
        for (int i = 0; i < MAXTW; ++i) mbn[i] = mp[moff[i]];
    };

    int unit = blockIdx.x;
    if (BITS == 2 && unit < nunits) mask_fetch(unit);
    if (unit < nunits) glds_band(unit, bandA_);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0) — the BUILTIN (the compiler's wait-count bookkeeping sees it)
    mask_park();
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the zero pixels are in LDS before anybody passes the (raw) barrier
    __builtin_amdgcn_s_barrier();
    unsigned long long t_issue = 0, t_tiles = 0, t_wait = 0, t_bar = 0, t_units = 0;      // (STAMP: per-wave cycle sums of a unit's four phases)
    auto do_unit = [&](int unit, const char* __restrict__ band, char* __restrict__ band_next) {
        const int next = unit + gridDim.x;
        unsigned long long c0 = 0, c1 = 0, c2 = 0, c3 = 0;
        if (STAMP) c0 = __builtin_readcyclecounter();
        if (BITS == 2 && next < nunits) mask_fetch(next);         // requested BEFORE the direct loads, parked behind the closing wait, used a unit later
        const bool spread = (p.dbg & 256) != 0;               // (HULC_BAND_PLANES_SPREAD: a wave's direct loads between its tiles instead of up front)
        if (next < nunits && !spread) glds_band(next, band_next);        // lands while this unit is multiplied
        if (STAMP) c1 = __builtin_readcyclecounter();

        auto do_tile = [&](int tile, unsigned mb_in) {
            int q = tile * 32 + r;
            const bool live = q < npix;
            if (!live) q = npix - 1;
            const int oy = fast_div(q, inv_OW), ox = q - oy * cl_OW;
            // band position of tap (ty, tx): pixel (oy + ty - pad_y) * WIN + (ox + tx - pad_x), or the zero pixel outside the frame
            const int q0 = (oy - p.pad_y) * WIN + (ox - p.pad_x);
            const char* tap_base[TH * TW];
#pragma unroll
            for (int ty = 0; ty < TH; ++ty)
#pragma unroll
                for (int tx = 0; tx < TW; ++tx) {
                    int qt = q0 + ty * WIN + tx;
                    if (PAD) {
                        const int iy = oy + ty - p.pad_y, ix = ox + tx - p.pad_x;
                        qt = (iy >= 0 && iy < HIN && ix >= 0 && ix < WIN) ? qt : P;
                    }
                    const int blk = qt >> 3;
                    tap_base[ty * TW + tx] = band + blk * 1024 + ((h ^ (blk & 1)) << 7) + ((qt & 7) << 4);
                }
            f32x16_t acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(sbias + set * 32 + 8 * g + 4 * h);
                acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
            }
            constexpr int RD = 8;
            auto frag = [&](int ks) {
                const int k0 = ks * 16, t = k0 / C, kc = (k0 % C) / 16;
                return *(const bf16x8_t*)(tap_base[t] + kc * 256);
            };
            bf16x8_t pf[RD];
#pragma unroll
            for (int i = 0; i < RD; ++i) pf[i] = frag(i);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8_t px = pf[ks % RD];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[ks], px, acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks + RD < KSTEPS) pf[ks % RD] = frag(ks + RD);
                __builtin_amdgcn_sched_barrier(0);
            }
            const long off0 = cl_yoff + (long)unit * p.y_sn + (long)oy * p.y_sy + (long)ox * p.y_sx + cl_co;
            if (BITS == 0 && p.Y16 && p.y_dtype == HULC_F32 && live) {
                // (hulc_conv_desc.y_bf16, precision site "a3") the exact map next to the bf16 one: Y is the fp32 tensor, Y16 the bf16 one
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
                    if (p.relu) { v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f); }
                    *(float4*)((float*)p.Y + off0 + 8 * g + 4 * h) = v;
                }
            }
            if (BITS == 0 && p.Y16 && p.y_dtype == HULC_F16) {
                // the fp16 twin (11 bits of mantissa at the bf16 map's two bytes: measured, the fp32 twin's 16-byte pieces at 256-byte pixel
                // pitch cost this launch 48 us per 2048 frames): packed and exchanged between the lane halves exactly like the bf16 map below
                auto pkh = [&](float a, float b) -> uint32_t {
                    if (p.relu) { a = fmaxf(a, 0.f); b = fmaxf(b, 0.f); }
                    union { _Float16 hh[2]; uint32_t u; } r; r.hh[0] = (_Float16)a; r.hh[1] = (_Float16)b;
                    return r.u;
                };
                uint2 ph[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) ph[g] = make_uint2(pkh(acc[4 * g], acc[4 * g + 1]), pkh(acc[4 * g + 2], acc[4 * g + 3]));
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const auto sx = __builtin_amdgcn_permlane32_swap(ph[2 * gp].x, ph[2 * gp + 1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(ph[2 * gp].y, ph[2 * gp + 1].y, false, false);
                    if (live) *(uint4*)((uint16_t*)p.Y + off0 + 16 * gp + 8 * h) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                }
            }
            uint16_t* const ybf = (uint16_t*)((BITS == 0 && p.Y16) ? p.Y16 : p.Y);
            uint2 pk[4];
            const uint32_t floor2 = p.relu ? 0u : 0x80008000u;  // (ReLU on the packed words: see conv_band_kernel's epilogue)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                pk[g] = make_uint2(max_s16x2(pack_bf16x2(acc[4 * g], acc[4 * g + 1]), floor2), max_s16x2(pack_bf16x2(acc[4 * g + 2], acc[4 * g + 3]), floor2));
#pragma unroll
            for (int gp = 0; gp < 2; ++gp) {
                const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * gp].x, pk[2 * gp + 1].x, false, false);
                const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * gp].y, pk[2 * gp + 1].y, false, false);
                uint32_t o[4] = {sx[0], sy[0], sx[1], sy[1]};                 // channels co_base + 16 gp + 8 h + {0..7}
                if (BITS == 2) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) o[e] = keep_u16x2(o[e], (mb_in >> (16 * gp + 8 * h + 2 * e)) & 3u);
                }
                if (live) *(uint4*)(ybf + off0 + 16 * gp + 8 * h) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        };
        int ti = 0, li = wave;                                // li: this wave's next block to load
        constexpr int PER_TILE = 2;                           // (a wave has ~NBLK / 8 = 7-9 blocks and 5-10 tiles per unit)
        for (int tile = part; tile < ntile; tile += WPS, ++ti) {
            if (spread && next < nunits) {
#pragma unroll
                for (int k = 0; k < PER_TILE; ++k, li += 8) glds_one(next, band_next, li);
            }
            do_tile(tile, BITS == 2 ? smask[wave][ti][r] : 0u);
        }
        if (spread && next < nunits)
            for (; li < NBLK; li += 8) glds_one(next, band_next, li);
        if (STAMP) c2 = __builtin_readcyclecounter();
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the next band has landed (and this unit's stores are acknowledged)
        if (STAMP) c3 = __builtin_readcyclecounter();
        if (next < nunits) mask_park();
        __builtin_amdgcn_s_barrier();                          // every wave is done reading this band
        if (STAMP) {
            const unsigned long long c4 = __builtin_readcyclecounter();
            t_issue += c1 - c0; t_tiles += c2 - c1; t_wait += c3 - c2; t_bar += c4 - c3; t_units += 1;
        }
    };
    for (; unit < nunits; unit += 2 * gridDim.x) {
        do_unit(unit, bandA_, bandB_);
        if (unit + (int)gridDim.x < nunits) do_unit(unit + gridDim.x, bandB_, bandA_);
    }
    if (STAMP && stamps && lane == 0) {
        unsigned long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 5;
        o[0] = t_issue; o[1] = t_tiles; o[2] = t_wait; o[3] = t_bar; o[4] = t_units;
    }
}

template <int NSET, int TH, int TW, int HIN, int WIN, int BITS, bool PAD>
int launch_planes(BandP& p, hipStream_t s) {
    const long npix = (long)p.OHmax * p.OWmax;
    if (BITS == 2 && (npix + 31) / 32 > (long)(NSET == 2 ? 6 : 12) * (8 / NSET)) return -1;      // the per-wave sign-word registers cover a unit's tiles
    const int nunits = p.Nimg, per = (nunits + 255) / 256, grid = (nunits + per - 1) / per;
    // a wave's direct loads between its tiles instead of up front: conv3's forward 70 vs 75 us (the ~250-cycle issue of each load then falls
    // behind MFMAs that are in flight), conv3's data gradient the same either way, conv2's 180 vs 173: default for the unpadded geometry only
    { const char* sp = getenv("HULC_BAND_PLANES_SPREAD"); p.dbg = (sp ? atoi(sp) != 0 : !PAD) ? 256 : 0; }
    // HULC_BAND_STAMPS=<device address of 256 x 8 x 5 uint64>: the instrumented instance leaves, per workgroup and wave, the cycle sums of a
    // unit's phases (issue of the next band's loads | tile loop | wait for loads + store acknowledgements | barrier) and the unit count
    const char* se = getenv("HULC_BAND_STAMPS");
    if (se && *se) {
        conv_band_planes_kernel<NSET, TH, TW, HIN, WIN, BITS, PAD, true><<<grid, 512, 0, s>>>(p, (unsigned long long*)strtoull(se, nullptr, 0));
        return 0;
    }
    conv_band_planes_kernel<NSET, TH, TW, HIN, WIN, BITS, PAD><<<grid, 512, 0, s>>>(p);
    return 0;
}

}  // namespace

namespace hulc_band {

// 0 = launched, -1 = not covered (the caller goes on to the other band kernels)
int launch_band_planes(BandP& p, int NSET, int TH, int TW, hipStream_t s) {
    // (an fp32 Y goes with its bf16 twin Y16 — hulc_conv_desc.y_bf16 — on the forward instances)
    const bool y_ok = (p.y_dtype == HULC_BF16 && !p.Y16) || ((p.y_dtype == HULC_F32 || p.y_dtype == HULC_F16) && p.Y16 && !p.bits_in);
    if (p.x_dtype != HULC_BF16 || !y_ok || p.w_dtype != HULC_BF16 || p.add || p.bits_out || (p.mask && !p.bits_in) || p.dbg) return -1;
    if (p.x_sx != 64 || p.x_sy != (long)p.W * 64 || p.x_sn != (long)p.H * p.W * 64 || ((uintptr_t)p.X % 16) != 0) return -1;
    const bool padded = p.pad_y != 0 || p.pad_x != 0;
    if (NSET == 2 && TH == 3 && TW == 3 && !padded && !p.bits_in && p.H == 23 && p.W == 23)                          // conv3 forward
        return launch_planes<2, 3, 3, 23, 23, 0, false>(p, s);
    if (NSET == 2 && TH == 3 && TW == 3 && p.pad_y == 2 && p.pad_x == 2 && p.bits_in && p.H == 21 && p.W == 21)      // conv3 data gradient
        return launch_planes<2, 3, 3, 21, 21, 2, true>(p, s);
    if (NSET == 4 && TH == 2 && TW == 2 && p.pad_y == 1 && p.pad_x == 1 && p.bits_in && p.H == 23 && p.W == 23)      // conv2 data gradient
        return launch_planes<4, 2, 2, 23, 23, 2, true>(p, s);
    return -1;
}

}  // namespace hulc_band
