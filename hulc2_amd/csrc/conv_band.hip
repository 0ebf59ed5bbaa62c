// conv_band.hip — LDS-band convolution for the NHWC bf16 layers (conv2 / conv3 forward and every data gradient).
//
// reference arithmetic: nn.Conv2d(+ReLU) of hulc2/models/perceptual_encoders/vision_network.py:41-46 and
// vision_network_gripper.py:15-19, and autograd's conv2d input gradient.
//
// The gather kernel in conv.hip re-reads every input element KH*KW/s^2 times through L1 in 16-byte pieces.  Here:
//   * weight-stationary: each of the 8 waves keeps ONE 32-row weight tile (32 output channels x K) in registers for the
//     whole launch (144 VGPRs at K = 576) — "weight sets" are the output-channel tiles of a forward conv, or the
//     stride^2 parity classes of a data gradient (each class = a dense stride-1 correlation with its own taps);
//   * the band of input rows a work unit (frame x band of output rows) needs is staged ONCE into LDS, coalesced, zero
//     padding included; the NEXT unit's band is prefetched into registers before the MFMA loop and written to LDS after
//     it, so HBM latency and the all-workgroups-at-once bandwidth burst hide under compute with a single LDS buffer;
//   * the k-loop has no global loads and no barriers: one ds_read_b128 per MFMA (pixel fragment straight out of the band,
//     pixel stride padded by 16 B -> conflict-free);
//   * MFMA roles are swapped (A = weights, B = pixels) so D[channel][pixel] leaves every lane with ONE pixel and groups of
//     four consecutive channels: the epilogue is 8-byte (bf16) / 16-byte (fp32) stores, no LDS transpose, no divisions.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "conv_band.h"
#include <stdlib.h>

using namespace hulc_band;

namespace {

// C: input channels, NSET: weight sets (32 output channels each), TH x TW taps, S: input stride, MAXCH: band chunks/thread
// BITS: 0 = no sign planes, 1 = written from the epilogue (forward), 2 = read as the ReLU mask (data gradient) — compile-time: the kernel sits
// at the 256-VGPR limit and a run-time switch cost every instance 20-60 bytes of scratch per lane
// DB: two LDS bands (round 4).  A unit's band is written while the PREVIOUS unit is still being multiplied — by every wave between the MFMA
// loop and the epilogue of one of its tiles, i.e. before that tile's stores: the wait the compiler puts in front of the LDS write (vmcnt(0):
// vector memory returns in order and the stores of the tile loop are counted with the loads) then covers the prefetch and stores that are a
// tile old, not the write acknowledgements of the stores just issued (single band: +1.4 us per unit, HULC_BAND_DBG=2) — and one workgroup
// barrier per unit instead of two.
template <int C, int NSET, int TH, int TW, int S, int MAXCH, bool MULTI, bool XF32, int BITS, bool DB, bool STAMP = false>
__global__ __launch_bounds__(512) void conv_band_kernel(BandP p, unsigned long long* stamps = nullptr) {
    constexpr int NT = 512;
    constexpr int K = TH * TW * C, KSTEPS = K / 16;
    constexpr int PS = C * 2 + 16;          // band pixel stride (bytes): +16 B keeps ds_read_b128 conflict-free
    constexpr int WPS = 8 / NSET;           // waves per weight set
    constexpr int CPP = C / 8;              // 16-byte chunks per pixel
    static_assert(NT % CPP == 0, "a thread keeps one channel chunk");
    extern __shared__ __attribute__((aligned(16))) char band0[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int set = wave % NSET, part = wave / NSET;           // wave-uniform (SGPRs): p.cls[set] is read with scalar loads
    const BandCls& cl = p.cls[set];
    // The class fields used inside the unit loop live in SGPRs: `set` comes from threadIdx, so p.cls[set] is a VECTOR load from the
    // kernel-argument segment, and under register pressure the compiler re-issued those loads inside the loop — each one consumed at
    // once, i.e. (vector memory returns in order) a wait for the whole prefetch in front of the MFMAs and in front of every store.
    const int cl_OH = __builtin_amdgcn_readfirstlane(cl.OH), cl_OW = __builtin_amdgcn_readfirstlane(cl.OW);
    const int cl_co = __builtin_amdgcn_readfirstlane(cl.co_base);
    const long cl_yoff = (long)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)((unsigned long long)cl.y_off >> 32)) << 32) |
                                (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)cl.y_off));
    const int Wb = (p.OWmax - 1) * S + TW;                  // band columns (padding included)
    const float inv_Wb = __builtin_amdgcn_rcpf((float)Wb), inv_OW = __builtin_amdgcn_rcpf((float)cl_OW);
    const int bands = (p.OHmax + p.R - 1) / p.R;
    const int nunits = MULTI ? (p.Nimg + p.F - 1) / p.F : p.Nimg * bands;

    // ---- band staging plan of this thread: chunk j covers band pixel (tid / CPP + j * NT / CPP), channel chunk tid % CPP
    const int cc = tid % CPP;
    uint4 pre[MAXCH];
    unsigned pre_live = 0;               // bit j: chunk j of the prefetched band is inside the frame (else it is zero padding)
    // unit -> first frame n, frames in the unit fu, first output row r0, output rows R, band rows per frame
    auto band_rows = [&](int unit, int& n, int& fu, int& r0, int& R, int& rows) {
        if (MULTI) { n = unit * p.F; fu = n + p.F <= p.Nimg ? p.F : p.Nimg - n; r0 = 0; R = p.OHmax; }
        else { n = unit / bands; fu = 1; const int b = unit % bands; r0 = b * p.R; R = (r0 + p.R <= p.OHmax) ? p.R : p.OHmax - r0; }
        rows = (R - 1) * S + TH;
    };
    auto stage_load = [&](int unit) {
        int n, fu, r0, R, rows; band_rows(unit, n, fu, r0, R, rows);
        const int iy0 = r0 * S - p.pad_y, fpx = rows * Wb, npx = fu * fpx;
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int px = tid / CPP + j * (NT / CPP);
            int pxc = px < npx ? px : 0, f = 0;
            if (MULTI) { f = fast_div(pxc, __builtin_amdgcn_rcpf((float)fpx)); pxc -= f * fpx; }
            const int br = fast_div(pxc, inv_Wb), bc = pxc - br * Wb;
            const int iy = iy0 + br, ix = bc - p.pad_x;
            const bool inb = px < npx && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            // unconditional load from a clamped address, then select (no branch around the load)
            const long off = inb ? (long)(n + f) * p.x_sn + (long)iy * p.x_sy + (long)ix * p.x_sx + cc * 8 : (long)n * p.x_sn;
            // the loaded value is not touched here (the select happens in stage_store): an ALU use would put the wait for the
            // prefetch in front of the MFMA loop it overlaps
            pre[j] = band_load_x<XF32>(p.X, off);
            pre_live = j == 0 ? (inb ? 1u : 0u) : (pre_live | ((inb ? 1u : 0u) << j));
        }
    };
    auto stage_store = [&](int unit, char* band) {
        int n, fu, r0, R, rows; band_rows(unit, n, fu, r0, R, rows);
        const int npx = fu * rows * Wb;
        // the thread id is made opaque here: inside the tile loop (double band) the compiler would otherwise compute the MAXCH LDS addresses
        // and predicates once per unit and keep them in registers across the MFMA loops (spills at the 256-VGPR budget)
        int t2 = tid;
        asm volatile("" : "+v"(t2));
        const int px0 = t2 / CPP, c2 = t2 % CPP;
#pragma unroll
        for (int j = 0; j < MAXCH; ++j) {
            const int px = px0 + j * (NT / CPP);
            if (px < npx) *(uint4*)(band + px * PS + c2 * 16) = ((pre_live >> j) & 1u) ? pre[j] : make_uint4(0, 0, 0, 0);
        }
    };

    // ---- prologue (round 4).  Was: every wave fetched its own 32 x K weight tile straight into registers, lane r = row r — 36 load
    // instructions per wave that each touch 32 rows (32 tag lookups for 1 KB), the same tile by every wave of a set, and the per-class tap
    // offsets as vector loads in front of them: 288 KB through the CU's L1 per workgroup and ~9.5 us before the first MFMA of a launch
    // (HULC_BAND_DBG=32 in round 3's build; twelve such launches per step).  Now the workgroup copies the NSET x 32 x K weights ONCE,
    // coalesced (a load instruction = 64 / CPP rows of one tap, whole 64 / 128-byte runs; tap and set are wave-uniform, so the offsets
    // are scalar loads), through registers into the (still empty) band area of the LDS, rows padded by 16 bytes, and every wave picks its
    // fragments up with ds_read_b128; the first band's loads are requested in between and land while that happens.
    constexpr int RPI = 64 / CPP;                            // weight rows one load instruction covers
    constexpr int WITEMS = (32 / RPI) * TH * TW;             // load instructions per weight set
    constexpr int NW = (WITEMS + WPS - 1) / WPS;             // ... per wave
    constexpr int WS = K * 2 + 16;                           // LDS bytes per weight row
    uint4 wtmp[NW];
    {
        const int wrow = lane / CPP, wc = lane % CPP;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int it = part + i * WPS;                   // item = (row group, tap) of this wave's set; wave-uniform
            const int itc = it < WITEMS ? it : WITEMS - 1;
            const int rg = itc / (TH * TW), t = itc % (TH * TW);
            wtmp[i] = band_load_bits(p.Wt, (cl.w_row0 + rg * RPI + wrow) * p.ldw + cl.w_tap_off[t] + wc * 8);
        }
    }
    int unit = blockIdx.x;
    if (unit < nunits) stage_load(unit);
    {
        const int wrow = lane / CPP, wc = lane % CPP;
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int it = part + i * WPS;
            if (it < WITEMS) {
                const int rg = it / (TH * TW), t = it % (TH * TW);
                *(uint4*)(band0 + (set * 32 + rg * RPI + wrow) * WS + (t * C + wc * 8) * 2) = wtmp[i];
            }
        }
    }
    // bias of this set's 32 output channels: in LDS (16 registers per lane otherwise); the accumulators start from it
    __shared__ float sbias[BAND_MAXCLS * 32];
    if (tid < NSET * 32) sbias[tid] = p.bias ? p.bias[p.cls[tid >> 5].co_base + (tid & 31)] : 0.f;
    __syncthreads();
    // ---- this wave's weight tile -> registers (A operand: lane = output channel row, 8 consecutive k)
    bf16x8_t wfrag[KSTEPS];
#pragma unroll
    for (int ks = 0; ks < KSTEPS; ++ks) wfrag[ks] = *(const bf16x8_t*)(band0 + (set * 32 + r) * WS + (ks * 16 + h * 8) * 2);
    // The resident operands are complete before the band area is overwritten (and before the unit loop: a fragment "possibly still in
    // flight" would be guarded by a wait in front of every MFMA that uses it).
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0) only
    __syncthreads();

    char* band = band0;
    if (unit < nunits) stage_store(unit, band);
    __syncthreads();
    // (STAMP, HULC_BANDK_STAMPS: per-wave cycle sums of a unit's phases — issue of the next band's loads | tile loop | barrier | LDS stores + barrier)
    unsigned long long t_ph[4] = {0, 0, 0, 0}, t_units = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (; unit < nunits; unit += gridDim.x) {
        const int next = unit + gridDim.x;
        const bool have_next = next < nunits && !(p.dbg & 4);
        if (STAMP) c0 = __builtin_readcyclecounter();
        // (round 6, single band) the next band's loads leave WAVE BY WAVE over the tile loop, each wave in front of its own tile slot, instead
        // of as one burst of 8 x MAXCH load instructions queueing at the CU's vector-memory path while no wave multiplies (the weight-gradient
        // kernels' finding, tools/study/wband_stamps.py).  HULC_BAND_DBG bit 64: the burst.  With two bands the band is written early: burst.
        // Measured: the gripper camera's small maps (several frames per unit) gain (conv2 forward 42 -> 35 us per 2048 frames, conv3 29 -> 25); the
        // static camera's conv2 forward does not (probe 170 / 170 us, inside the step 143 -> 148): staggered for the packed-frame instances only.
        const bool stagger = MULTI && !DB && BITS != 2 && !(p.dbg & 64);
        if (have_next && !stagger) stage_load(next);                         // in flight during the MFMA loop below
        char* band_next = DB ? (band == band0 ? band0 + p.lds_band : band0) : band;
        bool staged = !DB || !have_next;

        int n, fu, r0, R, rows; band_rows(unit, n, fu, r0, R, rows);
        const int Rc = r0 < cl_OH ? ((r0 + R <= cl_OH) ? R : cl_OH - r0) : 0;   // this class may have fewer rows/cols
        const int fpix = Rc * cl_OW, npix = fu * fpix;
        const int ntile = (npix + 31) / 32;
        const int my_tiles = ntile > part ? (ntile - part + WPS - 1) / WPS : 0;
        const int slot = (wave * my_tiles) >> 3;                             // (waves w and w + 4 share a SIMD: half the loop apart)
        bool issued = !have_next || !stagger;
        int it = 0;
        if (STAMP) c1 = __builtin_readcyclecounter();
        for (int tile = part; tile < ntile; tile += WPS, ++it) {
            if (!issued && it == slot) { stage_load(next); issued = true; }
            int q = tile * 32 + r;
            const bool live = q < npix;
            if (!live) q = npix - 1;
            int f = 0;
            if (MULTI) { f = fast_div(q, __builtin_amdgcn_rcpf((float)fpix)); q -= f * fpix; }
            const int oy = fast_div(q, inv_OW), ox = q - oy * cl_OW;
            const char* a0 = band + ((f * rows + oy * S) * Wb + ox * S) * PS + h * 16;
            // the accumulator starts as the bias (registers 4g..4g+3 = channels co_base + 8g + 4h + {0..3}): four LDS reads that travel with
            // the first pixel fragments instead of four read -> wait -> add round trips in the epilogue
            f32x16_t acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(sbias + set * 32 + 8 * g + 4 * h);
                acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
            }
            // the pixel's sign-plane word is requested before the MFMA loop (one dword per lane, both lane halves the same address): its
            // latency hides under the loop — the bf16 mask below is two 16-byte loads per lane consumed right where they are issued
            unsigned mb_in = 0;
            if (BITS == 2) {
                const long pix_off = cl_yoff + (long)(n + f) * p.y_sn + (long)(r0 + oy) * p.y_sy + (long)ox * p.y_sx;
                mb_in = p.bits_in[(long)(cl_co >> 5) * p.bplane + (pix_off >> p.bshift)];
            }
            // pixel fragments are read RD k-steps ahead of the MFMA that consumes them: left to the compiler (256 VGPRs in use) every
            // MFMA waited for the one ds_read_b128 issued just before it — an LDS round trip (~250 cycles with 8 waves reading) per
            // 32-cycle MFMA, 21-27 % of the matrix pipe
            constexpr int RD = KSTEPS <= 16 ? 8 : 4;            // (the 2 x 2-tap data-gradient classes keep 64 weight registers: room for 8)
            auto frag = [&](int ks) {
                const int k0 = ks * 16;
                const int t = k0 / C, c0 = k0 % C;
                const int ty = t / TW, tx = t % TW;
                return *(const bf16x8_t*)(a0 + (ty * Wb + tx) * PS + c0 * 2);
            };
            bf16x8_t pf[RD];
#pragma unroll
            for (int i = 0; i < RD; ++i) pf[i] = frag(i);
            if (!(p.dbg & 1))
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                const bf16x8_t px = pf[ks % RD];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wfrag[ks], px, acc, 0, 0, 0);   // D[channel][pixel]
                __builtin_amdgcn_sched_barrier(0);
                if (ks + RD < KSTEPS) pf[ks % RD] = frag(ks + RD);
                __builtin_amdgcn_sched_barrier(0);
            }
            // ---- epilogue: lane = pixel; registers 4g..4g+3 = channels co_base + 8g + 4h + {0..3}
            const long off0 = cl_yoff + (long)(n + f) * p.y_sn + (long)(r0 + oy) * p.y_sy + (long)ox * p.y_sx + cl_co;   // q is clamped: always valid
            if (p.y_dtype == HULC_BF16) {
                // bf16 outputs: the two lane halves of a pixel hold interleaved groups of 4 channels (8-byte pieces).  v_permlane32_swap
                // trades the odd pieces of the lower half for the even pieces of the upper half: every lane then owns 8 consecutive
                // channels — two 16-byte stores (and mask loads) per pixel instead of four 8-byte ones.
                // (ReLU on the PACKED words — max per signed 16-bit half —, sign bits from min(half, 1), mask bits applied as a packed multiply:
                //  a tile's MFMAs and its epilogue VALU share the SIMD's issue slot, ~60 instructions fewer per tile than float max / compare chains)
                uint2 pk[4];
                unsigned mb_out = 0;
                const uint32_t floor2 = p.relu ? 0u : 0x80008000u;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    if (p.add) {
                        const uint2 a = *(const uint2*)((const uint16_t*)p.add + off0 + 8 * g + 4 * h);
                        v[0] += __uint_as_float(a.x << 16); v[1] += __uint_as_float(a.x & 0xffff0000u);
                        v[2] += __uint_as_float(a.y << 16); v[3] += __uint_as_float(a.y & 0xffff0000u);
                    }
                    pk[g] = make_uint2(max_s16x2(pack_bf16x2(v[0], v[1]), floor2), max_s16x2(pack_bf16x2(v[2], v[3]), floor2));
                }
                // (double band) the next band goes to LDS here — behind the wave's SECOND MFMA loop, with the tile packed into 8 registers and
                // BEFORE its stores: the vmcnt(0) in front of the LDS write covers the prefetch and the previous tile's stores only
                if (DB && !staged && tile >= part + WPS) { stage_store(next, band_next); staged = true; }
#pragma unroll
                for (int gp = 0; gp < 2; ++gp) {
                    const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * gp].x, pk[2 * gp + 1].x, false, false);
                    const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * gp].y, pk[2 * gp + 1].y, false, false);
                    uint32_t o[4] = {sx[0], sy[0], sx[1], sy[1]};                 // channels co_base + 16 gp + 8 h + {0..7}
                    const long off = off0 + 16 * gp + 8 * h;
                    if (BITS == 2) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = keep_u16x2(o[e], (mb_in >> (16 * gp + 8 * h + 2 * e)) & 3u);
                    } else if (p.mask) {
                        const uint4 m = *(const uint4*)((const uint16_t*)p.mask + off);
                        const uint32_t mw[4] = {m.x, m.y, m.z, m.w};
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            if (!(__uint_as_float(mw[e] << 16) > 0.f)) o[e] &= 0xffff0000u;
                            if (!(__uint_as_float(mw[e] & 0xffff0000u) > 0.f)) o[e] &= 0x0000ffffu;
                        }
                    }
                    if (BITS == 1) {                                              // (rectified values: positive <=> half != 0; the dispatch insists on relu)
                        const unsigned a = nonzero_u16x2(o[0]) | (nonzero_u16x2(o[1]) << 2) | (nonzero_u16x2(o[2]) << 4) | (nonzero_u16x2(o[3]) << 6);
                        mb_out |= ((a | (a >> 15)) & 0xffu) << (16 * gp);         // even bits from the low halves, odd bits from the high halves
                    }
                    if (live && !(p.dbg & 2)) *(uint4*)((uint16_t*)p.Y + off) = make_uint4(o[0], o[1], o[2], o[3]);
                }
                if (BITS == 1) {
                    mb_out <<= 8 * h;
                    const auto other = __builtin_amdgcn_permlane32_swap(mb_out, mb_out, false, false);   // [1] on the lower lanes = the upper half's word
                    mb_out |= other[1];
                    if (live && h == 0) p.bits_out[(long)(cl_co >> 5) * p.bplane + ((off0 - cl_co) >> p.bshift)] = mb_out;   // 32 lanes: 128 contiguous bytes
                }
            } else {
              if (DB && !staged && tile >= part + WPS) { stage_store(next, band_next); staged = true; }
              if (live) {
                const long off = off0 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v[4] = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
                    if (p.relu) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
                    }
                    if (p.mask) {
                        const uint2 m = *(const uint2*)((const uint16_t*)p.mask + off + 8 * g);
                        if (!(__uint_as_float(m.x << 16) > 0.f)) v[0] = 0.f;
                        if (!(__uint_as_float(m.x & 0xffff0000u) > 0.f)) v[1] = 0.f;
                        if (!(__uint_as_float(m.y << 16) > 0.f)) v[2] = 0.f;
                        if (!(__uint_as_float(m.y & 0xffff0000u) > 0.f)) v[3] = 0.f;
                    }
                    *(float4*)((float*)p.Y + off + 8 * g) = make_float4(v[0], v[1], v[2], v[3]);
                    // (hulc_conv_desc.y_bf16) the bf16 map a bf16-output launch would have stored, next to the exact one
                    if (p.Y16) *(uint2*)((uint16_t*)p.Y16 + off + 8 * g) = make_uint2(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]));
                }
              }
            }
        }
        if (!issued) stage_load(next);                       // (a wave without a tile in this unit)
        if (STAMP) c2 = __builtin_readcyclecounter();
        if (DB) {
            if (!staged) stage_store(next, band_next);       // (a wave with fewer than two tiles in this unit)
            __syncthreads();                                 // every wave is done reading this band and has written its part of the next
            band = band_next;
        } else {
            __syncthreads();                                 // every wave is done reading this band
            if (STAMP) c3 = __builtin_readcyclecounter();
            if (have_next) stage_store(next, band);
            __syncthreads();
        }
        if (STAMP) {
            const unsigned long long c4 = __builtin_readcyclecounter();
            t_ph[0] += c1 - c0; t_ph[1] += c2 - c1; t_ph[2] += c3 - c2; t_ph[3] += c4 - c3; t_units += 1;
        }
    }
    if (STAMP && stamps && lane == 0) {
        unsigned long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 5;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = t_ph[e];
        o[4] = t_units;
    }
}

template <int C, int NSET, int TH, int TW, int S, int MAXCH, bool XF32, int BITS, bool DB>
int launch_band_x(BandP& p, hipStream_t s) {
    constexpr int PS = C * 2 + 16, CPP = C / 8;
    const int Wb = (p.OWmax - 1) * S + TW;
    const long budget = (160 * 1024 - 1024) / (DB ? 2 : 1) / 16 * 16;   // (512 B of static LDS: the bias table)
    const long max_px = (long)MAXCH * (512 / CPP);           // pixels one register-staged band can hold
    int R = p.OHmax;
    auto px_of = [&](int rr) { return (long)((rr - 1) * S + TH) * Wb; };
    while (R > 1 && (px_of(R) * PS > budget || px_of(R) > max_px)) --R;
    if (px_of(R) * PS > budget || px_of(R) > max_px) return -1;
    int F = 1, nunits;
    if (R == p.OHmax) {                                      // whole frames fit: pack several into one unit (small gripper maps)
        while (F < p.Nimg && px_of(R) * (F + 1) * PS <= budget && px_of(R) * (F + 1) <= max_px) ++F;
        nunits = (p.Nimg + F - 1) / F;
    } else {
        const int bands = (p.OHmax + R - 1) / R;
        R = (p.OHmax + bands - 1) / bands;                   // equal-ish bands
        nunits = p.Nimg * bands;
    }
    p.R = R; p.F = F;
    if ((long)F * R * p.OWmax < 128) return -1;              // a unit that cannot feed 8 waves: the gather kernel is faster
    p.lds_band = (int)(((size_t)px_of(R) * F * PS + 15) / 16 * 16);
    size_t lds = (size_t)p.lds_band * (DB ? 2 : 1);
    const size_t wbytes = (size_t)NSET * 32 * (TH * TW * C * 2 + 16);     // the prologue parks the weights in the band area
    if (lds < wbytes) lds = wbytes;
    const int per = (nunits + 255) / 256;                    // balanced persistent grid
    const int grid = (nunits + per - 1) / per;
    if (F > 1) {
        auto kern = conv_band_kernel<C, NSET, TH, TW, S, MAXCH, true, XF32, BITS, DB>;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -2;
            attr_set = true;
        }
        kern<<<grid, 512, lds, s>>>(p, nullptr);
    } else {
        auto kern = conv_band_kernel<C, NSET, TH, TW, S, MAXCH, false, XF32, BITS, DB>;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -2;
            attr_set = true;
        }
        const char* se = getenv("HULC_BANDK_STAMPS");          // <device address of grid x 8 x 5 uint64>: the instrumented instance (conv2 forward)
        if (se && *se && C == 32 && !XF32 && !DB && BITS == 1) {
            auto kst = conv_band_kernel<C, NSET, TH, TW, S, MAXCH, false, XF32, BITS, DB, true>;
            static bool st_attr = false;
            if (!st_attr) {
                if (hipFuncSetAttribute((const void*)kst, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512) != hipSuccess) return -2;
                st_attr = true;
            }
            kst<<<grid, 512, lds, s>>>(p, (unsigned long long*)strtoull(se, nullptr, 0));
        } else
        kern<<<grid, 512, lds, s>>>(p, nullptr);
    }
    return 0;
}


// ---- (round 4) the stride-1, 64-channel forward geometry (conv3 of the static camera) with its bands brought in by DIRECT global -> LDS
// loads (global_load_lds_dwordx4) into two LDS bands: no staging registers, no LDS write pass, one barrier per unit.  A wave instruction fills 1 KB
// of LDS in lane order (the destination is wave-uniform base + lane * 16), so the image is pixel-major 128-byte pixels WITHOUT padding and the
// bank spread of the fragment reads comes from an XOR swizzle applied on the SOURCE address: slot j of pixel q holds channel chunk
// j ^ ((q >> 1) & 7) — 16 consecutive pixels of one chunk fall on 16 different 16-byte bank columns.  Whole frames per unit, no zero padding
// (forward, pad 0), bf16 output, no mask / residual / sign planes.
constexpr int GLDS_BAND_BYTES = 68 * 1024;      // a 23 x 23 x 64-channel frame = 67 712 bytes, rounded to whole 1 KB instructions
// PAD: the band is the (unpadded) input frame + ONE zero pixel behind it; a tap that falls outside the frame reads that pixel (the data
// gradients' zero padding without materialising a padded band: 26 x 26 padded pixels of conv2's data gradient would not fit two bands).
// BITS = 2: the ReLU sign-plane word of every output pixel masks the result; the words of the NEXT unit's tiles are requested before the
// unit's direct loads and consumed a unit later (an ordinary load consumed while a direct load is in flight drains it: vmcnt counts both).
template <int NSET, int TH, int TW, int BITS, bool PAD>
__global__ __launch_bounds__(512) void conv_band_glds_kernel(BandP p) {
    constexpr int C = 64, NT = 512;
    constexpr int K = TH * TW * C, KSTEPS = K / 16, CPP = 8, WPS = 8 / NSET;
    constexpr int SPA = NSET / 2;                           // weight sets parked per band array in the prologue
    // two STATIC LDS arrays (distinct objects: an LDS read of one is then provably independent of a direct load in flight into the other —
    // with one dynamic array hipcc puts s_waitcnt vmcnt(0) in front of the first ds_read behind a glds, i.e. drains the next band before the tile)
    constexpr int BANDB = GLDS_BAND_BYTES;
    // (neither band may sit at LDS address 0: there the base folds into the offset arithmetic, the access loses its underlying object and is
    // guarded again — the bias table is given the larger alignment so that it takes the first slot)
    __shared__ __attribute__((aligned(2048))) float sbias[BAND_MAXCLS * 32 + 384];
    __shared__ __attribute__((aligned(256))) char bandA_[BANDB];
    __shared__ __attribute__((aligned(256))) char bandB_[BANDB];
    constexpr int MAXTW = BITS == 2 ? (NSET == 2 ? 6 : 12) : 1;             // tiles of one wave per unit that carry a sign-plane word
    __shared__ unsigned smask[8][MAXTW][32];                                // ... parked here between the unit that fetched them and the unit that uses them
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int set = wave % NSET, part = wave / NSET;
    const BandCls& cl = p.cls[set];
    const int cl_OH = cl.OH, cl_OW = cl.OW, cl_co = cl.co_base;
    const long cl_yoff = cl.y_off;
    const int Wb = p.W;                                     // the band IS the input frame (stride 1); padding = the zero pixel
    const float inv_OW = __builtin_amdgcn_rcpf((float)cl_OW);
    const int nunits = p.Nimg;
    const int P = p.H * p.W, NSLOT = P * CPP, NINS = (NSLOT + 63) / 64;     // glds instructions per band; pixel P = the zero pixel
    const int npix = cl_OH * cl_OW, ntile = (npix + 31) / 32;

    // ---- prologue: weights once through LDS (as conv_band_kernel), SPA sets per (still empty) band array
    constexpr int RPI = 64 / CPP, WITEMS = (32 / RPI) * TH * TW, NW = (WITEMS + WPS - 1) / WPS, WS = K * 2 + 16;
    static_assert((long)SPA * 32 * WS <= BANDB, "the parked weights fit a band array");
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
    if (PAD && tid < 16) *(uint4*)((tid < 8 ? bandA_ : bandB_) + P * 128 + (tid & 7) * 16) = make_uint4(0u, 0u, 0u, 0u);   // the zero pixels (never written again)

    // direct loads of one frame into a band: instruction i of this wave covers slots [64 i, 64 i + 64); lanes past the frame stay out
    // (EXEC-masked: the zero pixel sits right behind it)
    auto glds_band = [&](int unit, char* band) {
        const uint16_t* frame = (const uint16_t*)p.X + (long)unit * p.x_sn;
        for (int i = wave; i < NINS; i += 8) {
            const int slot = i * 64 + lane;
            if (slot < NSLOT) {
                const int q = slot >> 3, j = slot & 7, c = j ^ ((q >> 1) & 7);
                __builtin_amdgcn_global_load_lds((const void __attribute__((address_space(1)))*)(frame + (long)q * C + c * 8),
                                                 (void __attribute__((address_space(3)))*)(band + i * 1024), 16, 0, 0);
            }
        }
    };
    // sign-plane words of a unit's tiles (one dword per output pixel and 32-channel plane)
    // (registers while in flight, written to smask behind the unit's closing wait — a tile loop that indexes registers would have to be fully
    //  unrolled: 12 tiles' index arithmetic live at once spilled 400-550 registers)
    unsigned mbn[MAXTW];
    auto mask_park = [&]() {
        if (BITS == 2 && lane < 32) {
#pragma unroll
            for (int i = 0; i < MAXTW; ++i) smask[wave][i][lane] = mbn[i];
        }
    };
    auto mask_fetch = [&](int unit) {
#pragma unroll
        for (int i = 0; i < MAXTW; ++i) {
            const int tile = part + i * WPS;
            int q = tile * 32 + r;
            q = q < npix ? q : npix - 1;
            if (tile >= ntile) q = 0;
            const int oy = fast_div(q, inv_OW), ox = q - oy * cl_OW;
            const long pix_off = cl_yoff + (long)unit * p.y_sn + (long)oy * p.y_sy + (long)ox * p.y_sx;
            mbn[i] = p.bits_in[(long)(cl_co >> 5) * p.bplane + (pix_off >> p.bshift)];
        }
    };

    int unit = blockIdx.x;
    if (BITS == 2 && unit < nunits) mask_fetch(unit);
    if (unit < nunits) glds_band(unit, bandA_);
    __builtin_amdgcn_s_waitcnt(0x0F70);                      // vmcnt(0) — the BUILTIN: the compiler's own wait-count bookkeeping sees it (an asm wait it does not, and guards the band reads again)
    mask_park();
    __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0): the zero pixels are in LDS before anybody passes the (raw) barrier
    __builtin_amdgcn_s_barrier();
    auto do_unit = [&](int unit, const char* __restrict__ band, char* __restrict__ band_next) {
        const int next = unit + gridDim.x;
        if (BITS == 2 && next < nunits) mask_fetch(next);         // requested BEFORE the direct loads, parked behind the closing wait, used a unit later
        if (next < nunits) glds_band(next, band_next);        // lands while this unit is multiplied

        auto do_tile = [&](int tile, unsigned mb_in) {
            int q = tile * 32 + r;
            const bool live = q < npix;
            if (!live) q = npix - 1;
            const int oy = fast_div(q, inv_OW), ox = q - oy * cl_OW;
            // band pixel of tap (ty, tx) = (oy + ty - pad_y) * W + (ox + tx - pad_x), or the zero pixel outside the frame
            int rowq[TH], colq[TW];
#pragma unroll
            for (int ty = 0; ty < TH; ++ty) { const int iy = oy + ty - p.pad_y; rowq[ty] = (!PAD || (iy >= 0 && iy < p.H)) ? iy * Wb : -0x40000000; }
#pragma unroll
            for (int tx = 0; tx < TW; ++tx) { const int ix = ox + tx - p.pad_x; colq[tx] = (!PAD || (ix >= 0 && ix < p.W)) ? ix : -0x40000000; }
            f32x16_t acc;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 bv = *(const float4*)(sbias + set * 32 + 8 * g + 4 * h);
                acc[4 * g] = bv.x; acc[4 * g + 1] = bv.y; acc[4 * g + 2] = bv.z; acc[4 * g + 3] = bv.w;
            }
            constexpr int RD = 8;
            auto frag = [&](int ks) {
                const int k0 = ks * 16, t = k0 / C, kc = (k0 % C) / 16;
                const int ty = t / TW, tx = t % TW;
                int qs = rowq[ty] + colq[tx];
                if (PAD) qs = qs < 0 ? P : qs;
                const unsigned u = (unsigned)qs << 3;
                const unsigned off = (u << 4) + ((((unsigned)(2 * kc) + (unsigned)h) << 4) ^ (u & 0x70u));
                return *(const bf16x8_t*)(band + off);
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
            uint2 pk[4];
            const uint32_t floor2 = p.relu ? 0u : 0x80008000u;  // (ReLU on the packed words: see the register kernel's epilogue)
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
                if (live) *(uint4*)((uint16_t*)p.Y + off0 + 16 * gp + 8 * h) = make_uint4(o[0], o[1], o[2], o[3]);
            }
        };
        int ti = 0;
        for (int tile = part; tile < ntile; tile += WPS, ++ti) do_tile(tile, BITS == 2 ? smask[wave][ti][r] : 0u);
        __builtin_amdgcn_s_waitcnt(0x0F70);                   // vmcnt(0): the next band has landed (and this unit's stores are acknowledged)
        if (next < nunits) mask_park();                        // (this unit's words have all been read: same wave, program order)
        __builtin_amdgcn_s_barrier();                          // every wave is done reading this band
    };
    for (; unit < nunits; unit += 2 * gridDim.x) {
        do_unit(unit, bandA_, bandB_);
        if (unit + (int)gridDim.x < nunits) do_unit(unit + gridDim.x, bandB_, bandA_);
    }
}

template <int NSET, int TH, int TW, int BITS, bool PAD>
int launch_band_glds(BandP& p, hipStream_t s) {
    const long P = (long)p.H * p.W + (PAD ? 1 : 0);
    if ((P * 128 + 1023) / 1024 * 1024 > GLDS_BAND_BYTES) return -1;
    const long npix = (long)p.OHmax * p.OWmax;
    if (BITS == 2 && (npix + 31) / 32 > (long)(NSET == 2 ? 6 : 12) * (8 / NSET)) return -1;      // the per-wave sign-word registers cover a unit's tiles
    const size_t lds = 0;                                    // (static LDS: two bands + the bias table)
    const int nunits = p.Nimg, per = (nunits + 255) / 256, grid = (nunits + per - 1) / per;
    auto kern = conv_band_glds_kernel<NSET, TH, TW, BITS, PAD>;
    kern<<<grid, 512, lds, s>>>(p);
    return 0;
}

// BITS_OK: which sign-plane role this geometry is ever launched with (1: forward conv2 writes them, 2: the data gradients read them)
template <int C, int NSET, int TH, int TW, int S, int MAXCH, int BITS_OK, bool DB>
int launch_band_db(BandP& p, hipStream_t s) {
    const int want = p.bits_out ? 1 : (p.bits_in ? 2 : 0);
    if (want && want != BITS_OK) return -1;
    if (want) {
        constexpr int B = BITS_OK ? BITS_OK : 1;
        return p.x_dtype == HULC_F32 ? launch_band_x<C, NSET, TH, TW, S, MAXCH, true, B, DB>(p, s) : launch_band_x<C, NSET, TH, TW, S, MAXCH, false, B, DB>(p, s);
    }
    return p.x_dtype == HULC_F32 ? launch_band_x<C, NSET, TH, TW, S, MAXCH, true, 0, DB>(p, s) : launch_band_x<C, NSET, TH, TW, S, MAXCH, false, 0, DB>(p, s);
}
// MAXCH_DB: staging registers of the double-band instance — its bands are at most half the LDS, so fewer chunks per thread (and the registers
// they would occupy are what keeps the instance from spilling the prefetch around its MFMA loops)
template <int C, int NSET, int TH, int TW, int S, int MAXCH, int BITS_OK, int MAXCH_DB>
int launch_band(BandP& p, hipStream_t s) {
    static const char* e = getenv("HULC_BAND_DB");
    const bool db = e ? atoi(e) != 0 : false;               // (measured slower on every geometry of the policy: opt-in)
    if (db) {
        const int rc = launch_band_db<C, NSET, TH, TW, S, MAXCH_DB, BITS_OK, true>(p, s);
        if (rc != -1) return rc;                             // (-1: half the LDS cannot hold a useful band — single band)
    }
    return launch_band_db<C, NSET, TH, TW, S, MAXCH, BITS_OK, false>(p, s);
}

// a ReLU mask that comes as an activation tensor only (no sign planes): the direct-to-LDS instances read planes
inline bool mask_only(const void* mask, const unsigned* bits_in) { return mask != nullptr && bits_in == nullptr; }

}  // namespace

// One launch covering `ncls` weight sets (output-channel tiles of a forward conv, or parity classes of a data gradient).
// returns 0 when the band kernel took the launch, 1 when the geometry is not covered (caller falls back to the gather
// kernel), negative on error.  bf16 compute only; NHWC input with C in {32, 64}; 32 output channels per set.
int hulc_conv_band_dispatch(int C, int NSET, int TH, int TW, int S, const void* x, int x_dtype, int N, int H, int W, int pad_y, int pad_x,
                            long x_sn, long x_sy, long x_sx, void* y, int y_dtype, long y_sn, long y_sy, long y_sx, const void* wt,
                            int w_dtype, long ldw, const float* bias, const void* mask, int mask_dtype, int relu, int ncls,
                            const int* cls_OH, const int* cls_OW, const long* cls_yoff, const int* cls_cobase, const long* cls_wrow0,
                            const long* cls_wtap /* [ncls][16] */, const void* add, unsigned* bits_out, const unsigned* bits_in, int bits_channels,
                            void* y16, hipStream_t s) {
    if (getenv("HULC_NO_BAND")) return 1;
    if (ncls != NSET || ncls > BAND_MAXCLS || TH * TW > 16) return 1;
    if (y16 && ((y_dtype != HULC_F32 && y_dtype != HULC_F16) || mask || add || bits_out || bits_in)) return 1;      // (the copy rides on the plain forward only)
    if (y_dtype == HULC_F16 && !y16) return 1;
    BandP p;
    p.X = x; p.Y = y; p.Wt = wt; p.bias = bias; p.mask = mask; p.add = add; p.Y16 = y16;
    p.bits_out = bits_out; p.bits_in = bits_in; p.bshift = 0; p.bplane = 0;
    if (bits_out || bits_in) {
        // planes exist for bf16 tensors of 32 / 64 / 128 channels whose pixels are whole multiples of the channel count apart
        if (y_dtype != HULC_BF16 || (bits_channels != 32 && bits_channels != 64 && bits_channels != 128) || y_sx % bits_channels || y_sy % bits_channels ||
            y_sn % bits_channels) return 1;
        for (int c = 0; c < ncls; ++c) if (cls_yoff[c] % bits_channels) return 1;
        while ((1 << p.bshift) < bits_channels) ++p.bshift;
        p.bplane = (long)N * y_sn / bits_channels;           // pixels of the (dense) tensor the planes describe
    }
    if (add && y_dtype != HULC_BF16) return 1;
    if (bits_out && !relu) return 1;                        // (the planes describe a rectified map: positive <=> non-zero)
    if (w_dtype != HULC_BF16 || (mask && mask_dtype != HULC_BF16)) return 1;   // the gather kernel serves other storage types
    p.x_dtype = x_dtype; p.y_dtype = y_dtype; p.w_dtype = w_dtype; p.mask_dtype = mask_dtype;
    p.Nimg = N; p.H = H; p.W = W; p.pad_y = pad_y; p.pad_x = pad_x; p.R = 1; p.F = 1;
    p.x_sn = x_sn; p.x_sy = x_sy; p.x_sx = x_sx; p.y_sn = y_sn; p.y_sy = y_sy; p.y_sx = y_sx;
    p.ldw = ldw; p.relu = relu;
    { static const char* e = getenv("HULC_BAND_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.OHmax = 0; p.OWmax = 0;
    for (int c = 0; c < ncls; ++c) {
        p.cls[c].OH = cls_OH[c]; p.cls[c].OW = cls_OW[c]; p.cls[c].y_off = cls_yoff[c]; p.cls[c].co_base = cls_cobase[c];
        p.cls[c].w_row0 = cls_wrow0[c];
        for (int t = 0; t < 16; ++t) p.cls[c].w_tap_off[t] = t < TH * TW ? cls_wtap[c * 16 + t] : 0;
        if (cls_OH[c] > p.OHmax) p.OHmax = cls_OH[c];
        if (cls_OW[c] > p.OWmax) p.OWmax = cls_OW[c];
    }
    int rc = 1;
    {   // round 6: four waves per workgroup, every weight set in every wave (conv_band4.hip); HULC_BAND4=0: the kernels below only
        const char* e4 = getenv("HULC_BAND4");              // (read per launch: the tests switch between the kernels inside one process)
        const int band4 = e4 ? atoi(e4) : 0;
        if (band4 && !y16) {
            const int rc4 = launch_band4(p, C, NSET, TH, TW, S, s);
            if (rc4 == 0) return 0;
            if (rc4 == -2) return hulc_fail(-8, "conv band4: could not raise the dynamic LDS limit");
        }
    }
    if (y_dtype == HULC_F16 && !(C == 64 && S == 1 && NSET == 2 && TH == 3 && TW == 3)) return 1;
    if (C == 32 && NSET == 2 && TH == 4 && TW == 4 && S == 2) rc = launch_band<32, 2, 4, 4, 2, 12, 1, 7>(p, s);        // conv2 forward (writes sign planes)
    else if (C == 64 && S == 1 && ((NSET == 2 && TH == 3 && TW == 3) || (NSET == 4 && TH == 2 && TW == 2))) {
        // conv3 forward / data gradient, conv2 data gradient (4 parity classes).  Frame-sized maps take the direct-to-LDS instances
        // (conv_band_glds_kernel; HULC_BAND_GLDS=0: the register-staged kernel; small maps pack several frames into a unit there)
        static const char* ge = getenv("HULC_BAND_GLDS");
        rc = -1;
        {   // round 6: the chunk-major direct-to-LDS band (conv_band_planes.hip): fragment reads without address arithmetic
            // default: conv3's forward (70 vs 75 us per 2048 frames) and data gradient (89 vs 97); conv2's data gradient measures the same on
            // both kernels and stays; HULC_BAND_PLANES=1: every geometry it covers, =0: none
            const char* pe = getenv("HULC_BAND_PLANES");       // (read per launch: the tests switch inside one process)
            const int want = pe ? atoi(pe) : -1;
            const bool conv3 = NSET == 2 && TH == 3 && TW == 3;    // (forward: 70 vs 75 us with its loads spread over the tile loop)
            if ((want == 1 || (want == -1 && conv3)) && !(ge && !atoi(ge))) rc = launch_band_planes(p, NSET, TH, TW, s);
        }
        if (rc == -1 && y_dtype == HULC_F16) return 1;       // (the fp16 twin is stored by the direct-to-LDS kernel only)
        if (rc == -1) {
        const bool contiguous = x_sx == 64 && x_sy == (long)W * 64 && x_sn == (long)H * W * 64 && ((uintptr_t)x % 16) == 0;
        if (!(ge && !atoi(ge)) && !mask_only(mask, bits_in) && !add && !bits_out && x_dtype == HULC_BF16 && y_dtype == HULC_BF16 && contiguous &&
            (long)p.OHmax * p.OWmax >= 256 && !p.dbg) {
            const bool padded = pad_y != 0 || pad_x != 0;
            if (NSET == 2 && !padded && !bits_in) rc = launch_band_glds<2, 3, 3, 0, false>(p, s);
            else if (NSET == 2 && padded && bits_in) rc = launch_band_glds<2, 3, 3, 2, true>(p, s);
            else if (NSET == 4 && padded && bits_in) rc = launch_band_glds<4, 2, 2, 2, true>(p, s);
        }
        }
        if (rc == -1) rc = NSET == 2 ? launch_band<64, 2, 3, 3, 1, 10, 2, 9>(p, s) : launch_band<64, 4, 2, 2, 1, 12, 2, 9>(p, s);
    }
    else return 1;
    if (rc == -1) return 1;                      // band does not fit: gather kernel
    if (rc < 0) return hulc_fail(-8, "conv band: could not raise the dynamic LDS limit");
    return 0;
}
