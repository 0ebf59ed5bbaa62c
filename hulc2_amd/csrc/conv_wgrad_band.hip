// conv_wgrad_band.hip — weight gradient of the camera-encoder convs from LDS-staged bands (bf16 compute).
//
// reference arithmetic: autograd's conv2d weight/bias gradient for nn.Conv2d of
// hulc2/models/perceptual_encoders/vision_network.py:37-46 and vision_network_gripper.py:13-19.
//
//   dW[co][k] = sum over output pixels m of dY[m][co] * X(m, k),   k = (tap, channel)
//
// The reduction index is the pixel, which is the OUTERMOST index of both tensors in memory; the kernel in conv.hip
// gathers every (pixel, tap) chunk from global memory and transposes it through registers (3x - 9x re-reads, 16-byte
// scattered loads: its limiter).  Here a work unit (frame x band of output rows) is staged ONCE:
//   * the input band X goes to LDS in its natural layout (pixel stride padded; conv1: bf16 channel planes made from the
//     NCHW fp32 frame), dY goes to LDS TRANSPOSED ([channel][pixel]) — a few hundred 2-byte LDS writes per unit;
//   * MFMA D[co][k] += A * B with A = dY^T fragment (one ds_read_b128: 8 consecutive pixels of one channel) and
//     B = X^T fragment assembled by eight 2-byte LDS column reads (lane = k index, the 8 pixels of the k-slot come from
//     a per-unit pixel-offset table);
//   * accumulators (Cout x K, fp32) stay in registers across ALL units of a persistent workgroup: each wave owns up to 3
//     k tiles x all channel tiles; one slab per workgroup is written at the end and summed by a fixed-order pass;
//   * the next unit's bands are prefetched into registers during the MFMA loop (single LDS buffer).
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "u8_frames.h"
#include <stdlib.h>

namespace {

struct WBandP {
    const void* X; const void* dY;
    int x_dtype, dy_dtype;
    int Nimg, H, W, OH, OW, R, F;     // R output rows per band; F > 1: a unit is F whole frames (then R == OH)
    long x_sn, x_sy, x_sx, x_sc;      // input element strides (x_sc: channel stride, used by the NCHW layout)
    long dy_sn, dy_sy, dy_sx;         // dY element strides (channels contiguous)
    float* partial_w;                 // [grid][Cout][K]
    float* partial_b;                 // [grid][Cout]
    int u8, pad; const int* shift; const int* fidx;   // conv1 fed by uint8 NHWC frames: shift / scale / normalise applied while staging (see conv1_band.hip)
    int burst;                        // bf16 NHWC layers: 1 = the next unit's loads as ONE burst in front of the MFMA loop (else staggered by wave)
    int stag_num;                     // the bursts are spread over the first stag_num / 8 of the loop
};

// C: input channels, CT: Cout / 32, TH x TW taps, S stride, NCHW: conv1 layout (k = (c, kh, kw), fp32 planes)
// PURE16: X and dY are bf16 (the benchmarked NHWC layers) — compile-time, so that no run-time dtype branch surrounds a prefetch load
template <int C, int CT, int TH, int TW, int S, bool NCHW, int XCH, int YCH, int BPC, bool PURE16, bool STAMP = false>
__global__ __launch_bounds__(512, 2 * BPC) void conv_wgrad_band_kernel(WBandP p, unsigned long long* stamps = nullptr) {
    constexpr int NT = 512;
    constexpr int COUT = CT * 32;
    constexpr int K = TH * TW * C, KTN = K / 32;          // 32-wide k tiles
    constexpr int MAXT = (KTN + 7) / 8;                   // k tiles per wave
    // X band geometry in LDS.  NHWC: [row][col][C] bf16, pixel stride PS.  NCHW: [c][row][col] bf16 planes.
    constexpr int PS = NCHW ? 2 : C * 2 + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int Wb = p.W;                                   // bands hold full input rows: staging is a contiguous copy (no index division)
    const int Wp = Wb;                                    // NCHW plane row pitch (elements)
    const int bands = (p.OH + p.R - 1) / p.R;
    const float inv_OW = fast_rcp(p.OW);
    const int nunits = p.F > 1 ? (p.Nimg + p.F - 1) / p.F : p.Nimg * bands;
    const int rows_max = (p.F - 1) * p.H + (p.R - 1) * S + TH;
    const int npix_max = (p.F * p.R * p.OW + 15) / 16 * 16;
    // TR (bf16 NHWC layers): both MFMA operands are fetched with ds_read_b64_tr_b16 — a 16-lane group hands in the addresses of a
    // [4 pixels][16 channels] block (lane i: pixel i / 4, channels 4 (i % 4) .. + 3, any row stride) and lane i gets channel i of the 4
    // pixels, i.e. 4 consecutive reduction indices (tools/probe/tr_probe.py pins this down).  dY stays in its natural [pixel][channel]
    // layout (staged with 16-byte copies instead of eight 2-byte scatters per chunk) and the X fragment is two transpose reads through the
    // pixel-offset table instead of eight 2-byte column reads.
    constexpr bool TR = !NCHW && PURE16;
    constexpr int DROW = COUT * 2 + 16;                   // TR: dY row stride (bytes), one row per output pixel
    const int AT_ROW = TR ? 0 : npix_max * 2 + 16;        // dY^T row stride (bytes)
    char* xband = smem;
    const int PP = (rows_max * Wp + 7) / 8 * 8;           // NCHW plane pitch (elements)
    const int xbytes = NCHW ? C * PP * 2 : rows_max * Wb * PS;
    char* at = smem + (xbytes + 15) / 16 * 16;            // [COUT][AT_ROW]
    int* pixoff = (int*)(at + (TR ? npix_max * DROW : COUT * AT_ROW));   // [npix_max] byte offset of each output pixel's patch origin

    // ---- per-lane byte offset of this lane's k index inside a patch, for each k tile the wave owns
    // k tiles of a wave.  KTN = 16 (conv2): tiles wave and wave + 8, both output-channel tiles of each.  KTN = 18 (conv3, round 6): with
    // "wave + 8 t" waves 0 and 1 carried THREE k tiles x 2 channel tiles = 6 accumulator tiles against the others' 4, and the workgroup
    // waited for them (MFMA loop 7 970 vs 6 260 cycles, tools/study/wband_stamps.py); now every wave has k tiles 2 wave, 2 wave + 1 and
    // tiles 16 / 17 are shared out by channel tile among waves 0-3: five accumulator tiles there, four in waves 4-7
    constexpr bool SPLIT18 = KTN == 18 && CT == 2;
    int koff[MAXT]; bool kt_live[MAXT];
    int kts[MAXT];
    bool ct_live[MAXT][CT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        int kt = wave + 8 * t;
        if (SPLIT18) kt = t < 2 ? 2 * wave + t : 16 + (wave >> 1);
        kts[t] = kt;
        kt_live[t] = SPLIT18 ? (t < 2 || wave < 4) : kt < KTN;
#pragma unroll
        for (int i = 0; i < CT; ++i) ct_live[t][i] = kt_live[t] && (!SPLIT18 || t < 2 || i == (wave & 1));
        const int k = (kt_live[t] ? kt : 0) * 32 + r;
        if (NCHW) { const int c = k / (TH * TW), kh = (k / TW) % TH, kw = k % TW; koff[t] = (c * PP + kh * Wp + kw) * 2; }
        else if (TR) {   // transpose read: this lane hands in channels cb + 4 (i % 4) .. + 3 of its group's 16, for pixel row i / 4
            const int kb = (kt_live[t] ? kt : 0) * 32 + ((lane >> 4) & 1) * 16, tap = kb / C, cb = kb % C;
            koff[t] = ((tap / TW) * Wb + (tap % TW)) * PS + (cb + (lane & 3) * 4) * 2;
        } else { const int tap = k / C, c = k % C; koff[t] = ((tap / TW) * Wb + (tap % TW)) * PS + c * 2; }
    }
    f32x16_t acc[MAXT][CT];
#pragma unroll
    for (int t = 0; t < MAXT; ++t)
#pragma unroll
        for (int i = 0; i < CT; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[t][i][e] = 0.f;
    float bsum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bsum[j] = 0.f;

    // ---- staging plan.  X: NHWC thread -> (band pixel, 8-channel chunk); NCHW thread -> (plane row, 4-float group).
    //      dY: thread -> (output pixel, 8-channel chunk), always channel chunk tid % (COUT/8).
    constexpr int XCPP = NCHW ? 1 : C / 8;
    constexpr int YCPP = COUT / 8;
    uint4 xpre[XCH]; uint4 ypre[YCH];
    // unit -> first frame n, first output row r0, output rows per frame R, staged input rows, output pixels
    auto unit_geom = [&](int unit, int& n, int& r0, int& R, int& rows, int& npix) {
        if (p.F > 1) {
            n = unit * p.F; r0 = 0; R = p.OH;
            const int fu = n + p.F <= p.Nimg ? p.F : p.Nimg - n;
            rows = (fu - 1) * p.H + (R - 1) * S + TH; npix = fu * R * p.OW;
        } else {
            n = unit / bands; const int b = unit % bands;
            r0 = b * p.R; R = (r0 + p.R <= p.OH) ? p.R : p.OH - r0; rows = (R - 1) * S + TH; npix = R * p.OW;
        }
    };
    auto stage_load = [&](int unit) {
        int n, r0, R, rows, npix; unit_geom(unit, n, r0, R, rows, npix);
        if (NCHW) {                                        // fp32 planes: item = 4 consecutive floats of one (c, row)
            // full-width bands: the band rows of one channel plane are contiguous in memory -> flat copy, item = 8 floats
            // (32 contiguous bytes per lane, every lane active: measured 1.3 - 1.7x faster than per-row maps)
            const int nflt = rows * p.W, items = (nflt + 7) / 8;
            if (C == 3 && p.u8) {                          // uint8 NHWC frames: one chunk = 8 elements of all three planes
                const int sx = p.shift ? p.shift[2 * n] : p.pad, sy = p.shift ? p.shift[2 * n + 1] : p.pad;
                const unsigned char* img = (const unsigned char*)p.X + (long)(p.fidx ? p.fidx[n] : n) * p.H * p.W * 3;
#pragma unroll
                for (int i = 0; i < XCH / 3; ++i) {
                    const int id = tid + i * NT;
                    const bool inb = id < items;
                    u8_band_chunk3(img, p.H, p.W, r0 * S, inb ? id * 8 : 0, inb ? nflt : 0, sx - p.pad, sy - p.pad,
                                   xpre[i], xpre[(XCH / 3 + i) % XCH], xpre[(2 * (XCH / 3) + i) % XCH]);
                }
            } else
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int c = j / (XCH / C), id = tid + (j % (XCH / C)) * NT;
                const bool inb = id < items, inb2 = inb && id * 8 + 8 <= nflt;
                const long base = (long)n * p.x_sn + (long)c * p.x_sc + (long)(r0 * S) * p.x_sy;
                const long off = base + (inb ? (long)id * 8 : 0);
                const float4 a = *(const float4*)((const float*)p.X + off);
                const float4 b = *(const float4*)((const float*)p.X + (inb2 ? off + 4 : off));
                xpre[j].x = inb ? pack_bf16x2(a.x, a.y) : 0u; xpre[j].y = inb ? pack_bf16x2(a.z, a.w) : 0u;
                xpre[j].z = inb2 ? pack_bf16x2(b.x, b.y) : 0u; xpre[j].w = inb2 ? pack_bf16x2(b.z, b.w) : 0u;
            }
        } else {
            const int npx = rows * Wb, cc = tid % XCPP;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int px = tid / XCPP + j * (NT / XCPP);
                const bool inb = px < npx;
                const long off = (long)n * p.x_sn + (long)(r0 * S) * p.x_sy + (inb ? (long)px * p.x_sx + cc * 8 : 0);
                uint4 v;
                if (PURE16 || p.x_dtype == HULC_BF16) v = *(const uint4*)((const uint16_t*)p.X + off);
                else {
                    const float4* q = (const float4*)((const float*)p.X + off);
                    const float4 a = q[0], b = q[1];
                    v.x = pack_bf16x2(a.x, a.y); v.y = pack_bf16x2(a.z, a.w); v.z = pack_bf16x2(b.x, b.y); v.w = pack_bf16x2(b.z, b.w);
                }
                xpre[j] = v;
            }
        }
        const int ycc = tid % YCPP;
#pragma unroll
        for (int j = 0; j < YCH; ++j) {
            const int q = tid / YCPP + j * (NT / YCPP);
            const bool inb = q < npix;
            const int qc = inb ? q : 0;
            // output pixels of a band (and of consecutive whole frames) are contiguous in dY: dy_sy == OW * dy_sx, dy_sn == OH * dy_sy
            const long off = (long)n * p.dy_sn + (long)r0 * p.dy_sy + (long)qc * p.dy_sx + ycc * 8;
            uint4 v;
            const float keep = inb ? 1.f : 0.f;
            if (PURE16 || p.dy_dtype == HULC_BF16) {
                // (the bias partial sums of bf16 gradients are taken in stage_store: touching the value here would wait for the load
                // — and every load issued before it — ahead of the MFMA loop it is meant to overlap)
                v = *(const uint4*)((const uint16_t*)p.dY + off);
            } else {
                const float4* qq = (const float4*)((const float*)p.dY + off);
                const float4 a = qq[0], b = qq[1];
                v.x = pack_bf16x2(a.x, a.y); v.y = pack_bf16x2(a.z, a.w); v.z = pack_bf16x2(b.x, b.y); v.w = pack_bf16x2(b.z, b.w);
                bsum[0] += keep * a.x; bsum[1] += keep * a.y; bsum[2] += keep * a.z; bsum[3] += keep * a.w;
                bsum[4] += keep * b.x; bsum[5] += keep * b.y; bsum[6] += keep * b.z; bsum[7] += keep * b.w;
            }
            ypre[j] = v;                                     // (pixels past npix are zeroed in stage_store, not here: see the note above)
        }
    };
    int po_R = -1, po_npix = -1;                              // the shape the patch-origin table was built for
    auto stage_store = [&](int unit) {
        int n, r0, R, rows, npix; unit_geom(unit, n, r0, R, rows, npix);
        if (NCHW) {
            const int nflt = rows * p.W, items = (nflt + 7) / 8;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int c = j / (XCH / C), id = tid + (j % (XCH / C)) * NT;
                if (id < items) *(uint4*)(xband + (c * PP + id * 8) * 2) = xpre[j];
            }
        } else {
            const int npx = rows * Wb, cc = tid % XCPP;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int px = tid / XCPP + j * (NT / XCPP);
                if (px < npx) *(uint4*)(xband + px * PS + cc * 16) = xpre[j];
            }
        }
        const int npad = (npix + 15) / 16 * 16, ycc = tid % YCPP;
#pragma unroll
        for (int j = 0; j < YCH; ++j) {
            const int q = tid / YCPP + j * (NT / YCPP);
            if (q < npad) {                                  // pixels in [npix, npad) carry zeros
                const bool inb = q < npix;
                const uint32_t w[4] = {inb ? ypre[j].x : 0u, inb ? ypre[j].y : 0u, inb ? ypre[j].z : 0u, inb ? ypre[j].w : 0u};
                if (PURE16 || p.dy_dtype == HULC_BF16) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) { bsum[2 * e] += __uint_as_float(w[e] << 16); bsum[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
                }
                if (TR) *(uint4*)(at + q * DROW + ycc * 16) = make_uint4(w[0], w[1], w[2], w[3]);
                else
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t bits = (uint16_t)((e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xffffu));
                    *(uint16_t*)(at + (ycc * 8 + e) * AT_ROW + q * 2) = bits;
                }
            }
        }
        // patch origin of every output pixel of the band: a function of the unit's shape only — rebuilt when the shape changes (with an even
        // grid a workgroup sees the same band of every frame: once per launch)
        if (R != po_R || npix != po_npix)
        for (int q = tid; q < npad; q += NT) {
            const int qc = q < npix ? q : npix - 1;
            const int ppf = R * p.OW, f = fast_div(qc, fast_rcp(ppf)), qq = qc - f * ppf;
            const int oy = fast_div(qq, inv_OW), ox = qq - oy * p.OW;
            pixoff[q] = NCHW ? ((oy * S) * Wp + ox * S) * 2 : ((f * p.H + oy * S) * Wb + ox * S) * PS;
        }
        po_R = R; po_npix = npix;
    };

    int unit = blockIdx.x;
    if (unit < nunits) { stage_load(unit); stage_store(unit); }
    __syncthreads();
    // (STAMP, HULC_WB_STAMPS: per-wave cycle sums of a unit's phases — issue of the next unit's loads | MFMA loop | barrier | wait for the
    //  loads | LDS stores | barrier; tools/study/wband_stamps.py)
    unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, t_units = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    for (; unit < nunits; unit += gridDim.x) {
        const int next = unit + gridDim.x;
        if (STAMP) c0 = __builtin_readcyclecounter();
        // (round 6, bf16 NHWC layers) the next unit's loads leave in two bursts: the second wave of every SIMD in front of the loop, the
        // first wave half way through it.  A load instruction of a wave is a kilobyte through the CU's one vector-memory path (~35 cycles each,
        // 120 of them per unit: tools/study/wband_stamps.py — 2300 cycles for the first wave of a SIMD, 4300 for the second, and the same
        // total when the loads are spread between the steps), and a wave queueing there issues no MFMAs: with all eight waves in one burst the
        // matrix pipes idle meanwhile (one workgroup per CU); staggered, the other wave of the SIMD has them.  HULC_WB_BURST=1: one burst.
        const bool pre = next < nunits, stagger = TR && !p.burst;
        int n, r0, R, rows, npix; unit_geom(unit, n, r0, R, rows, npix);
        const int nsteps = (npix + 15) / 16;
        // the step in front of which this wave issues its loads: every wave its own slot over the loop (the two waves of a SIMD half of it
        // apart) — a wave then pays for its own 15 loads, not for its place in a queue of 120 (conv2, 2048 frames: 139 us as one burst, 129 in
        // two half-workgroup bursts, 122 / 116 / 114 with the slots over 5/8, 7/8, all of the loop; the last wave's data still arrives in time)
        const int bat = stagger ? ((wave * ((nsteps * p.stag_num) >> 3)) >> 3) & ~1 : 0;
        if (pre && !stagger) stage_load(next);
        if (STAMP) c1 = __builtin_readcyclecounter();

        if (TR) {
            typedef short v4s __attribute__((ext_vector_type(4)));
            typedef v4s __attribute__((address_space(3))) * lds_v4s;
            auto tr = [](const char* a) -> v4s { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)a); };
            const int prow = (lane & 15) >> 2;                                  // pixel row of the 4 x 16 block this lane addresses
            const char* abase = at + prow * DROW + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
            // software-pipelined by hand, two steps per trip (fragment sets A / B alternate, nothing is copied): the patch offsets of a step are
            // read two steps ahead, its fragments one step ahead (their addresses need the offsets), its MFMAs last.  Left as
            // read offsets -> wait -> read fragments -> wait -> multiply, every step paid two LDS round trips with two waves per SIMD to hide them
            // (MFMA pipe 20 % busy).  Steps past the end re-read the last step (clamped), they are not multiplied.
            auto load_po = [&](int st, int& q0, int& q1) { const int m0 = st * 16 + h * 8; q0 = pixoff[m0 + prow]; q1 = pixoff[m0 + 4 + prow]; };
            auto load_fr = [&](int st, int q0, int q1, bf16x8_t (&a)[CT], bf16x8_t (&x)[MAXT]) {
                const int m0 = st * 16 + h * 8;                                 // this lane half's 8 pixels
#pragma unroll
                for (int i = 0; i < CT; ++i) {
                    union { v4s v[2]; bf16x8_t b; } f;
                    f.v[0] = tr(abase + m0 * DROW + i * 64); f.v[1] = tr(abase + (m0 + 4) * DROW + i * 64);
                    a[i] = f.b;
                }
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (!kt_live[t]) continue;                                  // wave-uniform
                    union { v4s v[2]; bf16x8_t b; } f;
                    f.v[0] = tr(xband + q0 + koff[t]); f.v[1] = tr(xband + q1 + koff[t]);
                    x[t] = f.b;
                }
            };
            auto mm = [&](const bf16x8_t (&a)[CT], const bf16x8_t (&x)[MAXT]) {
#pragma unroll
                for (int t = 0; t < MAXT; ++t) {
                    if (!kt_live[t]) continue;
#pragma unroll
                    for (int i = 0; i < CT; ++i)
                        if (ct_live[t][i]) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], x[t], acc[t][i], 0, 0, 0);
                }
            };
            // (conv2's geometry only: with conv3's three k tiles per wave the alternating sets cost registers and tail re-reads, measured 5 % slower)
            constexpr bool PIPE = (C == 32);
            if (!PIPE) {
                auto one_step = [&](int st) {
                    const int m0 = st * 16 + h * 8;                             // this lane half's 8 pixels
                    const int po0 = pixoff[m0 + prow], po1 = pixoff[m0 + 4 + prow];
                    bf16x8_t a[CT];
#pragma unroll
                    for (int i = 0; i < CT; ++i) {
                        union { v4s v[2]; bf16x8_t b; } f;
                        f.v[0] = tr(abase + m0 * DROW + i * 64); f.v[1] = tr(abase + (m0 + 4) * DROW + i * 64);
                        a[i] = f.b;
                    }
#pragma unroll
                    for (int t = 0; t < MAXT; ++t) {
                        if (!kt_live[t]) continue;                              // wave-uniform
                        union { v4s v[2]; bf16x8_t b; } x;
                        x.v[0] = tr(xband + po0 + koff[t]); x.v[1] = tr(xband + po1 + koff[t]);
#pragma unroll
                        for (int i = 0; i < CT; ++i)
                            if (ct_live[t][i]) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], x.b, acc[t][i], 0, 0, 0);
                    }
                };
                for (int st = 0; st < nsteps; ++st) {
                    if (pre && stagger && st == bat) stage_load(next);
                    one_step(st);
                }
            } else {
            const int last = nsteps - 1;
            int pa0, pa1, pb0, pb1;
            bf16x8_t aA[CT], xA[MAXT], aB[CT], xB[MAXT];
            load_po(0, pa0, pa1); load_po(last < 1 ? last : 1, pb0, pb1);
            load_fr(0, pa0, pa1, aA, xA);
            auto trip = [&](int st) {
                load_po(st + 2 < last ? st + 2 : last, pa0, pa1);
                load_fr(st + 1 < last ? st + 1 : last, pb0, pb1, aB, xB);
                __builtin_amdgcn_sched_barrier(0);
                mm(aA, xA);
                __builtin_amdgcn_sched_barrier(0);
                if (st + 1 < nsteps) {
                    load_po(st + 3 < last ? st + 3 : last, pb0, pb1);
                    load_fr(st + 2 < last ? st + 2 : last, pa0, pa1, aA, xA);
                    __builtin_amdgcn_sched_barrier(0);
                    mm(aB, xB);
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            for (int st = 0; st < nsteps; st += 2) {
                if (pre && stagger && st == bat) stage_load(next);
                trip(st);
            }
            }
        } else
        for (int s = 0; s < nsteps; ++s) {
            const int m0 = s * 16 + h * 8;                   // this lane half's 8 pixels
            const int4 po0 = *(const int4*)(pixoff + m0), po1 = *(const int4*)(pixoff + m0 + 4);
            const int po[8] = {po0.x, po0.y, po0.z, po0.w, po1.x, po1.y, po1.z, po1.w};
            bf16x8_t a[CT];
#pragma unroll
            for (int i = 0; i < CT; ++i) a[i] = *(const bf16x8_t*)(at + (i * 32 + r) * AT_ROW + m0 * 2);
#pragma unroll
            for (int t = 0; t < MAXT; ++t) {
                if (!kt_live[t]) continue;                   // wave-uniform
                union { uint32_t w[4]; bf16x8_t b; } x;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint32_t lo = *(const uint16_t*)(xband + po[2 * e] + koff[t]);
                    const uint32_t hi = *(const uint16_t*)(xband + po[2 * e + 1] + koff[t]);
                    x.w[e] = lo | (hi << 16);
                }
#pragma unroll
                for (int i = 0; i < CT; ++i)
                    if (ct_live[t][i]) acc[t][i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], x.b, acc[t][i], 0, 0, 0);
            }
        }
        if (STAMP) c2 = __builtin_readcyclecounter();
        __syncthreads();
        if (STAMP) { c3 = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c4 = __builtin_readcyclecounter(); }
        if (next < nunits) stage_store(next);
        if (STAMP) c5 = __builtin_readcyclecounter();
        __syncthreads();
        if (STAMP) {
            const unsigned long long c6 = __builtin_readcyclecounter();
            t_ph[0] += c1 - c0; t_ph[1] += c2 - c1; t_ph[2] += c3 - c2; t_ph[3] += c4 - c3; t_ph[4] += c5 - c4; t_ph[5] += c6 - c5; t_units += 1;
        }
    }
    if (STAMP && stamps && lane == 0) {
        unsigned long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 7;
#pragma unroll
        for (int e = 0; e < 6; ++e) o[e] = t_ph[e];
        o[6] = t_units;
    }

    // ---- slabs: dW partial [COUT][K] (lane = k column, register = channel row) and the bias partial
    float* pw = p.partial_w + (long)blockIdx.x * COUT * K;
#pragma unroll
    for (int t = 0; t < MAXT; ++t) {
        if (!kt_live[t]) continue;
        const int k = kts[t] * 32 + r;
#pragma unroll
        for (int i = 0; i < CT; ++i) {
            if (!ct_live[t][i]) continue;
#pragma unroll
            for (int e = 0; e < 16; ++e) pw[(long)(i * 32 + acc_row(e, lane)) * K + k] = acc[t][i][e];
        }
    }
    if (p.partial_b) {
        float* red = (float*)smem;                            // bands are dead: reuse LDS, [NT][8] floats = 16 KB
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[tid * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < COUT) {
            const int ycc = tid / 8, e = tid % 8;
            float sacc = 0.f;
            for (int q = ycc; q < NT; q += YCPP) sacc += red[q * 8 + e];   // threads with tid % YCPP == ycc, fixed order
            p.partial_b[(long)blockIdx.x * COUT + tid] = sacc;
        }
    }
}

// perm_c > 0: forward k order (tap, c) -> the parameter's OIHW order (c, tap); accumulate: add to the destination (gradient arena)
// the workgroups past the weight rows (blockIdx.x >= nbw, optional) sum the bias partials: one launch for dW and db
__global__ __launch_bounds__(1024) void wband_reduce_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, long R, int accumulate,
                                                            int perm_c, int perm_taps, int nbw = 1 << 30, const float* __restrict__ partial_b = nullptr,
                                                            float* __restrict__ out_b = nullptr, long Rb = 0) {
    __shared__ float red[16][64];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    int bx = blockIdx.x;
    if (bx >= nbw) { bx -= nbw; partial = partial_b; out = out_b; R = Rb; perm_c = 0; }
    const long rr = (long)bx * 64 + lane;
    const int per = (P + 15) / 16;
    const int q0 = sl * per, q1 = q0 + per < P ? q0 + per : P;
    float s = 0.f;
    if (rr < R) {
        // four independent loads per trip: with one load per trip every partial waited for its own memory round trip (12 us per launch)
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int q = q0;
        for (; q + 3 < q1; q += 4) {
            const float a = partial[(long)q * R + rr], b = partial[(long)(q + 1) * R + rr], c = partial[(long)(q + 2) * R + rr],
                        d = partial[(long)(q + 3) * R + rr];
            s0 += a; s1 += b; s2 += c; s3 += d;
        }
        for (; q < q1; ++q) s0 += partial[(long)q * R + rr];
        s = (s0 + s1) + (s2 + s3);
    }
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && rr < R) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        long o = rr;
        if (perm_c > 0) { const long K = (long)perm_c * perm_taps, co = rr / K, k = rr % K; o = co * K + (k % perm_c) * perm_taps + k / perm_c; }
        out[o] = accumulate ? out[o] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// conv1 (3 -> 32 channels, 8x8 stride 4; fp32 NCHW frames or uint8 NHWC frames): phase-plane bands.
// With channel planes as they are in memory, the B fragment of tap (c, kh, kw) — 8 consecutive output pixels, i.e. input columns
// 4*ox + kw — is eight 2-byte LDS reads 8 bytes apart, and with the pixel-offset table the MFMA loop issued 11 LDS instructions per
// MFMA: the kernel was LDS-issue-bound at ~3 TB/s of frames.  Here the band is de-interleaved while it is staged: plane phi of a row
// holds columns phi, phi + 4, phi + 8, ... so that 8 consecutive output pixels of tap kw are the 8 consecutive elements
// j = ox + kw / 4 of plane kw % 4.  Output pixels are enumerated per row in blocks of 8 (row padded to OWP = 8 * ceil(OW / 8) slots, dY^T
// zero in the padding), a block's fragment is ONE aligned ds_read_b128.  The taps kw >= 4 read plane kw - 4 one element further:
// dW[kw + 4] = sum_ox dY[ox] * P[ox + 1] = sum_ox' dY[ox' - 1] * P[ox'] — so instead of realigning the X fragment (round 4: one more
// ds_read_b32 and four v_alignbit per MFMA) a wave holds 32 taps of ONE half (kw < 4 or kw >= 4) and the upper-half waves read their dY^T
// fragment ONE SLOT EARLIER (the slot in front of a row's first pixel is the previous row's padding, or the guard slot in front of the
// array: zero).  Plane stride PSTR (multiple of 16 B with an odd number of 16-byte slots) makes the 16 (kh, phi) addresses of a
// ds_read_b128 lane group fall on 16 different bank quads.  The bias gradient is the seventh wave's MFMA chain: the same dY^T
// fragments against a fragment of ones (wave 6 was idle in the loop; as per-thread sums in stage_store it cost 32 VALU per thread and unit).
struct W1P {
    const void* X; const void* dY; int dy_dtype;
    int Nimg, H, W, OH, OW, OWP, R, PSTR, AT_ROW;
    long dy_sn, dy_sy, dy_sx;
    float* partial_w; float* partial_b;
    int u8, pad; const int* shift; const int* fidx;
    int dbg;
    const void* X2; int nsplit;             // fp32 frames: frames n >= nsplit come from X2 (pre-offset by -nsplit frames); X2 == X when unused
    const void* const* xs; const void* const* xs2;   // optional device slots holding the frame tensors' addresses (see conv1_band.hip)
};

// U8: uint8 NHWC frames (else fp32 NCHW planes); dY is bf16 (other gradients dtypes take the generic kernel).  Both are compile-time so
// that no join of two load paths makes the compiler wait for the prefetch early.
template <int XCH, int YCH, bool U8, bool STAMP = false>
__global__ __launch_bounds__(512, 4) void conv1_wgrad_kernel(W1P p, unsigned long long* stamps = nullptr) {
    constexpr int NT = 512, C = 3, S = 4, K = 192, KTN = 6, YCPP = 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bands = (p.OH + p.R - 1) / p.R;
    const float inv_OW = fast_rcp(p.OW), inv_W = fast_rcp(p.W);
    // (round 5, ABI 5: hulc_conv_desc.x_slot / x2_slot) the frame tensors' base addresses read from DEVICE slots at kernel start: a captured
    // hipGraph then follows whatever batch the caller points the slots at — the step node updates two pointers instead of copying 1.16 GB of
    // frames into the graph's input buffers.  One scalar-valued load per slot, once per workgroup, in front of the first band's loads.
    const float* xbase = (const float*)p.X;
    const float* xbase2 = (const float*)p.X2;
    if (!U8 && p.xs) {
        const unsigned long long a = (unsigned long long)*p.xs;
        xbase = (const float*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(a >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)a));
        xbase2 = xbase;
        if (p.xs2) {
            const unsigned long long b = (unsigned long long)*p.xs2;
            xbase2 = (const float*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b))
                     - (long)p.nsplit * 3 * p.H * p.W;
        }
    }
    // a workgroup walks whole frames (blockIdx, blockIdx + grid, ...), band after band: the 4 halo rows a band shares with its
    // predecessor were read by this CU a moment ago and come back from L2, so small bands (few staging registers) cost no HBM traffic
    const int nunits = ((p.Nimg - blockIdx.x + gridDim.x - 1) / gridDim.x) * bands;     // this workgroup's units
    const int rows_max = (p.R - 1) * S + 8;
    const int nbx = p.OWP / 8;                              // pixel blocks per output row
    // dY in its natural layout, one 96-byte row (32 channels + pad) per pixel slot; the A fragment (lane = channel, 8 consecutive slots) is two
    // ds_read_b64_tr_b16 (see the generic kernel above): no 2-byte transposing scatter while staging
    constexpr int DR1 = 96;
    const int nslots = p.R * p.OWP;
    char* xband = smem;                                     // [c][row][phi][PSTR bytes]
    const int xbytes = C * rows_max * 4 * p.PSTR;
    char* at = smem + xbytes + DR1;                         // [nslots][DR1] behind one guard slot (zero: the upper-half waves' slot -1)
    const int ones_off = xbytes + (nslots + 9) * DR1;       // 32 bytes of bf16 1.0: the bias wave's "X fragment"

    // this lane's tap: wave = (channel, half of kw), lane r = (kh, phi): k = (c, kh, kw = phi + 4 * half).  Wave 6 = the bias chain.
    const bool live = wave < KTN, biasw = wave == KTN && p.partial_b != nullptr;
    const int kc = (live ? wave : 0) >> 1, hi = live ? (wave & 1) : 0, kh = r >> 2, phi = r & 3;
    const int k = kc * 64 + kh * 8 + phi + 4 * hi;
    const int koff = biasw ? ones_off : ((kc * rows_max + kh) * 4 + phi) * p.PSTR;
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    // zero once: the guard slot, the dY^T padding slots and the plane elements past W / 4 are read (times zero / as never-used taps) but never written
    for (int o = tid * 16; o < xbytes + (nslots + 1) * DR1; o += NT * 16) *(uint4*)(smem + o) = make_uint4(0, 0, 0, 0);
    if (tid < 2) *(uint4*)(smem + ones_off + tid * 16) = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);

    // prefetched data stays RAW in registers (fp32 frames: 8 floats per item; uint8 frames: the aligned dword windows) and is
    // converted in stage_store: any ALU use at load time would put the wait for the loads in front of the MFMA loop they overlap
    float4 xraw[XCH][2]; uint4 ypre[YCH];
    auto unit_geom = [&](int unit, int& n, int& r0, int& R, int& rows) {
        const int b = unit % bands;
        n = blockIdx.x + (unit / bands) * gridDim.x;
        r0 = b * p.R; R = (r0 + p.R <= p.OH) ? p.R : p.OH - r0; rows = (R - 1) * S + 8;
    };
    // (uint8 frames) per-frame parameters — augmentation shift, frame index — of this workgroup's first MAXU units, read once into LDS (see
    // conv1_band.hip: as global loads they were two dependent round trips to memory in front of every unit's prefetch)
    constexpr int MAXU = 256;
    int4* ftab = (int4*)(smem + xbytes + (p.R * p.OWP + 9) * 96 + 64);
    auto frame_params = [&](int unit, int n, int& sx, int& sy, int& fi) {
        if (unit < MAXU) {
            const int4 e = ftab[unit];
            sx = __builtin_amdgcn_readfirstlane(e.x); sy = __builtin_amdgcn_readfirstlane(e.y); fi = __builtin_amdgcn_readfirstlane(e.z);
        } else {
            sx = p.shift ? p.shift[2 * n] : p.pad; sy = p.shift ? p.shift[2 * n + 1] : p.pad; fi = p.fidx ? p.fidx[n] : n;
        }
    };
    if (U8) {
        for (int u = tid; u < (nunits < MAXU ? nunits : MAXU); u += NT) {
            int n, r0, R, rows; unit_geom(u, n, r0, R, rows);
            ftab[u] = make_int4(p.shift ? p.shift[2 * n] : p.pad, p.shift ? p.shift[2 * n + 1] : p.pad, p.fidx ? p.fidx[n] : n, 0);
        }
        __syncthreads();
    }
    // thread-only parts of the load addresses: the item's float index inside the band of plane c, the gradient element of pixel q
    unsigned xld[C], yld[YCH];
#pragma unroll
    for (int c = 0; c < C; ++c) xld[c] = (unsigned)(c * p.H * p.W + tid * 8) * 4u;                       // (bytes)
#pragma unroll
    for (int j = 0; j < YCH; ++j) yld[j] = (unsigned)((tid / YCPP + j * (NT / YCPP)) * (int)p.dy_sx + (tid % YCPP) * 8) * 2u;
    // a unit's geometry is carried from unit to unit (next band of the frame, or the first band of the workgroup's next frame): as
    // unit / bands and unit % bands it was two scalar division sequences in front of every unit's loads
    struct Geo { int n, r0, R, rows; };
    auto geo_next = [&](const Geo& g) -> Geo {
        const int b1 = g.r0 + p.R;
        const bool wrapf = b1 >= p.OH;
        Geo o;
        o.n = wrapf ? g.n + (int)gridDim.x : g.n;
        o.r0 = wrapf ? 0 : b1;
        o.R = (o.r0 + p.R <= p.OH) ? p.R : p.OH - o.r0;
        o.rows = (o.R - 1) * S + 8;
        return o;
    };
    auto stage_load = [&](int unit, const Geo& g) {
        const int n = g.n, r0 = g.r0, R = g.R, rows = g.rows;
        const int nflt = rows * p.W, items = (nflt + 7) / 8;
        if (U8) {                                          // uint8 NHWC frames: item = 8 elements of all three planes = 2 x (4 aligned dwords)
            int sx, sy, fi; frame_params(unit, n, sx, sy, fi);
            const unsigned char* img = (const unsigned char*)(n < p.nsplit ? p.X : p.X2) + (long)fi * p.H * p.W * 3;   // (n is uniform: a scalar select)
#pragma unroll
            for (int i = 0; i < XCH / 3; ++i) {
                const int id = tid + i * NT;
                const bool inb = id < items;
                uint32_t raw[8];
                u8_band_chunk3_load(img, p.H, p.W, r0 * S, inb ? id * 8 : 0, inb ? nflt : 0, sx - p.pad, sy - p.pad, raw);
                xraw[i][0] = make_float4(__uint_as_float(raw[0]), __uint_as_float(raw[1]), __uint_as_float(raw[2]), __uint_as_float(raw[3]));
                xraw[i][1] = make_float4(__uint_as_float(raw[4]), __uint_as_float(raw[5]), __uint_as_float(raw[6]), __uint_as_float(raw[7]));
            }
        } else if (XCH == C) {
            // fp32 planes: the band rows of a channel are contiguous -> flat copy, 8 floats per item (rows * W is a multiple of 16: no partial
            // item).  Address = uniform base of the band (scalar registers) + a 32-bit offset that depends on the thread only: one load
            // instruction each, no vector address arithmetic (as 64-bit per-lane addresses the eight loads took ~100 instructions to issue)
            const float* ub = (n < p.nsplit ? xbase : xbase2) + ((long)n * C * p.H * p.W + (long)(r0 * S) * p.W);
            typedef float f32x4n __attribute__((ext_vector_type(4)));
            typedef const f32x4n __attribute__((address_space(1))) * gvec;
            // (measured: with the loads of the four halo rows a band shares with its predecessor — L2 hits, a fifth of the requests — switched
            //  off, the kernel is 2 % faster: a row ring in LDS that keeps them is not worth building)
            const bool inb = tid < items;
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const unsigned o = inb ? xld[j] : (unsigned)(j * p.H * p.W) * 4u;     // BYTE offset (zero-extended: the scalar-base addressing form)
                const f32x4n v0 = *(gvec)((const char*)ub + o), v1 = *(gvec)((const char*)ub + o + 16);
                xraw[j][0] = make_float4(v0.x, v0.y, v0.z, v0.w);
                xraw[j][1] = make_float4(v1.x, v1.y, v1.z, v1.w);
            }
        } else
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int c = j / (XCH / C), id = tid + (j % (XCH / C)) * NT;
            const bool inb = id < items, inb2 = inb && id * 8 + 8 <= nflt;
            const long off = ((long)n * C + c) * p.H * p.W + (long)(r0 * S) * p.W + (inb ? (long)id * 8 : 0);
            const float* xb = n < p.nsplit ? xbase : xbase2;                   // (n is uniform: a scalar select)
            // (global address space spelled out: a base address read from a device slot is an integer and a pointer made of one is a FLAT
            //  pointer; flat loads count against lgkmcnt as well as vmcnt.  Measured: no difference here — the loads have landed by then)
            typedef float f32x4n __attribute__((ext_vector_type(4)));
            typedef const f32x4n __attribute__((address_space(1))) * gvec;
            const f32x4n v0 = *(gvec)(xb + off), v1 = *(gvec)(xb + (inb2 ? off + 4 : off));
            xraw[j][0] = make_float4(v0.x, v0.y, v0.z, v0.w);
            xraw[j][1] = make_float4(v1.x, v1.y, v1.z, v1.w);
        }
        const int npix = R * p.OW;
        // the output pixels of a band are contiguous in dY (dy_sy == OW * dy_sx): uniform base + the thread's element offset
        const uint16_t* yb = (const uint16_t*)p.dY + ((long)n * p.dy_sn + (long)r0 * p.dy_sy);
#pragma unroll
        for (int j = 0; j < YCH; ++j) {
            typedef unsigned u32x4n __attribute__((ext_vector_type(4)));
            typedef const u32x4n __attribute__((address_space(1))) * gvec;
            const int q = tid / YCPP + j * (NT / YCPP);
            const u32x4n v = *(gvec)((const char*)yb + (q < npix ? yld[j] : (unsigned)((tid % YCPP) * 16)));
            ypre[j] = make_uint4(v.x, v.y, v.z, v.w);
        }
    };
    // thread-only parts of the staging addresses (fp32 frames, XCH == C: a thread's item is id == tid in every channel plane)
    int xdst;
    { const int e0 = tid * 8, row = fast_div(e0, inv_W), col = e0 - row * p.W; xdst = row * 4 * p.PSTR + (col >> 2) * 2; }
    int ydst[YCH];
#pragma unroll
    for (int j = 0; j < YCH; ++j) {
        const int q = tid / YCPP + j * (NT / YCPP), oy = fast_div(q, inv_OW);
        ydst[j] = (oy * p.OWP + (q - oy * p.OW)) * DR1 + (tid % YCPP) * 16;
    }
    auto stage_store = [&](int unit, const Geo& g) {
        const int n = g.n, r0 = g.r0, R = g.R, rows = g.rows;
        const int nflt = rows * p.W, items = (nflt + 7) / 8;
        const bool w8 = (p.W & 7) == 0;                    // chunks never straddle a row and start at an even plane index
        uint4 xpre[XCH];
        if (!U8 && XCH == C && w8) {
            // plane phi takes columns (col + phi, col + 4 + phi) as one dword: exactly v_cvt_pk_bf16_f32 of the item's floats phi and 4 + phi —
            // four conversions and four 4-byte LDS stores per item, at an offset that depends on the thread only (xdst, taken once)
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                if (tid >= items) continue;
                const float4 a = xraw[j][0], b = xraw[j][1];
                char* dst = xband + j * rows_max * 4 * p.PSTR + xdst;
                *(uint32_t*)(dst) = pack_bf16x2(a.x, b.x);
                *(uint32_t*)(dst + p.PSTR) = pack_bf16x2(a.y, b.y);
                *(uint32_t*)(dst + 2 * p.PSTR) = pack_bf16x2(a.z, b.z);
                *(uint32_t*)(dst + 3 * p.PSTR) = pack_bf16x2(a.w, b.w);
            }
        } else {
        if (U8) {
            int sx, sy, fi; frame_params(unit, n, sx, sy, fi);
#pragma unroll
            for (int i = 0; i < XCH / 3; ++i) {
                const int id = tid + i * NT;
                const bool inb = id < items;
                const uint32_t raw[8] = {__float_as_uint(xraw[i][0].x), __float_as_uint(xraw[i][0].y), __float_as_uint(xraw[i][0].z), __float_as_uint(xraw[i][0].w),
                                         __float_as_uint(xraw[i][1].x), __float_as_uint(xraw[i][1].y), __float_as_uint(xraw[i][1].z), __float_as_uint(xraw[i][1].w)};
                u8_band_chunk3_convert(p.W, inb ? id * 8 : 0, inb ? nflt : 0, sx - p.pad, raw, xpre[i], xpre[XCH / 3 + i], xpre[2 * (XCH / 3) + i]);
            }
        } else {
#pragma unroll
            for (int j = 0; j < XCH; ++j) {
                const int id = tid + (j % (XCH / C)) * NT;
                const bool inb2 = id * 8 + 8 <= nflt;
                const float4 a = xraw[j][0], b = xraw[j][1];
                xpre[j] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), inb2 ? pack_bf16x2(b.x, b.y) : 0u, inb2 ? pack_bf16x2(b.z, b.w) : 0u);
            }
        }
#pragma unroll
        for (int j = 0; j < XCH; ++j) {
            const int c = j / (XCH / C), id = tid + (j % (XCH / C)) * NT;
            if (id >= items) continue;
            const int e0 = id * 8, row = fast_div(e0, inv_W), col = e0 - row * p.W;
            const uint32_t w[4] = {xpre[j].x, xpre[j].y, xpre[j].z, xpre[j].w};          // columns col .. col + 7, two per dword
            char* dst = xband + (long)((c * rows_max + row) * 4) * p.PSTR + (col >> 2) * 2;
            if (w8) {                                      // plane phi gets columns col + phi and col + 4 + phi: elements j, j + 1
                *(uint32_t*)(dst) = (w[0] & 0xffffu) | (w[2] << 16);
                *(uint32_t*)(dst + p.PSTR) = (w[0] >> 16) | (w[2] & 0xffff0000u);
                *(uint32_t*)(dst + 2 * p.PSTR) = (w[1] & 0xffffu) | (w[3] << 16);
                *(uint32_t*)(dst + 3 * p.PSTR) = (w[1] >> 16) | (w[3] & 0xffff0000u);
            } else {                                       // W % 8 == 4: the second quad may be the start of the next row
                *(uint16_t*)(dst) = (uint16_t)w[0]; *(uint16_t*)(dst + p.PSTR) = (uint16_t)(w[0] >> 16);
                *(uint16_t*)(dst + 2 * p.PSTR) = (uint16_t)w[1]; *(uint16_t*)(dst + 3 * p.PSTR) = (uint16_t)(w[1] >> 16);
                if (e0 + 4 < nflt) {
                    const bool wrap = col + 4 >= p.W;
                    char* d2 = wrap ? xband + (long)((c * rows_max + row + 1) * 4) * p.PSTR : dst + 2;
                    *(uint16_t*)(d2) = (uint16_t)w[2]; *(uint16_t*)(d2 + p.PSTR) = (uint16_t)(w[2] >> 16);
                    *(uint16_t*)(d2 + 2 * p.PSTR) = (uint16_t)w[3]; *(uint16_t*)(d2 + 3 * p.PSTR) = (uint16_t)(w[3] >> 16);
                }
            }
        }
        }
        const int npix = R * p.OW, ycc = tid % YCPP;
        // an odd number of 8-slot blocks: the MFMA loop's last step pairs the last block with the 8 slots behind it — zeros (after a partial band
        // they would hold the previous band's next row; behind a full band they are the padding slots, zero anyway)
        if (((R * nbx) & 1) && tid < 8 * DR1 / 16) *(uint4*)(at + R * p.OWP * DR1 + tid * 16) = make_uint4(0, 0, 0, 0);
#pragma unroll
        for (int j = 0; j < YCH; ++j) {
            const int q = tid / YCPP + j * (NT / YCPP);
            if (q < npix) *(uint4*)(at + ydst[j]) = ypre[j];     // (slot offset of pixel q: a function of the thread, taken once)
        }
    };

    int unit = 0;
    __syncthreads();                                         // the zero fill precedes the first stage_store
    Geo gc;
    unit_geom(0, gc.n, gc.r0, gc.R, gc.rows);
    if (unit < nunits) { stage_load(unit, gc); stage_store(unit, gc); }
    __syncthreads();
    // (STAMP, HULC_W1_STAMPS: per-wave cycle sums of a unit's phases — issue of the next band's loads | MFMA loop | barrier | wait for the
    //  loads | convert + LDS stores | barrier)
    unsigned long long t_ph[6] = {0, 0, 0, 0, 0, 0}, t_units = 0, c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0;
    for (; unit < nunits; ++unit) {
        const int next = unit + 1;
        if (STAMP) c0 = __builtin_readcyclecounter();
        const Geo gn = geo_next(gc);
        if (next < nunits && !(p.dbg & 8)) stage_load(next, gn);
        if (STAMP) c1 = __builtin_readcyclecounter();

        const int R = gc.R;
        if ((live || biasw) && !(p.dbg & 1)) {
            const int nblk = R * nbx, nsteps = (nblk + 1) / 2;
            // this lane half's pixel block b = 2 * s + h as (row, column bx) — only its X offset `xo` is kept: + 32 bytes per step, one row of
            // planes further when the column wraps.  An odd block count leaves the last step's second block empty: its dY^T slots are zero
            // (stage_store clears them), so its X fragment may be anything finite in LDS.  (The MFMA loop is bound by instruction issue: this
            // form spends ~11 VALU per MFMA where block validity selects and a multiply per address spent 21.)
            // (the bias wave reads the 32 bytes of ones at every step: no walk)
            int bx = h, xo = koff + h * 16;
            int xwrap = biasw ? 0 : S * 4 * p.PSTR - nbx * 16, xstep = biasw ? 0 : 32;
            asm volatile("" : "+v"(xwrap), "+v"(xstep));     // (kept in vector registers)
            // the upper-half taps pair plane element ox' with dY slot ox' - 1: their dY^T fragment starts one slot earlier
            const char* ab = at + (h * 8 + ((lane & 15) >> 2) - hi) * DR1 + (((lane >> 4) & 1) * 16 + (lane & 3) * 4) * 2;
            // one step = 16 pixel slots: dY^T fragment (two ds_read_b64_tr_b16), X fragment (one ds_read_b128), one MFMA
            auto fetch = [&](int st, uint4& a, uint4& w) {
                {
                    typedef short v4s __attribute__((ext_vector_type(4)));
                    typedef v4s __attribute__((address_space(3))) * lds_v4s;
                    union { v4s v[2]; uint4 u; } f;
                    f.v[0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)ab);
                    f.v[1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_v4s)(__attribute__((address_space(3))) char*)(ab + 4 * DR1));
                    a = f.u;
                    ab += 16 * DR1;
                }
                w = *(const uint4*)(xband + xo);
                const int t = bx + 2 - nbx;                  // (selects between vector registers only: a select with a scalar operand next to the
                const bool wrap = t >= 0;                    //  condition mask costs a move per step on this target)
                bx = wrap ? t : bx + 2;
                xo += xstep + (wrap ? xwrap : 0);
            };
            auto mma = [&](f32x16_t& c, const uint4& a, const uint4& w) {
                union { uint4 u; bf16x8_t b; } af, x; af.u = a; x.u = w;
                c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af.b, x.b, c, 0, 0, 0);
            };
            // software-pipelined: the next step's fragments are in flight while this step's MFMA issues (one accumulator: a second
            // chain or deeper prefetch spills at the 128-VGPR budget of two workgroups per CU)
            if (nsteps > 0) {                                // (two steps per trip: the fragment registers alternate instead of being copied)
                // fragments TWO steps ahead (three register sets, three steps per trip): with one step of distance a step cost the LDS round
                // trip (~140 cycles against the MFMA's 32).  Steps past the end read the zeroed slots behind the band / stay inside the
                // band's planes (finite), and are not multiplied
                uint4 a0, w0, a1, w1, a2, w2;
                fetch(0, a0, w0);
                fetch(1, a1, w1);
                int st = 0;
                for (; st + 3 <= nsteps; st += 3) {
                    fetch(st + 2, a2, w2);
                    mma(acc, a0, w0);
                    fetch(st + 3, a0, w0);
                    mma(acc, a1, w1);
                    fetch(st + 4, a1, w1);
                    mma(acc, a2, w2);
                }
                if (st < nsteps) mma(acc, a0, w0);
                if (st + 1 < nsteps) mma(acc, a1, w1);
            }
        }
        if (STAMP) c2 = __builtin_readcyclecounter();
        __syncthreads();
        if (STAMP) { c3 = __builtin_readcyclecounter(); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); c4 = __builtin_readcyclecounter(); }
        if (next < nunits) stage_store(next, gn);
        if (STAMP) c5 = __builtin_readcyclecounter();
        __syncthreads();
        gc = gn;
        if (STAMP) {
            const unsigned long long c6 = __builtin_readcyclecounter();
            t_ph[0] += c1 - c0; t_ph[1] += c2 - c1; t_ph[2] += c3 - c2; t_ph[3] += c4 - c3; t_ph[4] += c5 - c4; t_ph[5] += c6 - c5; t_units += 1;
        }
    }
    if (STAMP && stamps && lane == 0) {
        unsigned long long* o = stamps + ((long)blockIdx.x * 8 + wave) * 7;
#pragma unroll
        for (int e = 0; e < 6; ++e) o[e] = t_ph[e];
        o[6] = t_units;
    }

    // ---- slabs: dW partial [32][K] (lane = k column, register = channel row) and the bias partial
    if (live) {
        float* pw = p.partial_w + (long)blockIdx.x * 32 * K;
#pragma unroll
        for (int e = 0; e < 16; ++e) pw[(long)acc_row(e, lane) * K + k] = acc[e];
    }
    if (biasw && r == 0) {                                    // every column of the bias wave's tile holds the channel sums: lanes 0 and 32 cover the rows
#pragma unroll
        for (int e = 0; e < 16; ++e) p.partial_b[(long)blockIdx.x * 32 + acc_row(e, lane)] = acc[e];
    }
}

int launch_conv1_wgrad(W1P& p, float* dw, float* db, void* ws, long ws_bytes, int accumulate, hipStream_t s) {
    constexpr int XCH = 3, YCH = 2, K = 192;
    if (p.W % 4) return -1;
    p.OWP = (p.OW + 7) / 8 * 8;
    if (p.OWP < 16) return -1;                               // (the MFMA loop's column walk wraps at most once per step: two 8-pixel blocks per row at least)
    p.PSTR = ((p.OWP + 2) * 2 + 15) / 16 * 16;
    if (((p.PSTR / 16) & 1) == 0) p.PSTR += 16;
    auto at_row = [&](int R) -> int { int a = R * p.OWP * 2 + 16; if (((a / 16) & 1) == 0) a += 16; return a; };
    auto lds_of = [&](int R) -> long { return 3L * ((R - 1) * 4 + 8) * 4 * p.PSTR + (long)(R * p.OWP + 9) * 96 + 64 + (p.u8 ? 256 * 16 : 0); };
    auto fits = [&](int R) -> bool {
        const long rows = (R - 1) * 4 + 8;
        return lds_of(R) <= (160 * 1024 - 256) / 2 && (rows * p.W + 7) / 8 * 3 <= (long)XCH * 512 && (long)R * p.OW * 4 <= (long)YCH * 512;
    };
    int R = p.OH;
    while (R > 1 && !fits(R)) --R;
    if (!fits(R)) return -1;
    const int bands = (p.OH + R - 1) / R;
    R = (p.OH + bands - 1) / bands;
    p.R = R; p.AT_ROW = at_row(R);
    // two workgroups per CU (LDS <= 80 KB each).  HULC_CONV1_SLOTS (tests): fewer, so that a workgroup walks more than the 256 units its LDS
    // table of per-frame parameters holds and the direct loads behind the table are exercised
    const int slots = getenv("HULC_CONV1_SLOTS") && atoi(getenv("HULC_CONV1_SLOTS")) > 0 ? atoi(getenv("HULC_CONV1_SLOTS")) : 512;
    const int per = (p.Nimg + slots - 1) / slots;            // frames per workgroup
    const int grid = (p.Nimg + per - 1) / per;
    if ((long)grid * 32 * (K + 1) * 4 > ws_bytes) return -1;
    p.partial_w = (float*)ws;
    p.partial_b = db ? p.partial_w + (long)grid * 32 * K : nullptr;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv1_wgrad_kernel<XCH, YCH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv1_wgrad_kernel<XCH, YCH, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return -2;
        attr_set = true;
    }
    // HULC_W1_STAMPS=<device address of grid x 8 x 7 uint64>: the instrumented instance (tools/study/conv1_stamps.py)
    const char* se = getenv("HULC_W1_STAMPS");
    if (se && *se && !p.u8) {
        static bool stamp_attr = false;
        if (!stamp_attr) {
            if (hipFuncSetAttribute((const void*)conv1_wgrad_kernel<XCH, YCH, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return -2;
            stamp_attr = true;
        }
        conv1_wgrad_kernel<XCH, YCH, false, true><<<grid, 512, (size_t)lds_of(R), s>>>(p, (unsigned long long*)strtoull(se, nullptr, 0));
    } else
    if (p.u8) conv1_wgrad_kernel<XCH, YCH, true><<<grid, 512, (size_t)lds_of(R), s>>>(p);
    else conv1_wgrad_kernel<XCH, YCH, false><<<grid, 512, (size_t)lds_of(R), s>>>(p);
    const long Rw = 32L * K;
    const int nbw = (int)((Rw + 63) / 64);
    wband_reduce_kernel<<<nbw + (db ? 1 : 0), 1024, 0, s>>>(p.partial_w, dw, grid, Rw, accumulate, 0, 64, nbw, p.partial_b, db, 32);
    return 0;
}

template <int C, int CT, int TH, int TW, int S, bool NCHW, int XCH, int YCH, int BPC, bool PURE16 = false>
int launch_wband(WBandP& p, float* dw, float* db, void* ws, long ws_bytes, int dw_oihw, int accumulate, hipStream_t s) {
    constexpr int COUT = CT * 32, K = TH * TW * C;
    constexpr int PS = NCHW ? 2 : C * 2 + 16;
    const int Wb = p.W, Wp = Wb;
    if (NCHW && p.W % 4) return -1;
    static_assert(!NCHW || XCH % C == 0, "NCHW staging: XCH items split evenly over the channel planes");
    auto lds_of = [&](int R, int F) -> long {
        const int rows = (F - 1) * p.H + (R - 1) * S + TH;
        const long xb = NCHW ? (long)C * (((long)rows * Wp + 7) / 8 * 8) * 2 : (long)rows * Wb * PS;
        const long npad = ((long)F * R * p.OW + 15) / 16 * 16;
        const long dyb = (!NCHW && PURE16) ? npad * (COUT * 2 + 16) : (long)COUT * (npad * 2 + 16);
        return (xb + 15) / 16 * 16 + dyb + npad * 4 + 64;
    };
    auto fits = [&](int R, int F) -> bool {
        const int rows = (F - 1) * p.H + (R - 1) * S + TH;
        const long xitems = NCHW ? ((long)rows * p.W + 7) / 8 * C : (long)rows * Wb * (C / 8);
        const long yitems = (((long)F * R * p.OW + 15) / 16 * 16) * (COUT / 8);
        return lds_of(R, F) <= (160 * 1024 - 256) / BPC && xitems <= (long)XCH * 512 && yitems <= (long)YCH * 512;
    };
    int R = p.OH, F = 1;
    while (R > 1 && !fits(R, 1)) --R;
    if (!fits(R, 1)) return -1;
    int nunits;
    if (R == p.OH && !NCHW) {                                // whole frames fit: pack several (contiguous) frames into one unit
        while (F < p.Nimg && fits(R, F + 1)) ++F;
        nunits = (p.Nimg + F - 1) / F;
    } else {
        const int bands = (p.OH + R - 1) / R;
        R = (p.OH + bands - 1) / bands;
        nunits = p.Nimg * bands;
    }
    if ((long)F * R * p.OW < 96) return -1;                  // too few pixels per unit: the gather kernel wins
    p.R = R; p.F = F;
    const int slots = 256 * BPC;                             // resident workgroups
    const int per = (nunits + slots - 1) / slots;            // balanced persistent grid: every workgroup gets `per` (or per - 1) units
    const int grid = (nunits + per - 1) / per;
    if ((long)grid * COUT * (K + 1) * 4 > ws_bytes) return -1;
    p.partial_w = (float*)ws;
    p.partial_b = db ? p.partial_w + (long)grid * COUT * K : nullptr;
    auto kern = conv_wgrad_band_kernel<C, CT, TH, TW, S, NCHW, XCH, YCH, BPC, PURE16>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
        attr_set = true;
    }
    const char* se = getenv("HULC_WB_STAMPS");               // <device address of grid x 8 x 7 uint64>: the instrumented instance
    if (se && *se && PURE16 && !NCHW) {
        auto kst = conv_wgrad_band_kernel<C, CT, TH, TW, S, NCHW, XCH, YCH, BPC, PURE16, true>;
        static bool st_attr = false;
        if (!st_attr) {
            if (hipFuncSetAttribute((const void*)kst, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
            st_attr = true;
        }
        kst<<<grid, 512, (size_t)lds_of(R, F), s>>>(p, (unsigned long long*)strtoull(se, nullptr, 0));
    } else
    kern<<<grid, 512, (size_t)lds_of(R, F), s>>>(p, nullptr);
    const long Rw = (long)COUT * K;
    const int nbw = (int)((Rw + 63) / 64);
    wband_reduce_kernel<<<nbw + (db ? 1 : 0), 1024, 0, s>>>(p.partial_w, dw, grid, Rw, accumulate, (dw_oihw && !NCHW) ? C : 0, TH * TW, nbw, p.partial_b, db,
                                                          COUT);
    return 0;
}

}  // namespace

// 0 = launched, 1 = geometry not covered (caller uses the gather kernel), < 0 = error.  dw is [Cout][K] fp32 in the forward k order.
int hulc_conv_wgrad_band_dispatch(int nchw, int Cin, int Cout, int KH, int KW, int S, const void* x, int x_dtype, const void* dy, int dy_dtype,
                                  int N, int H, int W, float* dw, float* db, void* ws, long ws_bytes, int dw_oihw, int accumulate, int u8, int pad,
                                  const int* shift, const int* fidx, const void* x2, int n_split, const void* x_slot, const void* x2_slot, hipStream_t s) {
    if (getenv("HULC_NO_BAND_WGRAD") && !u8) return 1;
    WBandP p;
    p.u8 = u8; p.pad = pad; p.shift = shift; p.fidx = fidx;
    // (read per launch: the tests switch between the arrangements inside one process)
    { const char* e = getenv("HULC_WB_BURST"); p.burst = e && atoi(e); }
    { const char* e = getenv("HULC_WB_STAG"); p.stag_num = e ? atoi(e) : 8; }
    p.X = x; p.dY = dy; p.x_dtype = x_dtype; p.dy_dtype = dy_dtype;
    p.Nimg = N; p.H = H; p.W = W; p.OH = (H - KH) / S + 1; p.OW = (W - KW) / S + 1; p.R = 1; p.F = 1;
    if (nchw) { p.x_sn = (long)Cin * H * W; p.x_sc = (long)H * W; p.x_sy = W; p.x_sx = 1; }
    else { p.x_sn = (long)H * W * Cin; p.x_sy = (long)W * Cin; p.x_sx = Cin; p.x_sc = 1; }
    p.dy_sn = (long)p.OH * p.OW * Cout; p.dy_sy = (long)p.OW * Cout; p.dy_sx = Cout;
    int rc = 1;
    const bool pure16 = x_dtype == HULC_BF16 && dy_dtype == HULC_BF16;
    if (!nchw && Cin == 64 && Cout == 64 && KH == 3 && KW == 3 && S == 1)
        rc = pure16 ? launch_wband<64, 2, 3, 3, 1, false, 5, 4, 1, true>(p, dw, db, ws, ws_bytes, dw_oihw, accumulate, s)
                    : launch_wband<64, 2, 3, 3, 1, false, 5, 4, 1>(p, dw, db, ws, ws_bytes, dw_oihw, accumulate, s);
    else if (!nchw && Cin == 32 && Cout == 64 && KH == 4 && KW == 4 && S == 2)
        rc = pure16 ? launch_wband<32, 2, 4, 4, 2, false, 10, 5, 1, true>(p, dw, db, ws, ws_bytes, dw_oihw, accumulate, s)
                    : launch_wband<32, 2, 4, 4, 2, false, 10, 5, 1>(p, dw, db, ws, ws_bytes, dw_oihw, accumulate, s);
    else if (nchw && Cin == 3 && Cout == 32 && KH == 8 && KW == 8 && S == 4 && (x_dtype == HULC_F32 || (u8 && W % 4 == 0 && (uintptr_t)x % 4 == 0))) {
        if (x2 && (dy_dtype != HULC_BF16 || n_split < 0 || n_split > N || ((uintptr_t)x2 % (u8 ? 4 : 16)) || (u8 && fidx) || getenv("HULC_CONV1_WGRAD_OLD")))
            return hulc_fail(-6, "conv1 weight gradient: x2 needs a bf16 gradient map, 0 <= n_split <= N, 16-byte (uint8 frames: 4-byte) alignment, no frame_index");
        if ((x_slot || x2_slot) && (getenv("HULC_CONV1_WGRAD_OLD") || dy_dtype != HULC_BF16))
            return hulc_fail(-6, "conv1 weight gradient: frame slots need the phase-plane kernel (bf16 gradient map)");
        if (getenv("HULC_CONV1_WGRAD_OLD") || dy_dtype != HULC_BF16) rc = launch_wband<3, 1, 8, 8, 4, true, 6, 4, 2>(p, dw, db, ws, ws_bytes, dw_oihw, accumulate, s);
        else {
            W1P q;
            q.X = x; q.dY = dy; q.dy_dtype = dy_dtype; q.Nimg = N; q.H = H; q.W = W; q.OH = p.OH; q.OW = p.OW;
            q.dy_sn = p.dy_sn; q.dy_sy = p.dy_sy; q.dy_sx = p.dy_sx;
            q.u8 = u8; q.pad = pad; q.shift = shift; q.fidx = fidx;
            q.X2 = x; q.nsplit = N;
            q.xs = (const void* const*)x_slot; q.xs2 = (const void* const*)x2_slot;
            if ((x_slot || x2_slot) && (u8 || !x_slot || (x2_slot && !x2) || ((uintptr_t)x_slot | (uintptr_t)x2_slot) % 8))
                return hulc_fail(-6, "conv1 weight gradient: frame slots are for fp32 frames (x_slot with every launch, x2_slot next to x2), 8-byte aligned");
            if (x2) {
                q.X2 = u8 ? (const float*)((const unsigned char*)x2 - (long)n_split * 3 * H * W) : (const float*)x2 - (long)n_split * 3 * H * W;
                q.nsplit = n_split;
            }
            q.dbg = getenv("HULC_W1_DBG") ? atoi(getenv("HULC_W1_DBG")) : 0;
            rc = launch_conv1_wgrad(q, dw, db, ws, ws_bytes, accumulate, s);
        }
    }
    else return (x2 || x_slot) ? hulc_fail(-6, "conv weight gradient: x2 / frame slots are for conv1 only") : 1;
    if (rc == -1) return 1;
    if (rc < 0) return hulc_fail(-8, "conv wgrad band: could not raise the dynamic LDS limit");
    return 0;
}
