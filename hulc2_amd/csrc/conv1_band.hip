// conv1_band.hip — forward of the first camera-encoder conv (NCHW fp32 frames, 3 -> 32 channels, 8x8 stride 4) from LDS bands.
//
// reference arithmetic: nn.Conv2d(3, 32, 8, stride=4) + ReLU, hulc2/models/perceptual_encoders/vision_network.py:37-39 and
// vision_network_gripper.py:13.
//
// The layer is HBM-bound (30 GFLOP against 491 MB of fp32 frames + 157 MB of bf16 output per 1024 static frames); the
// gather kernel re-reads every input element 4x from L2 in 32-byte pieces and converts it 4x.  Here a work unit (frame x band
// of R output rows, full width) is copied ONCE into LDS as bf16 channel planes (flat 32-byte-per-lane copy, converted once,
// next band prefetched into registers during the MFMAs), the 32 x 192 weight matrix lives in registers as A-operand
// fragments, and a wave computes D[channel][pixel] for 32 consecutive output pixels per tile: the B fragment of k-step
// (c, kh pair) is the 8 contiguous kw of the patch row — two aligned ds_read_b64, conflict-free across the 32 pixels.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "u8_frames.h"
#include <stdlib.h>

namespace {

struct C1P {
    const void* X; const void* Wt; const float* bias; void* Y;
    const void* Wlo;                             // X3: bf16 remainders of the weights (same layout)
    int w_dtype, y_dtype, relu;
    int Nimg, H, W, OH, OW, R;
    unsigned* bits;                              // optional ReLU sign plane: one dword per output pixel (bit c = channel c > 0)
    int sweep;                                   // unit order: 0 = a workgroup walks whole frames, 1 = the grid sweeps memory in address order
    long ldw;
    int dbg;
    int u8, pad; const int* shift; const int* fidx;   // uint8 NHWC source with the shift / scale / normalise transforms applied while staging
    const void* X2; int nsplit;                  // fp32 frames: frames n >= nsplit come from X2 (pre-offset by -nsplit frames); X2 == X when unused
    const void* const* xs; const void* const* xs2;   // optional device slots holding the frame tensors' addresses (fp32 frames; xs2 un-offset): read at kernel start
};

// U8: uint8 NHWC frames (else fp32 NCHW planes) — compile-time, so that the two load paths never join in front of the MFMA loop
// X3 (fp32 frames only): the band is staged as hi + lo bf16 planes, the weights as hi + remainder, every product from the splits of both
// operands (a_hi b_hi + a_lo b_hi + a_hi b_lo): fp32-class outputs; the layer is HBM-bound, the two extra MFMAs per k-step are not what it waits for
template <int XCH, bool U8, bool X3 = false>
__global__ __launch_bounds__(512, X3 ? 2 : 4) void conv1_band_kernel(C1P p) {     // (X3: twice the LDS per workgroup, two per CU: 256 registers)
    constexpr int NT = 512, C = 3, TH = 8, TW = 8, S = 4, K = C * TH * TW, KSTEPS = K / 16;   // 12 k-steps of (c, kh pair)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int bands = (p.OH + p.R - 1) / p.R;
    const float inv_OW = fast_rcp(p.OW);
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
    // sweep order (HULC_CONV1_SWEEP=1, experiment): unit u of the launch = (frame u / bands, band u % bands), workgroup w takes w, w + grid, ...
    // — at any moment the grid reads one compact window of a few dozen frames instead of one stream per workgroup 480 KB apart
    const int total_units = p.Nimg * bands;
    const int nunits = p.sweep ? (total_units - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x
                               : ((p.Nimg - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x) * bands;     // this workgroup's units
    const int rows_max = (p.R - 1) * S + TH;
    const int PP = (rows_max * p.W + 7) / 8 * 8;                 // plane pitch (elements)

    // ---- weights: 32 x 192 bf16 in LDS, row stride 400 B (conflict-free ds_read_b128 of the A-operand fragments: lane =
    //      output channel r, 8 consecutive k of k-step ks at half h).  Registers stay under 128 -> two workgroups per CU.
    constexpr int WROW = K * 2 + 16;
    char* wlds = smem;
    float* blds = (float*)(smem + 32 * WROW);                    // bias in LDS: a global load inside the tile loop would wait, in order, behind the prefetch
    char* wlo = smem + 32 * WROW + 128;                          // (X3) the weights' remainders
    char* xlds = smem + (X3 ? 2 : 1) * 32 * WROW + 128;
    char* xlo = xlds + (long)C * PP * 2;                         // (X3) lo planes behind the hi planes
    if (tid < 32) blds[tid] = p.bias ? p.bias[tid] : 0.f;
    for (int id = tid; id < 32 * (K / 8); id += NT) {
        const int row = id / (K / 8), ch = id % (K / 8);
        *(uint4*)(wlds + row * WROW + ch * 16) = *(const uint4*)((const uint16_t*)p.Wt + (long)row * p.ldw + ch * 8);   // bf16 weights (dispatch checks)
        if (X3) *(uint4*)(wlo + row * WROW + ch * 16) = *(const uint4*)((const uint16_t*)p.Wlo + (long)row * p.ldw + ch * 8);
    }

    // the prefetched band stays RAW in registers (8 floats per item / the aligned dword windows of uint8 frames) and is converted in
    // stage_store: any ALU use of a loaded value here would put the wait for the loads in front of the MFMA loop they overlap
    constexpr int XF = U8 ? C : XCH;                          // register slots of the fp32 path (dead code in the uint8 instance)
    f32x4_t xraw[U8 ? 1 : XCH][2];
    uint32_t uraw[U8 ? XCH * 8 : 1];                          // (uint8 frames: plain dwords — packing a dwordx3 + dword into float4 tuples made the compiler
                                                             //  copy registers right behind the loads, i.e. wait for every load where it was issued)
    auto unit_geom = [&](int unit, int& n, int& r0, int& R, int& rows) {
        int b = unit % bands;
        n = blockIdx.x + (unit / bands) * gridDim.x;
        if (p.sweep) { const int g = blockIdx.x + unit * gridDim.x; n = g / bands; b = g - n * bands; }
        r0 = b * p.R; R = (r0 + p.R <= p.OH) ? p.R : p.OH - r0; rows = (R - 1) * S + TH;
    };
    // (uint8 frames) per-frame parameters — augmentation shift, frame index — of this workgroup's first MAXU units, read once into LDS: as global
    // loads in front of every unit's prefetch they were two dependent round trips to memory per unit (the pointers sit in a by-value struct,
    // so the compiler cannot prove them read-only and will not use scalar loads)
    constexpr int MAXU = 256;
    int4* ftab = (int4*)(xlds + (long)C * PP * 2);
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
    // a unit's geometry is carried from unit to unit (next band of the frame, or the first band of the workgroup's next frame): as
    // unit / bands and unit % bands it was two scalar division sequences in front of every unit's loads (sweep order keeps the divisions)
    struct Geo { int n, r0, R, rows; };
    auto geo_next = [&](int unit_next, const Geo& g) -> Geo {
        Geo o;
        if (p.sweep) { unit_geom(unit_next, o.n, o.r0, o.R, o.rows); return o; }
        const int b1 = g.r0 + p.R;
        const bool wrapf = b1 >= p.OH;
        o.n = wrapf ? g.n + (int)gridDim.x : g.n;
        o.r0 = wrapf ? 0 : b1;
        o.R = (o.r0 + p.R <= p.OH) ? p.R : p.OH - o.r0;
        o.rows = (o.R - 1) * S + TH;
        return o;
    };
    // thread-only part of the fp32 load addresses: the byte offset of the thread's item inside the band of plane c
    unsigned xld[C];
#pragma unroll
    for (int c = 0; c < C; ++c) xld[c] = (unsigned)(c * p.H * p.W + tid * 8) * 4u;
    auto stage_load = [&](int unit, const Geo& g) {
        const int n = g.n, r0 = g.r0, R = g.R, rows = g.rows;
        const int nflt = rows * p.W, items = (nflt + 7) / 8;
        if (U8) {                                          // uint8 NHWC frames: one item = 8 elements of all three planes
            int sx, sy, fi; frame_params(unit, n, sx, sy, fi);
            const unsigned char* img = (const unsigned char*)(n < p.nsplit ? p.X : p.X2) + (long)fi * p.H * p.W * 3;   // (n is uniform: a scalar select)
            // (a uint8 item — 8 pixels of all three channels — is 8 raw dwords, a third of what the same pixels cost as fp32: all XCH register
            //  slots hold items, so a band is up to three times as tall and the per-unit costs (barriers, index arithmetic, a partly filled
            //  last tile) are paid a third as often)
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int id = tid + i * NT;
                const bool inb = id < items;
                u8_band_chunk3_load(img, p.H, p.W, r0 * S, inb ? id * 8 : 0, inb ? nflt : 0, sx - p.pad, sy - p.pad, &uraw[(U8 ? i : 0) * 8]);
            }
            return;
        }
        if (XF == C) {
            // the band rows of a channel plane are contiguous -> flat copy, 8 floats per item (rows * W is a multiple of 16: no partial item).
            // Address = uniform base of the band (scalar registers) + a 32-bit byte offset that depends on the thread only: one load
            // instruction each, no 64-bit vector address arithmetic
            const char* ub = (const char*)((n < p.nsplit ? xbase : xbase2) + ((long)n * C * p.H * p.W + (long)(r0 * S) * p.W));
            typedef const f32x4_t __attribute__((address_space(1))) * gvec;
            // (measured: with the loads of the four halo rows a band shares with its predecessor — L2 hits, a fifth of the requests — switched
            //  off, the kernel is 2 % faster: a row ring in LDS that keeps them is not worth building)
            const bool inb = tid < items;
#pragma unroll
            for (int j = 0; j < XF; ++j) {
                const unsigned o = inb ? xld[j] : (unsigned)(j * p.H * p.W) * 4u;
                xraw[U8 ? 0 : j][0] = *(gvec)(ub + o);
                xraw[U8 ? 0 : j][1] = *(gvec)(ub + o + 16);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < XF; ++j) {
            const int c = j / (XF / C), id = tid + (j % (XF / C)) * NT;
            const bool inb = id < items, inb2 = inb && id * 8 + 8 <= nflt;
            const long off = ((long)n * C + c) * p.H * p.W + (long)(r0 * S) * p.W + (inb ? (long)id * 8 : 0);
            const float* xb = n < p.nsplit ? xbase : xbase2;                   // (n is uniform: a scalar select)
            // (global address space spelled out: a base address read from a device slot is an integer, and a pointer made of one is a FLAT
            //  pointer — flat loads count against lgkmcnt as well, so every LDS wait of the tile loop waited for the frame prefetch too)
            typedef const f32x4_t __attribute__((address_space(1))) * gvec;
            xraw[U8 ? 0 : j][0] = *(gvec)(xb + off);
            xraw[U8 ? 0 : j][1] = *(gvec)(xb + (inb2 ? off + 4 : off));
        }
    };
    auto stage_store = [&](int unit, const Geo& g) {
        const int n = g.n, r0 = g.r0, R = g.R, rows = g.rows;
        const int nflt = rows * p.W, items = (nflt + 7) / 8;
        uint4 xpre[XF];
        if (U8) {
            int sx, sy, fi; frame_params(unit, n, sx, sy, fi);
#pragma unroll
            for (int i = 0; i < XCH; ++i) {
                const int id = tid + i * NT;
                const bool inb = id < items;
                const uint32_t* raw = &uraw[(U8 ? i : 0) * 8];
                uint4 q0, q1, q2;
                u8_band_chunk3_convert(p.W, inb ? id * 8 : 0, inb ? nflt : 0, sx - p.pad, raw, q0, q1, q2);
                if (inb) {
                    *(uint4*)(xlds + ((long)0 * PP + id * 8) * 2) = q0;
                    *(uint4*)(xlds + ((long)1 * PP + id * 8) * 2) = q1;
                    *(uint4*)(xlds + ((long)2 * PP + id * 8) * 2) = q2;
                }
            }
            return;
        } else {
#pragma unroll
            for (int j = 0; j < XF; ++j) {
                const int id = tid + (j % (XF / C)) * NT;
                const bool inb2 = id * 8 + 8 <= nflt;
                const f32x4_t a = xraw[U8 ? 0 : j][0], b = xraw[U8 ? 0 : j][1];
                xpre[j] = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), inb2 ? pack_bf16x2(b.x, b.y) : 0u, inb2 ? pack_bf16x2(b.z, b.w) : 0u);
            }
        }
#pragma unroll
        for (int j = 0; j < XF; ++j) {
            const int c = j / (XF / C), id = tid + (j % (XF / C)) * NT;
            if (id < items) *(uint4*)(xlds + ((long)c * PP + id * 8) * 2) = xpre[j];
            if (X3 && id < items) {                             // remainders a - bf16(a) of the same 8 values
                const f32x4_t a = xraw[U8 ? 0 : j][0], b = xraw[U8 ? 0 : j][1];
                const uint4 hi = xpre[j];
                const uint4 lo = make_uint4(pack_bf16x2(a.x - __uint_as_float(hi.x << 16), a.y - __uint_as_float(hi.x & 0xffff0000u)),
                                            pack_bf16x2(a.z - __uint_as_float(hi.y << 16), a.w - __uint_as_float(hi.y & 0xffff0000u)),
                                            hi.z | hi.w ? pack_bf16x2(b.x - __uint_as_float(hi.z << 16), b.y - __uint_as_float(hi.z & 0xffff0000u)) : 0u,
                                            hi.z | hi.w ? pack_bf16x2(b.z - __uint_as_float(hi.w << 16), b.w - __uint_as_float(hi.w & 0xffff0000u)) : 0u);
                *(uint4*)(xlo + ((long)c * PP + id * 8) * 2) = lo;
            }
        }
    };

    int unit = 0;
    Geo gc;
    unit_geom(0, gc.n, gc.r0, gc.R, gc.rows);
    if (unit < nunits) { stage_load(unit, gc); stage_store(unit, gc); }
    __syncthreads();
    for (; unit < nunits; ++unit) {
        const int next = unit + 1;
        const Geo gn = geo_next(next, gc);
        // (uint8 frames: the prefetch is unconditional — after the last unit it re-reads that unit's band, from L2, into registers nobody uses.  With
        //  the previous unit's values flowing around a skipped prefetch the register allocator copied loaded dwords into the loop-carried registers
        //  right behind each load: a wait for the load where it was issued)
        if (U8) { if (!(p.dbg & 2)) { if (next < nunits) stage_load(next, gn); else stage_load(unit, gc); } }
        else if (next < nunits && !(p.dbg & 2)) stage_load(next, gn);
        // this lane's A-operand fragments (output channel r, half h of every k-step) live in registers over the unit's tiles only: re-read per
        // unit (12 LDS reads against 36+ fragment reads), they do not sit on the register budget while the next band is converted
        // this lane's A-operand fragments (output channel r, half h of every k-step) live in registers over the unit's tiles only: re-read per
        // unit (12 LDS reads against 36+ fragment reads), they do not sit on the register budget while the next band is converted
        bf16x8_t wreg[X3 ? 1 : KSTEPS];
        if (!X3) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) wreg[ks] = *(const bf16x8_t*)(wlds + r * WROW + h * 16 + ks * 32);
        }
        const int n = gc.n, r0 = gc.r0, R = gc.R;
        const int npix = R * p.OW, ntiles = (npix + 31) / 32;
        for (int t = wave; t < ((p.dbg & 1) ? 0 : ntiles); t += 8) {
            const int q = t * 32 + r;
            const int qc = q < npix ? q : npix - 1;
            const int oy = fast_div(qc, inv_OW), ox = qc - oy * p.OW;
            // byte offset of the lane's patch in LDS, band base included (8-byte aligned: ox*S*2 = 8*ox; lane half h = odd patch row): a k-step adds
            // ONE scalar to it (as pointer + (c, kh) offset + band base the compiler spent two vector adds per k-step)
            const int patch = (int)(xlds - smem) + ((oy * S) * p.W + ox * S + h * p.W) * 2;
            const char* wrow = wlds + r * WROW + h * 16;
            f32x16_t acc;                                          // starts from the bias (read next to the first band fragments: one wait, no adds in the epilogue)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const float4 b4 = *(const float4*)(blds + 8 * g4 + 4 * h);
                acc[g4 * 4 + 0] = b4.x; acc[g4 * 4 + 1] = b4.y; acc[g4 * 4 + 2] = b4.z; acc[g4 * 4 + 3] = b4.w;
            }
            // the band fragments of PF k-steps ahead are requested before each MFMA (left alone the compiler reads one fragment, waits for
            // it, multiplies: the LDS round trip twelve times per tile with four waves per SIMD to hide it)
            // (round 6, measured: weight fragments read per k-step, two accumulator chains and fragments two k-steps ahead — 265 -> 262 us per
            //  2048 fp32 frames, 206 -> 215 with uint8 frames: the tile phase is not what this kernel waits for; removed)
            constexpr int PF = X3 ? 1 : 1;
            auto frag = [&](int ks, int base) {
                const int c = ks / 4, kh2 = (ks % 4) * 2;                         // patch row kh2 + h of channel c
                const char* src = smem + (base + (c * PP + kh2 * p.W) * 2);
                union { uint2 u[2]; bf16x8_t b; } x;
                x.u[0] = *(const uint2*)src; x.u[1] = *(const uint2*)(src + 8);
                return x.b;
            };
            bf16x8_t xb[PF + 1];
#pragma unroll
            for (int ks = 0; ks < PF; ++ks) xb[ks] = frag(ks, patch);
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) {
                if (ks + PF < KSTEPS) xb[(ks + PF) % (PF + 1)] = frag(ks + PF, patch);
                if (!X3) __builtin_amdgcn_sched_barrier(0);
                const bf16x8_t wf = X3 ? *(const bf16x8_t*)(wrow + ks * 32) : wreg[X3 ? 0 : ks];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xb[ks % (PF + 1)], acc, 0, 0, 0);
                if (X3) {
                    const bf16x8_t xl = frag(ks, patch + (int)(xlo - xlds));
                    const bf16x8_t wl = *(const bf16x8_t*)(wrow + (wlo - wlds) + ks * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf, xl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, xb[ks % (PF + 1)], acc, 0, 0, 0);
                }
            }
            {
                const long yo = (((long)n * p.OH + r0) * p.OW + qc) * 32;       // band pixels are contiguous in the NHWC output (qc: clamped, always valid)
                if (p.y_dtype == HULC_BF16) {
                    // the two lane halves of a pixel hold interleaved groups of 4 channels: v_permlane32_swap gives every lane 8 consecutive
                    // channels — two 16-byte stores per pixel instead of four 8-byte ones
                    // ReLU on the PACKED words: bf16 bit patterns compare like sign-magnitude integers, so max(x, 0) per signed 16-bit half clears
                    // exactly the negative halves (-0 included) — 8 v_pk_max_i16 where fmaxf on the 16 floats was 32 VALU (canonicalise + max).
                    // The tile loop is bound by instruction issue (MFMA 12 x 8 passes against ~200 VALU per tile before this), not by memory.
                    const uint32_t floor2 = p.relu ? 0u : 0x80008000u;
                    uint2 pk[4];
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        pk[g4] = make_uint2(pack_bf16x2(acc[g4 * 4 + 0], acc[g4 * 4 + 1]), pack_bf16x2(acc[g4 * 4 + 2], acc[g4 * 4 + 3]));
                        pk[g4].x = max_s16x2(pk[g4].x, floor2); pk[g4].y = max_s16x2(pk[g4].y, floor2);
                    }
                    if (p.bits) {
                        // sign plane of the stored (rectified) bf16 values: pk[g4] = channels 8 g4 + 4 h + {0..3}; positive <=> half != 0.
                        // min(half, 1) per unsigned half gives the bits at positions 0 / 16; two words make a nibble with three more operations
                        unsigned mb = 0;
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) {
                            const unsigned u = nonzero_u16x2(pk[g4].x) | (nonzero_u16x2(pk[g4].y) << 2);     // bits 0, 16, 2, 18 = channels 0, 1, 2, 3
                            mb |= ((u | (u >> 15)) & 0xfu) << (8 * g4);
                        }
                        mb <<= 4 * h;
                        const auto other = __builtin_amdgcn_permlane32_swap(mb, mb, false, false);   // [1] on the lower lanes = the upper half's word
                        if (q < npix && h == 0) p.bits[yo >> 5] = mb | other[1];
                    }
#pragma unroll
                    for (int gp = 0; gp < 2; ++gp) {
                        const auto sx = __builtin_amdgcn_permlane32_swap(pk[2 * gp].x, pk[2 * gp + 1].x, false, false);
                        const auto sy = __builtin_amdgcn_permlane32_swap(pk[2 * gp].y, pk[2 * gp + 1].y, false, false);
                        if (q < npix) *(uint4*)((uint16_t*)p.Y + yo + 16 * gp + 8 * h) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
                    }
                } else if (q < npix) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) {
                        float v0 = acc[g4 * 4 + 0], v1 = acc[g4 * 4 + 1], v2 = acc[g4 * 4 + 2], v3 = acc[g4 * 4 + 3];
                        if (p.relu) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                        *(float4*)((float*)p.Y + yo + 8 * g4 + 4 * h) = make_float4(v0, v1, v2, v3);
                    }
                }
            }
        }
        __syncthreads();
        if (next < nunits && !(p.dbg & 2)) stage_store(next, gn);
        __syncthreads();
        gc = gn;
    }
}

template <int XCH, int UX>
int launch_conv1(C1P& p, hipStream_t s) {
    const int x3 = p.Wlo != nullptr;
    auto lds_of = [&](int R) -> long { const long rows = (R - 1) * 4 + 8; return (1 + x3) * 32 * (192 * 2 + 16) + 128 + (1 + x3) * 3 * ((rows * p.W + 7) / 8 * 8) * 2 + 64 + (p.u8 ? 256 * 16 : 0); };
    auto fits = [&](int R) -> bool {
        const long rows = (R - 1) * 4 + 8;
        static const bool tall = !(getenv("HULC_CONV1_U8_TALL") && atoi(getenv("HULC_CONV1_U8_TALL")) == 0);
        return lds_of(R) <= (160 * 1024 - 256) / 2 && (rows * p.W + 7) / 8 * ((p.u8 && tall) ? 1 : 3) <= (long)(p.u8 ? UX : XCH) * 512;
    };
    int R = p.OH;
    while (R > 1 && !fits(R)) --R;
    if (!fits(R)) return -1;
    const int bands = (p.OH + R - 1) / R;
    R = (p.OH + bands - 1) / bands;
    p.R = R;
    // two workgroups per CU (LDS <= 80 KB each).  HULC_CONV1_SLOTS (tests): fewer, so that a workgroup walks more than the 256 units its LDS
    // table of per-frame parameters holds and the direct loads behind the table are exercised
    const int slots = getenv("HULC_CONV1_SLOTS") && atoi(getenv("HULC_CONV1_SLOTS")) > 0 ? atoi(getenv("HULC_CONV1_SLOTS")) : 512;
    const int per = (p.Nimg + slots - 1) / slots;                // frames per workgroup
    int grid = (p.Nimg + per - 1) / per;
    { static const char* e = getenv("HULC_CONV1_SWEEP"); p.sweep = e ? atoi(e) : 0; }
    if (p.sweep) { const long tu = (long)p.Nimg * bands; grid = (int)(tu < slots ? tu : slots); }
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)conv1_band_kernel<XCH, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv1_band_kernel<XCH, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess ||
            hipFuncSetAttribute((const void*)conv1_band_kernel<UX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return -2;
        attr_set = true;
    }
    if (x3) conv1_band_kernel<XCH, false, true><<<grid, 512, (size_t)lds_of(R), s>>>(p);
    else if (p.u8) conv1_band_kernel<UX, true><<<grid, 512, (size_t)lds_of(R), s>>>(p);
    else conv1_band_kernel<XCH, false><<<grid, 512, (size_t)lds_of(R), s>>>(p);
    return 0;
}

}  // namespace

// 0 = launched, 1 = geometry not covered (caller uses the gather kernel), < 0 = error
int hulc_conv1_band_dispatch(const float* x, const void* w, int w_dtype, long ldw, const float* bias, void* y, int y_dtype, int relu,
                             int N, int H, int W, int u8, int pad, const int* shift, const int* fidx, unsigned* relu_bits, const void* w_lo,
                             const void* x2, int n_split, const void* x_slot, const void* x2_slot, hipStream_t s) {
    if (getenv("HULC_NO_BAND_CONV1") && !u8) return 1;
    if (w_dtype != HULC_BF16 || ((uintptr_t)w % 16) || ldw % 8) return u8 ? hulc_fail(-6, "conv1 band: bf16 weights, 16-byte aligned rows") : 1;
    if (W % 4 || ((uintptr_t)x % (u8 ? 4 : 16)) || (bias && ((uintptr_t)bias % 16)) || (H - 8) % 4 || (W - 8) % 4) return 1;
    C1P p;
    if (relu_bits && (y_dtype != HULC_BF16 || !relu)) return 1;       // (planes describe the stored bf16 ReLU output)
    if (w_lo && (u8 || (uintptr_t)w_lo % 16)) return hulc_fail(-6, "conv1 band: split operands need fp32 frames and 16-byte aligned remainders");
    p.u8 = u8; p.pad = pad; p.shift = shift; p.fidx = fidx; p.bits = relu_bits; p.Wlo = w_lo;
    { static const char* e = getenv("HULC_C1_DBG"); p.dbg = e ? atoi(e) : 0; }
    p.X = x; p.Wt = w; p.bias = bias; p.Y = y; p.w_dtype = w_dtype; p.y_dtype = y_dtype; p.relu = relu;
    p.X2 = x; p.nsplit = N;
    p.xs = (const void* const*)x_slot; p.xs2 = (const void* const*)x2_slot;
    if ((x_slot || x2_slot) && (u8 || !x_slot || (x2_slot && !x2) || ((uintptr_t)x_slot | (uintptr_t)x2_slot) % 8))
        return hulc_fail(-6, "conv1 band: frame slots are for fp32 frames (x_slot with every launch, x2_slot next to x2), 8-byte aligned");
    if (x2) {
        if (n_split < 0 || n_split > N || ((uintptr_t)x2 % (u8 ? 4 : 16)) || (u8 && fidx))
            return hulc_fail(-6, "conv1 band: x2 needs 0 <= n_split <= N, 16-byte (uint8 frames: 4-byte) alignment and no frame_index");
        p.X2 = u8 ? (const float*)((const unsigned char*)x2 - (long)n_split * 3 * H * W) : (const float*)x2 - (long)n_split * 3 * H * W;
        p.nsplit = n_split;
    }
    p.Nimg = N; p.H = H; p.W = W; p.OH = (H - 8) / 4 + 1; p.OW = (W - 8) / 4 + 1; p.R = 1; p.ldw = ldw;
    const int rc = launch_conv1<3, 2>(p, s);
    if (rc == -1) return 1;
    if (rc < 0) return hulc_fail(-8, "conv1 band: could not raise the dynamic LDS limit");
    return 0;
}
