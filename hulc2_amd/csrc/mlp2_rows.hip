// mlp2_rows.hip — Linear(128, H) + ReLU + Linear(H, OUT) over many rows as ONE launch per direction (bf16 MFMA, fp32 accumulate).
//
// reference arithmetic: the head of both camera encoders, fc2(relu(fc1(x))) with fc1 = Linear(128, 512), fc2 = Linear(512, 64) on the
// 2048 frames of a step — hulc2/models/perceptual_encoders/vision_network.py:49-52,60-66 and vision_network_gripper.py:36-39,52-55 — and
// its autograd backward.
//
// As two GEMMs per direction the (rows x 512) hidden activation is written and re-read in fp32 and each of the four launches costs its
// floor (14 us for 0.13-0.27 GFLOP).  Here a workgroup owns 32 rows and the hidden activation never leaves its registers — the
// feed-forward stage of txl_block.hip without the LayerNorm: wave w owns 32 of every 128 hidden units, computes the TRANSPOSED hidden tile
// z^T[hidden][row] = W1s x^T, applies bias + ReLU in registers and feeds the packed accumulator back as the B operand of
// y^T[out][row] += W2[out][hidden] h^T (W2's columns read in the accumulator's register order, txl_attn.h); the four waves' partial
// output tiles meet once in LDS.  Backward recomputes z^T (same instructions: the same ReLU gate), forms dh^T = (W2^T dy^T) * gate and
// dx^T += W1^T dh^T in registers, and stores h and dh once (bf16, row-major) as the operands of dW1 = dh^T x and dW2 = dy^T h, which join
// the pass's grouped weight-gradient launch (row sums = bias gradients).
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "txl_attn.h"

namespace {

struct Mlp2P {
    const float* x;                 // (T, 128)
    const uint16_t *W1, *W2;        // bf16 [H][128], [OUT][H]
    const uint16_t *W1lo, *W2lo;    // X3 forward: bf16 of the rounding remainders w - bf16(w), same layouts
    const uint16_t *W1T, *W2T;      // backward: [128][H], [H][OUT]
    const float *b1, *b2;
    int T, H, OUT;
    float* y;                       // (T, OUT)
    const float* dy;                // (T, OUT)
    float* dx;                      // (T, 128) or null
    uint16_t *h, *dh;               // (T, H) bf16
};

// sum of the four waves' partial tiles for THIS wave's output tile (fixed order); part: [wave][tile][register][lane]
template <int NT>
HULC_DEVICE void exchange(const f32x16_t (&acc)[NT], float* part, int w, int lane, f32x16_t& out) {
#pragma unroll
    for (int ot = 0; ot < NT; ++ot)
#pragma unroll
        for (int e = 0; e < 16; ++e) part[((w * NT + ot) * 16 + e) * 64 + lane] = acc[ot][e];
    __syncthreads();
    if (w < NT) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a = part[((0 * NT + w) * 16 + e) * 64 + lane], b = part[((1 * NT + w) * 16 + e) * 64 + lane];
            const float c = part[((2 * NT + w) * 16 + e) * 64 + lane], d = part[((3 * NT + w) * 16 + e) * 64 + lane];
            out[e] = (a + b) + (c + d);
        }
    }
}

// the same for four tiles in two rounds of two (32 KB of LDS): waves 0, 1 end up with tiles 0, 1, waves 2, 3 with tiles 2, 3
HULC_DEVICE void exchange4(const f32x16_t (&acc)[4], float* part, int w, int lane, f32x16_t& out) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (half) __syncthreads();
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) part[((w * 2 + t) * 16 + e) * 64 + lane] = acc[2 * half + t][e];
        __syncthreads();
        if ((w >> 1) == half) {
            const int t = w & 1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float a = part[((0 * 2 + t) * 16 + e) * 64 + lane], b = part[((1 * 2 + t) * 16 + e) * 64 + lane];
                const float c = part[((2 * 2 + t) * 16 + e) * 64 + lane], d = part[((3 * 2 + t) * 16 + e) * 64 + lane];
                out[e] = (a + b) + (c + d);
            }
        }
    }
}

// X3: both products from hi / lo splits of both operands (three bf16 MFMAs each: fp32-class values, the selective-precision site "encfc")
template <int OT, bool X3>          // OT = OUT / 32 output row tiles
__global__ __launch_bounds__(256) void mlp2_fwd_kernel(Mlp2P p) {
    __shared__ float part[4 * OT * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const long tok0 = (long)blockIdx.x * 32;
    const int S = p.T - tok0 < 32 ? (int)(p.T - tok0) : 32;
    bf16x8_t xf[8], xl[X3 ? 8 : 1];
    if constexpr (X3) load_x_frags_hl(xf, xl, p.x, tok0, r, hf, S);
    else load_x_frags(xf, p.x, tok0, r, hf, S);
    f32x16_t acc[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) acc[ot] = zero16();
    for (int s = 0; s < p.H / 128; ++s) {
        const int j0 = s * 128 + 32 * w;
        bf16x8_t w1f[8], w2f[OT * 2], w1l[X3 ? 8 : 1], w2l[X3 ? OT * 2 : 1];
        const uint16_t* a = p.W1 + (long)(j0 + r) * E + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w1f[ks] = ldg16(a + ks * 16);
#pragma unroll
        for (int ot = 0; ot < OT; ++ot)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) w2f[ot * 2 + kk] = ldg_split(p.W2 + (long)(ot * 32 + r) * p.H + j0 + 16 * kk + 4 * hf);
        if constexpr (X3) {
            const uint16_t* al = p.W1lo + (long)(j0 + r) * E + hf * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) w1l[ks] = ldg16(al + ks * 16);
#pragma unroll
            for (int ot = 0; ot < OT; ++ot)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) w2l[ot * 2 + kk] = ldg_split(p.W2lo + (long)(ot * 32 + r) * p.H + j0 + 16 * kk + 4 * hf);
        }
        f32x16_t zT = zero16();                                     // [hidden j0 + arow][row r]
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            zT = MFMA(w1f[ks], xf[ks], zT);
            if constexpr (X3) { zT = MFMA(w1f[ks], xl[ks], zT); zT = MFMA(w1l[ks], xf[ks], zT); }
        }
        add_row_vec(zT, p.b1 + j0, hf, 1.f);
        float hv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) hv[e] = fmaxf(zT[e], 0.f);
        if constexpr (X3) {
            bf16x8_t h0, h0l, h1, h1l;
            pack8f_hl(hv, h0, h0l); pack8f_hl(hv + 8, h1, h1l);
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                acc[ot] = MFMA(w2f[ot * 2], h0, acc[ot]); acc[ot] = MFMA(w2f[ot * 2], h0l, acc[ot]); acc[ot] = MFMA(w2l[ot * 2], h0, acc[ot]);
                acc[ot] = MFMA(w2f[ot * 2 + 1], h1, acc[ot]); acc[ot] = MFMA(w2f[ot * 2 + 1], h1l, acc[ot]); acc[ot] = MFMA(w2l[ot * 2 + 1], h1, acc[ot]);
            }
        } else {
            const bf16x8_t h0 = pack8f(hv), h1 = pack8f(hv + 8);
#pragma unroll
            for (int ot = 0; ot < OT; ++ot) {
                acc[ot] = MFMA(w2f[ot * 2], h0, acc[ot]);
                acc[ot] = MFMA(w2f[ot * 2 + 1], h1, acc[ot]);
            }
        }
    }
    f32x16_t o;
    exchange<OT>(acc, part, w, lane, o);
    if (w < OT && r < S) {                                          // wave w holds output features 32 w + arow(e, hf) of row r
        float* dst = p.y + (tok0 + r) * p.OUT + 32 * w + 4 * hf;
        const float* bb = p.b2 + 32 * w + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 bv = *(const float4*)(bb + 8 * g);
            *(float4*)(dst + 8 * g) = make_float4(o[4 * g] + bv.x, o[4 * g + 1] + bv.y, o[4 * g + 2] + bv.z, o[4 * g + 3] + bv.w);
        }
    }
}

template <int OT>
__global__ __launch_bounds__(256) void mlp2_bwd_kernel(Mlp2P p) {
    __shared__ float part[4 * 2 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const long tok0 = (long)blockIdx.x * 32;
    const int S = p.T - tok0 < 32 ? (int)(p.T - tok0) : 32;
    const bool live = r < S;
    bf16x8_t xf[8];
    load_x_frags(xf, p.x, tok0, r, hf, S);
    // dy^T fragments: k = output feature in the register order of an accumulator tile (16 kk + 4 hf + {0..3, 8..11}), n = row r
    bf16x8_t dff[OT * 2];
    {
        const float* src = p.dy + (tok0 + (live ? r : 0)) * p.OUT + 4 * hf;
#pragma unroll
        for (int kk = 0; kk < OT * 2; ++kk) {
            const float4 lo = *(const float4*)(src + 16 * kk), hi = *(const float4*)(src + 16 * kk + 8);
            const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            Frag f; f.b = pack8f(v);
            if (!live) f.u = make_uint4(0u, 0u, 0u, 0u);
            dff[kk] = f.b;
        }
    }
    f32x16_t acc[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = zero16();
    uint16_t* hrow = p.h + (tok0 + r) * (long)p.H + 4 * hf;
    uint16_t* dhrow = p.dh + (tok0 + r) * (long)p.H + 4 * hf;
    struct Frags { bf16x8_t w1f[8], w2t[OT * 2], w1t[8]; };
    auto load = [&](Frags& f, int s) {
        const int j0 = s * 128 + 32 * w;
        const uint16_t* a = p.W1 + (long)(j0 + r) * E + hf * 8;
        const uint16_t* c = p.W2T + (long)(j0 + r) * p.OUT + 4 * hf;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) f.w1f[ks] = ldg16(a + ks * 16);
#pragma unroll
        for (int kk = 0; kk < OT * 2; ++kk) f.w2t[kk] = ldg_split(c + 16 * kk);
        if (p.dx) {
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) f.w1t[ot * 2 + kk] = ldg_split(p.W1T + (long)(ot * 32 + r) * p.H + j0 + 16 * kk + 4 * hf);
        }
    };
    auto compute = [&](const Frags& f, int s) {
        const int j0 = s * 128 + 32 * w;
        f32x16_t zT = zero16(), dT = zero16();                      // [hidden j0 + arow][row r]
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) zT = MFMA(f.w1f[ks], xf[ks], zT);
#pragma unroll
        for (int kk = 0; kk < OT * 2; ++kk) dT = MFMA(f.w2t[kk], dff[kk], dT);
        add_row_vec(zT, p.b1 + j0, hf, 1.f);
        float hv[16], dv[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool on = zT[e] > 0.f && live;
            hv[e] = on ? zT[e] : 0.f;
            dv[e] = on ? dT[e] : 0.f;
        }
        if (live) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *(uint2*)(hrow + j0 + 8 * g4) = make_uint2(pack_bf16x2(hv[4 * g4], hv[4 * g4 + 1]), pack_bf16x2(hv[4 * g4 + 2], hv[4 * g4 + 3]));
                *(uint2*)(dhrow + j0 + 8 * g4) = make_uint2(pack_bf16x2(dv[4 * g4], dv[4 * g4 + 1]), pack_bf16x2(dv[4 * g4 + 2], dv[4 * g4 + 3]));
            }
        }
        if (p.dx) {
            const bf16x8_t d0 = pack8f(dv), d1 = pack8f(dv + 8);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                acc[ot] = MFMA(f.w1t[ot * 2], d0, acc[ot]);
                acc[ot] = MFMA(f.w1t[ot * 2 + 1], d1, acc[ot]);
            }
        }
    };
    const int NS = p.H / 128;
    Frags fa, fb;
    load(fa, 0);
    for (int s = 0; s < NS; s += 2) {                               // the next slice's weights are in flight under this slice's products
        if (s + 1 < NS) load(fb, s + 1);
        compute(fa, s);
        if (s + 2 < NS) load(fa, s + 2);
        if (s + 1 < NS) compute(fb, s + 1);
    }
    if (!p.dx) return;
    f32x16_t ax;
    exchange4(acc, part, w, lane, ax);
    if (live) {
        float* dst = p.dx + (tok0 + r) * E + 32 * w + 4 * hf;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) *(float4*)(dst + 8 * g4) = make_float4(ax[4 * g4], ax[4 * g4 + 1], ax[4 * g4 + 2], ax[4 * g4 + 3]);
    }
}

int mlp2_check(const float* x, const void* W1, const void* W2, const float* b1, int T, int K, int H, int OUT, const char* who) {
    if (!x || !W1 || !W2 || !b1) return hulc_fail(-1, who);
    if (K != E || H < 128 || H % 128 || OUT < 32 || OUT > 128 || OUT % 32 || T < 1) return hulc_fail(-2, who);
    if ((uintptr_t)x % 16 || (uintptr_t)b1 % 16) return hulc_fail(-4, who);
    return 0;
}

}  // namespace

// see include/hulc2_amd.h
extern "C" int hulc_mlp2_rows_fwd(const float* x, const void* W1, const float* b1, const void* W2, const float* b2, const void* W1_lo, const void* W2_lo,
                                  int T, int K, int H, int OUT, float* y, void* stream) {
    if (int rc = mlp2_check(x, W1, W2, b1, T, K, H, OUT, "hulc_mlp2_rows_fwd: needs K = 128, H a multiple of 128, OUT in {32, 64, 96, 128}, 16-byte aligned operands")) return rc;
    if (!b2 || !y || (uintptr_t)b2 % 16 || (uintptr_t)y % 16) return hulc_fail(-1, "hulc_mlp2_rows_fwd: null or misaligned pointer");
    if ((W1_lo != nullptr) != (W2_lo != nullptr)) return hulc_fail(-3, "hulc_mlp2_rows_fwd: both remainder arrays or none");
    Mlp2P p = {};
    p.x = x; p.W1 = (const uint16_t*)W1; p.W2 = (const uint16_t*)W2; p.b1 = b1; p.b2 = b2; p.T = T; p.H = H; p.OUT = OUT; p.y = y;
    p.W1lo = (const uint16_t*)W1_lo; p.W2lo = (const uint16_t*)W2_lo;
    const unsigned grid = (unsigned)((T + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
    if (W1_lo) {
        switch (OUT / 32) {
            case 1: mlp2_fwd_kernel<1, true><<<grid, 256, 0, s>>>(p); break;
            case 2: mlp2_fwd_kernel<2, true><<<grid, 256, 0, s>>>(p); break;
            case 3: mlp2_fwd_kernel<3, true><<<grid, 256, 0, s>>>(p); break;
            default: mlp2_fwd_kernel<4, true><<<grid, 256, 0, s>>>(p); break;
        }
    } else {
        switch (OUT / 32) {
            case 1: mlp2_fwd_kernel<1, false><<<grid, 256, 0, s>>>(p); break;
            case 2: mlp2_fwd_kernel<2, false><<<grid, 256, 0, s>>>(p); break;
            case 3: mlp2_fwd_kernel<3, false><<<grid, 256, 0, s>>>(p); break;
            default: mlp2_fwd_kernel<4, false><<<grid, 256, 0, s>>>(p); break;
        }
    }
    return hulc_check_launch("hulc_mlp2_rows_fwd");
}

extern "C" int hulc_mlp2_rows_bwd(const float* x, const float* dy, const void* W1, const float* b1, const void* W1T, const void* W2T, int T, int K, int H,
                                  int OUT, float* dx, void* h, void* dh, void* stream) {
    if (int rc = mlp2_check(x, W1, W2T, b1, T, K, H, OUT, "hulc_mlp2_rows_bwd: needs K = 128, H a multiple of 128, OUT in {32, 64, 96, 128}, 16-byte aligned operands")) return rc;
    if (!dy || !W1T || !h || !dh || (uintptr_t)dy % 16 || (uintptr_t)dx % 16 || (uintptr_t)h % 8 || (uintptr_t)dh % 8)
        return hulc_fail(-1, "hulc_mlp2_rows_bwd: null or misaligned pointer");
    Mlp2P p = {};
    p.x = x; p.dy = dy; p.W1 = (const uint16_t*)W1; p.W1T = (const uint16_t*)W1T; p.W2T = (const uint16_t*)W2T; p.b1 = b1; p.T = T; p.H = H; p.OUT = OUT;
    p.dx = dx; p.h = (uint16_t*)h; p.dh = (uint16_t*)dh;
    const unsigned grid = (unsigned)((T + 31) / 32);
    hipStream_t s = (hipStream_t)stream;
    switch (OUT / 32) {
        case 1: mlp2_bwd_kernel<1><<<grid, 256, 0, s>>>(p); break;
        case 2: mlp2_bwd_kernel<2><<<grid, 256, 0, s>>>(p); break;
        case 3: mlp2_bwd_kernel<3><<<grid, 256, 0, s>>>(p); break;
        default: mlp2_bwd_kernel<4><<<grid, 256, 0, s>>>(p); break;
    }
    return hulc_check_launch("hulc_mlp2_rows_bwd");
}
