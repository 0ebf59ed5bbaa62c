// txl_attn.h — the attention half of the plan-recognition transformer layer as device functions: the bodies of hulc_txl_attn_fwd / _bwd
// (txl_fused.hip, one launch per layer half) and the stages of the whole-block launches (txl_block.hip).  See txl_fused.hip for the method.
#pragma once
#include "hulc_common.h"

namespace {


constexpr int E = 128;          // d_model
constexpr int NH = 8;           // heads
constexpr int SMAX = 32;

struct TxlP {
    const float* x;
    const uint16_t *Wqkv, *Wo, *WqkvT, *WoT;
    const uint16_t *Wqkv_lo, *Wo_lo;               // X3 forward: bf16 of the rounding remainders w - bf16(w)
    const float *bqkv, *bo, *gamma, *beta;
    float eps;
    int B, S;
    float drop_p; unsigned long long seed_attn, seed_ln; const unsigned long long* seed_dev;
    float *y, *pre, *mean, *rstd; uint16_t* ctx;
    const float* dy; const float* dy_slab; int n_slab; long slab_stride;
    float* dx; uint16_t* d_o; uint16_t* dqkv; float* ln_partial;
};

HULC_DEVICE int arow(int e, int hf) { return (e & 3) + 8 * (e >> 2) + 4 * hf; }

union Frag { uint4 u; bf16x8_t b; };

template <int BASE>
HULC_DEVICE bf16x8_t pack8(const f32x16_t& a) {
    Frag f;
    f.u = make_uint4(pack_bf16x2(a[BASE], a[BASE + 1]), pack_bf16x2(a[BASE + 2], a[BASE + 3]), pack_bf16x2(a[BASE + 4], a[BASE + 5]),
                     pack_bf16x2(a[BASE + 6], a[BASE + 7]));
    return f.b;
}
HULC_DEVICE bf16x8_t pack8f(const float* a) {
    Frag f;
    f.u = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7]));
    return f.b;
}
// hi = bf16(a), lo = bf16(a - hi): a_hi b_hi + a_lo b_hi + a_hi b_lo reproduces the fp32 product to ~2^-16 (three bf16 MFMAs)
HULC_DEVICE void split2(float a0, float a1, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf16x2(a0, a1);
    lo = pack_bf16x2(a0 - __uint_as_float(hi << 16), a1 - __uint_as_float(hi & 0xffff0000u));
}
HULC_DEVICE void pack8f_hl(const float* a, bf16x8_t& hi, bf16x8_t& lo) {
    Frag h, l;
    split2(a[0], a[1], h.u.x, l.u.x); split2(a[2], a[3], h.u.y, l.u.y); split2(a[4], a[5], h.u.z, l.u.z); split2(a[6], a[7], h.u.w, l.u.w);
    hi = h.b; lo = l.b;
}
template <int BASE>
HULC_DEVICE void pack8_hl(const f32x16_t& a, bf16x8_t& hi, bf16x8_t& lo) {
    const float v[8] = {a[BASE], a[BASE + 1], a[BASE + 2], a[BASE + 3], a[BASE + 4], a[BASE + 5], a[BASE + 6], a[BASE + 7]};
    pack8f_hl(v, hi, lo);
}
HULC_DEVICE bf16x8_t ldg16(const uint16_t* p) { Frag f; f.u = *(const uint4*)p; return f.b; }
// 8 k-slots = two groups of 4 consecutive columns 8 apart (the register order of an accumulator tile, see the header)
HULC_DEVICE bf16x8_t ldg_split(const uint16_t* p) {
    const uint2 lo = *(const uint2*)p, hi = *(const uint2*)(p + 8);
    Frag f; f.u = make_uint4(lo.x, lo.y, hi.x, hi.y);
    return f.b;
}
HULC_DEVICE f32x16_t zero16() {
    f32x16_t a;
#pragma unroll
    for (int e = 0; e < 16; ++e) a[e] = 0.f;
    return a;
}
#define MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)

// x row of this lane's token as 8 fragments (k = feature, natural order); rows >= S are zero
HULC_DEVICE void load_x_frags(bf16x8_t (&xf)[8], const float* x, long tok0, int r, int hf, int S) {
    const float* xr = x + (tok0 + (r < S ? r : 0)) * E + hf * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const float4 a = *(const float4*)(xr + ks * 16), c = *(const float4*)(xr + ks * 16 + 4);
        Frag f;
        f.u = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(c.x, c.y), pack_bf16x2(c.z, c.w));
        if (r >= S) f.u = make_uint4(0u, 0u, 0u, 0u);
        xf[ks] = f.b;
    }
}

// the same row as hi / lo fragment pairs (X3)
HULC_DEVICE void load_x_frags_hl(bf16x8_t (&xh)[8], bf16x8_t (&xl)[8], const float* x, long tok0, int r, int hf, int S) {
    const float* xr = x + (tok0 + (r < S ? r : 0)) * E + hf * 8;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const float4 a = *(const float4*)(xr + ks * 16), c = *(const float4*)(xr + ks * 16 + 4);
        const float v[8] = {a.x, a.y, a.z, a.w, c.x, c.y, c.z, c.w};
        pack8f_hl(v, xh[ks], xl[ks]);
        if (r >= S) { Frag z; z.u = make_uint4(0u, 0u, 0u, 0u); xh[ks] = z.b; xl[ks] = z.b; }
    }
}

// add a per-ROW vector (index 32w + arow(e, hf)) to an accumulator tile: four float4 loads
HULC_DEVICE void add_row_vec(f32x16_t& a, const float* v, int hf, float scale) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 t = *(const float4*)(v + 8 * g + 4 * hf);
        a[4 * g] = (a[4 * g] + t.x) * scale; a[4 * g + 1] = (a[4 * g + 1] + t.y) * scale;
        a[4 * g + 2] = (a[4 * g + 2] + t.z) * scale; a[4 * g + 3] = (a[4 * g + 3] + t.w) * scale;
    }
}

// ---------------------------------------------------------------------------------------------------------------- forward
// cs: 8 KB (ctx fragments [k-step 2w+q][lane half][token]), red: [2][4][32] floats.  Waves 0..3 of the workgroup; the caller separates it from
// other users of the same LDS with a barrier.
// X3: every product from hi / lo splits of both operands (three bf16 MFMAs; needs p.Wqkv_lo / p.Wo_lo and 8 KB more LDS for the lo halves of
// the context fragments): fp32-class forward values on the bf16 matrix pipe — the selective-precision site "txl" 
constexpr int TXL_FWD_LDS = 8 * 2 * 32 * 16 + 2 * 4 * 32 * 4;
constexpr int TXL_FWD_LDS_X3 = TXL_FWD_LDS + 8 * 2 * 32 * 16;
template <bool X3 = false>
HULC_DEVICE void txl_attn_fwd_body(const TxlP& p, const int b, char* lds) {
    uint4* cs = (uint4*)lds;
    float (*red)[4][32] = (float (*)[4][32])(lds + 8 * 2 * 32 * 16);
    uint4* csl = (uint4*)(lds + TXL_FWD_LDS);                 // (X3) lo halves of the ctx fragments
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int S = p.S;
    const long tok0 = (long)b * S;
    const unsigned long long sd = p.seed_dev ? p.seed_dev[0] : 0ull;
    const unsigned long long seed_attn = p.seed_attn ^ sd, seed_ln = p.seed_ln ^ sd;

    bf16x8_t xf[8], xl[X3 ? 8 : 1];
    if constexpr (X3) load_x_frags_hl(xf, xl, p.x, tok0, r, hf, S);
    else load_x_frags(xf, p.x, tok0, r, hf, S);
    // transposed q / k tiles D[col][token] (lane <-> token), plain v tile D[token][col] (lane <-> column)
    f32x16_t qT = zero16(), kT = zero16(), v = zero16();
    {
        // every weight fragment of the three projections is requested before the first MFMA (one memory round trip, not 24)
        const uint16_t* wq = p.Wqkv + (long)(32 * w + r) * E + hf * 8;
        bf16x8_t fq[8], fk[8], fv[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { fq[ks] = ldg16(wq + ks * 16); fk[ks] = ldg16(wq + (long)E * E + ks * 16); fv[ks] = ldg16(wq + 2L * E * E + ks * 16); }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            qT = MFMA(fq[ks], xf[ks], qT);
            kT = MFMA(fk[ks], xf[ks], kT);
            v = MFMA(xf[ks], fv[ks], v);
            if constexpr (X3) { qT = MFMA(fq[ks], xl[ks], qT); kT = MFMA(fk[ks], xl[ks], kT); v = MFMA(xl[ks], fv[ks], v); }
        }
        if constexpr (X3) {                                 // the weights' remainders against the activations' hi parts (same registers, second trip)
            const uint16_t* wl = p.Wqkv_lo + (long)(32 * w + r) * E + hf * 8;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { fq[ks] = ldg16(wl + ks * 16); fk[ks] = ldg16(wl + (long)E * E + ks * 16); fv[ks] = ldg16(wl + 2L * E * E + ks * 16); }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { qT = MFMA(fq[ks], xf[ks], qT); kT = MFMA(fk[ks], xf[ks], kT); v = MFMA(xf[ks], fv[ks], v); }
        }
    }
    // operands of the later phases that do not depend on anything computed here: requested now, consumed behind the barriers
    bf16x8_t wof[8], wol[X3 ? 8 : 1];
    {
        const uint16_t* wo = p.Wo + (long)(32 * w + r) * E + 4 * hf;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wof[kk] = ldg_split(wo + 32 * (kk >> 1) + 16 * (kk & 1));
        if constexpr (X3) {
            const uint16_t* wl = p.Wo_lo + (long)(32 * w + r) * E + 4 * hf;
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) wol[kk] = ldg_split(wl + 32 * (kk >> 1) + 16 * (kk & 1));
        }
    }
    float4 xres[4], bov[4], gmv[4], btv[4];
    {
        const float* xr0 = p.x + (tok0 + (r < S ? r : 0)) * E + 32 * w + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xres[g] = *(const float4*)(xr0 + 8 * g); bov[g] = *(const float4*)(p.bo + 32 * w + 4 * hf + 8 * g);
            gmv[g] = *(const float4*)(p.gamma + 32 * w + 4 * hf + 8 * g); btv[g] = *(const float4*)(p.beta + 32 * w + 4 * hf + 8 * g);
        }
    }
    add_row_vec(qT, p.bqkv + 32 * w, hf, 0.25f);           // (q + bq) / sqrt(16)
    add_row_vec(kT, p.bqkv + E + 32 * w, hf, 1.f);
    {
        const float bv = p.bqkv[2 * E + 32 * w + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] += bv;
    }
    bf16x8_t vf0, vf1, vl0, vl1;
    if constexpr (X3) { pack8_hl<0>(v, vf0, vl0); pack8_hl<8>(v, vf1, vl1); }
    else { vf0 = pack8<0>(v); vf1 = pack8<8>(v); }
    f32x16_t c;                                              // ctx^T: lane <-> query token, registers <-> this wave's 32 columns
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        bf16x8_t qf, kf, ql, kl;
        if constexpr (X3) {
            if (hh) { pack8_hl<8>(qT, qf, ql); pack8_hl<8>(kT, kf, kl); } else { pack8_hl<0>(qT, qf, ql); pack8_hl<0>(kT, kf, kl); }
        } else {
            qf = hh ? pack8<8>(qT) : pack8<0>(qT); kf = hh ? pack8<8>(kT) : pack8<0>(kT);
        }
        f32x16_t s = MFMA(kf, qf, zero16());                // D[key j][query i]
        if constexpr (X3) { s = MFMA(kl, qf, s); s = MFMA(kf, ql, s); }
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) { if (arow(e, hf) >= S) s[e] = -INFINITY; m = fmaxf(m, s[e]); }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { s[e] = __expf(s[e] - m); sum += s[e]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        const long pbase = (((long)b * NH + 2 * w + hh) * S + r) * S;
        float pk[16];
        if (p.drop_p > 0.f) dropout_scale_acc16(seed_attn, (uint64_t)pbase, hf, p.drop_p, pk);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            s[e] *= inv;
            if (p.drop_p > 0.f) s[e] *= pk[e];
        }
        f32x16_t o;
        if constexpr (X3) {
            bf16x8_t p0, p0l, p1, p1l;
            pack8_hl<0>(s, p0, p0l); pack8_hl<8>(s, p1, p1l);
            o = MFMA(vf0, p0, zero16()); o = MFMA(vl0, p0, o); o = MFMA(vf0, p0l, o);
            o = MFMA(vf1, p1, o); o = MFMA(vl1, p1, o); o = MFMA(vf1, p1l, o);
        } else {
            o = MFMA(vf0, pack8<0>(s), zero16());           // D[v column][query]: rows of the OTHER head of this wave are garbage
            o = MFMA(vf1, pack8<8>(s), o);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) c[8 * hh + e] = o[8 * hh + e];
    }
    if (p.ctx && r < S) {                                    // row-major bf16 copy: the operand of the out_proj weight gradient
        uint16_t* dst = p.ctx + (tok0 + r) * E + 32 * w + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) *(uint2*)(dst + 8 * g) = make_uint2(pack_bf16x2(c[4 * g], c[4 * g + 1]), pack_bf16x2(c[4 * g + 2], c[4 * g + 3]));
    }
    if constexpr (X3) {
        Frag fh, fl;
        pack8_hl<0>(c, fh.b, fl.b); cs[((2 * w) * 2 + hf) * 32 + r] = fh.u; csl[((2 * w) * 2 + hf) * 32 + r] = fl.u;
        pack8_hl<8>(c, fh.b, fl.b); cs[((2 * w + 1) * 2 + hf) * 32 + r] = fh.u; csl[((2 * w + 1) * 2 + hf) * 32 + r] = fl.u;
    } else { Frag f; f.b = pack8<0>(c); cs[((2 * w) * 2 + hf) * 32 + r] = f.u; f.b = pack8<8>(c); cs[((2 * w + 1) * 2 + hf) * 32 + r] = f.u; }
    __syncthreads();
    // out_proj for output features 32w..32w+31: D[feature][token] = Wo (columns in fragment order) x ctx^T
    f32x16_t o = zero16();
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        Frag f; f.u = cs[(kk * 2 + hf) * 32 + r];
        o = MFMA(wof[kk], f.b, o);
        if constexpr (X3) { Frag g; g.u = csl[(kk * 2 + hf) * 32 + r]; o = MFMA(wof[kk], g.b, o); o = MFMA(wol[kk], f.b, o); }
    }
    // residual + dropout + LayerNorm over the 128 features of token r (spread over 4 waves x 2 lane halves x 16 registers)
    float pre[16];
    float s1 = 0.f;
    float lk[16];
    if (p.drop_p > 0.f) dropout_scale_acc16(seed_ln, (uint64_t)((tok0 + r) * E + 32 * w), hf, p.drop_p, lk);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 xv = xres[g], bv = bov[g];
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ba[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * g + q;
            float ov = o[e] + ba[q];
            if (p.drop_p > 0.f) ov *= lk[e];
            pre[e] = xa[q] + ov;
            s1 += pre[e];
        }
    }
    s1 += __shfl_xor(s1, 32, 64);
    if (hf == 0) red[0][w][r] = s1;
    __syncthreads();
    const float mean = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) * (1.0f / E);
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = pre[e] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 32, 64);
    if (hf == 0) red[1][w][r] = s2;
    __syncthreads();
    const float rstd = rsqrtf(((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) * (1.0f / E) + p.eps);
    if (r < S) {
        float* yr = p.y + (tok0 + r) * E + 32 * w + 4 * hf;
        float* pr = p.pre ? p.pre + (tok0 + r) * E + 32 * w + 4 * hf : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 gv = gmv[g], bt = btv[g];
            float4 out;
            out.x = (pre[4 * g] - mean) * rstd * gv.x + bt.x; out.y = (pre[4 * g + 1] - mean) * rstd * gv.y + bt.y;
            out.z = (pre[4 * g + 2] - mean) * rstd * gv.z + bt.z; out.w = (pre[4 * g + 3] - mean) * rstd * gv.w + bt.w;
            *(float4*)(yr + 8 * g) = out;
            if (pr) *(float4*)(pr + 8 * g) = make_float4(pre[4 * g], pre[4 * g + 1], pre[4 * g + 2], pre[4 * g + 3]);
        }
        if (w == 0 && hf == 0 && p.mean) { p.mean[tok0 + r] = mean; p.rstd[tok0 + r] = rstd; }
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward
// dy (+ the dy_slab partials, e.g. the 16 hidden-slice input gradients of the fused feed-forward block) -> LayerNorm1 backward ->
// out_proj / attention / in_proj data gradients -> dx = dpre + dqkv Wqkv.  Weight gradients are left to two GEMMs over all tokens
// (dWo = d_o^T ctx, dWqkv = dqkv^T x): this kernel stores d_o and dqkv row-major in bf16 and the per-sequence LayerNorm partials.
constexpr int TXL_BWD_LDS = 8 * 2 * 32 * 16 + 4 * 2 * 32 * 33 * 4 + 2 * 4 * 32 * 4 + 4 * 3 * 32 * 4;
// REGS (the whole-block launch): the incoming gradient arrives in registers (dyr[e] <-> feature 32 w + arow(e, hf) of token r, zero for
// r >= S) and the input gradient leaves the same way (dxr) instead of through p.dy / p.dx
template <bool REGS = false>
HULC_DEVICE void txl_attn_bwd_body(const TxlP& p, const int b, char* lds, const float* dyr = nullptr, float* dxr = nullptr) {
    uint4* dos = (uint4*)lds;                  // d_o fragments   [k-step][lane half][token]
    // dqkv fragments [k-step][lane half][token] (24 KB, written last) share their storage with the LayerNorm parameter-gradient
    // transposes (33 KB, dead after the first barrier)
    char* pool = lds + 8 * 2 * 32 * 16;
    uint4* dqs = (uint4*)pool;
    float (*lt)[2][32][33] = (float (*)[2][32][33])pool;
    float (*red)[4][32] = (float (*)[4][32])(pool + 4 * 2 * 32 * 33 * 4);
    float (*st)[3][32] = (float (*)[3][32])(pool + 4 * 2 * 32 * 33 * 4 + 2 * 4 * 32 * 4);   // per wave: softmax max, 1/sum, sum_j dP P of the current head
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int S = p.S;
    const long tok0 = (long)b * S;
    const unsigned long long sd = p.seed_dev ? p.seed_dev[0] : 0ull;
    const unsigned long long seed_attn = p.seed_attn ^ sd, seed_ln = p.seed_ln ^ sd;
    const bool live = r < S;
    const long tok = tok0 + (live ? r : 0);

    // Wo^T fragments (consumed behind the first barriers) are requested before anything else
    bf16x8_t wotf[8];
    {
        const uint16_t* wt = p.WoT + (long)(32 * w + r) * E + 4 * hf;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) wotf[kk] = ldg_split(wt + 32 * (kk >> 1) + 16 * (kk & 1));
    }
    // ---- LayerNorm1 backward in the (lane <-> token, registers <-> features 32w + arow) layout
    float dpre[16];
    {
        const long off = tok * E + 32 * w + 4 * hf;
        const float mean = p.mean[tok], rstd = p.rstd[tok];
        float g[16], xh[16], dyv[16];
        float s1 = 0.f, s2 = 0.f;
        // dy + the partial slabs: all loads of a trip are issued before any is consumed (4 slabs x 4 pieces in flight; a one-load-per-trip
        // accumulation chain cost 64 serial memory round trips here: 40 us per launch)
        float4 dsum[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            if constexpr (REGS) dsum[q4] = make_float4(dyr[4 * q4], dyr[4 * q4 + 1], dyr[4 * q4 + 2], dyr[4 * q4 + 3]);
            else dsum[q4] = *(const float4*)(p.dy + off + 8 * q4);
        }
        if constexpr (!REGS) {
            int sl = 0;
            for (; sl + 3 < p.n_slab; sl += 4) {
                float4 t[4][4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) t[u][q4] = *(const float4*)(p.dy_slab + (long)(sl + u) * p.slab_stride + off + 8 * q4);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int q4 = 0; q4 < 4; ++q4) { dsum[q4].x += t[u][q4].x; dsum[q4].y += t[u][q4].y; dsum[q4].z += t[u][q4].z; dsum[q4].w += t[u][q4].w; }
            }
            for (; sl < p.n_slab; ++sl)
#pragma unroll
                for (int q4 = 0; q4 < 4; ++q4) {
                    const float4 a = *(const float4*)(p.dy_slab + (long)sl * p.slab_stride + off + 8 * q4);
                    dsum[q4].x += a.x; dsum[q4].y += a.y; dsum[q4].z += a.z; dsum[q4].w += a.w;
                }
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 d = dsum[q4];
            const float4 pv = *(const float4*)(p.pre + off + 8 * q4), gm = *(const float4*)(p.gamma + 32 * w + 4 * hf + 8 * q4);
            const float da[4] = {d.x, d.y, d.z, d.w}, pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = 4 * q4 + q;
                dyv[e] = live ? da[q] : 0.f;
                xh[e] = (pa[q] - mean) * rstd;
                g[e] = dyv[e] * ga[q];
                s1 += g[e]; s2 += g[e] * xh[e];
                lt[w][0][arow(e, hf)][r] = dyv[e] * xh[e];
                lt[w][1][arow(e, hf)][r] = dyv[e];
            }
        }
        s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
        if (hf == 0) { red[0][w][r] = s1; red[1][w][r] = s2; }
        __syncthreads();
        s1 = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) * (1.0f / E);
        s2 = ((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) * (1.0f / E);
        float dov[16], lk[16];
        if (p.drop_p > 0.f) dropout_scale_acc16(seed_ln, (uint64_t)((tok0 + r) * E + 32 * w), hf, p.drop_p, lk);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            dpre[e] = rstd * (g[e] - s1 - xh[e] * s2);
            dov[e] = dpre[e];
            if (p.drop_p > 0.f) dov[e] *= lk[e];
        }
        { Frag f; f.b = pack8f(dov); dos[((2 * w) * 2 + hf) * 32 + r] = f.u; f.b = pack8f(dov + 8); dos[((2 * w + 1) * 2 + hf) * 32 + r] = f.u; }
        if (live) {
            uint16_t* dst = p.d_o + (tok0 + r) * E + 32 * w + 4 * hf;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4)
                *(uint2*)(dst + 8 * q4) = make_uint2(pack_bf16x2(dov[4 * q4], dov[4 * q4 + 1]), pack_bf16x2(dov[4 * q4 + 2], dov[4 * q4 + 3]));
        }
        // dgamma / dbeta partials of this sequence: lane (feature r of this wave, array hf) sums over the 32 tokens
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 32; ++i) acc += lt[w][hf][r][i];
        p.ln_partial[((long)b * 2 + hf) * E + 32 * w + r] = acc;
    }
    // operands of the projection recompute: independent of the exchange, requested before the barrier
    bf16x8_t xf[8], fq[8], fk[8], fv[8];
    load_x_frags(xf, p.x, tok0, r, hf, S);
    {
        const uint16_t* wq = p.Wqkv + (long)(32 * w + r) * E + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { fq[ks] = ldg16(wq + ks * 16); fk[ks] = ldg16(wq + (long)E * E + ks * 16); fv[ks] = ldg16(wq + 2L * E * E + ks * 16); }
    }
    __syncthreads();

    // ---- dctx = d_o Wo in both orientations, for this wave's 32 context columns
    f32x16_t dcT = zero16(), dc = zero16();     // dcT: lane <-> token, registers <-> column;  dc: lane <-> column, registers <-> token
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
        Frag f; f.u = dos[(kk * 2 + hf) * 32 + r];
        dcT = MFMA(wotf[kk], f.b, dcT);
        dc = MFMA(f.b, wotf[kk], dc);
    }
    // ---- recompute the projections (both orientations where both are consumed)
    f32x16_t qT = zero16(), kT = zero16(), vT = zero16(), qn = zero16(), kn = zero16();
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        qT = MFMA(fq[ks], xf[ks], qT); qn = MFMA(xf[ks], fq[ks], qn);
        kT = MFMA(fk[ks], xf[ks], kT); kn = MFMA(xf[ks], fk[ks], kn);
        vT = MFMA(fv[ks], xf[ks], vT);
    }
    add_row_vec(qT, p.bqkv + 32 * w, hf, 0.25f);
    add_row_vec(kT, p.bqkv + E + 32 * w, hf, 1.f);
    add_row_vec(vT, p.bqkv + 2 * E + 32 * w, hf, 1.f);
    {
        const float bq = p.bqkv[32 * w + r], bk = p.bqkv[E + 32 * w + r];
#pragma unroll
        for (int e = 0; e < 16; ++e) { qn[e] = (qn[e] + bq) * 0.25f; kn[e] += bk; }
    }
    const bf16x8_t qnf0 = pack8<0>(qn), qnf1 = pack8<8>(qn), knf0 = pack8<0>(kn), knf1 = pack8<8>(kn);
    const bf16x8_t dcf0 = pack8<0>(dc), dcf1 = pack8<8>(dc);
    f32x16_t gq, gk, gv;                       // dq^T, dk^T, dv^T: lane <-> token, registers <-> this wave's 32 columns
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) {
        const bf16x8_t qf = hh ? pack8<8>(qT) : pack8<0>(qT), kf = hh ? pack8<8>(kT) : pack8<0>(kT);
        const bf16x8_t vf = hh ? pack8<8>(vT) : pack8<0>(vT), df = hh ? pack8<8>(dcT) : pack8<0>(dcT);
        f32x16_t sT = MFMA(kf, qf, zero16());               // [key][query]: lane <-> query
        f32x16_t s2 = MFMA(qf, kf, zero16());               // [query][key]: lane <-> key
        const f32x16_t dpT = MFMA(vf, df, zero16());        // dP^T [key][query]
        const f32x16_t dp2 = MFMA(df, vf, zero16());        // dP   [query][key]
        const long hb = ((long)b * NH + 2 * w + hh) * S;
        // orientation 1 (lane <-> query r): softmax statistics and dS^T
        float m = -INFINITY;
#pragma unroll
        for (int e = 0; e < 16; ++e) { if (arow(e, hf) >= S) sT[e] = -INFINITY; m = fmaxf(m, sT[e]); }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { sT[e] = __expf(sT[e] - m); sum += sT[e]; }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = 1.0f / sum;
        float mk[16], rs = 0.f;
        if (p.drop_p > 0.f) dropout_scale_acc16(seed_attn, (uint64_t)((hb + r) * S), hf, p.drop_p, mk);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            sT[e] *= inv;                                                                       // P (before dropout)
            if (!(p.drop_p > 0.f)) mk[e] = 1.f;
            rs += dpT[e] * mk[e] * sT[e];
        }
        rs += __shfl_xor(rs, 32, 64);
        if (hf == 0) { st[w][0][r] = m; st[w][1][r] = inv; st[w][2][r] = rs; }
        float dsT[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) dsT[e] = sT[e] * (dpT[e] * mk[e] - rs);
        __syncthreads();
        // orientation 2 (lane <-> key r, registers <-> query arow): the same P, its dropout mask and dS from the shared row statistics
        float ds2[16], pd2[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = arow(e, hf);
            const float pv = r < S ? __expf(s2[e] - st[w][0][i]) * st[w][1][i] : 0.f;
            const float mk2 = p.drop_p > 0.f ? dropout_scale(seed_attn, (uint64_t)((hb + i) * S + r), p.drop_p) : 1.f;
            pd2[e] = pv * mk2;                                                                  // post-dropout probabilities
            ds2[e] = pv * (dp2[e] * mk2 - st[w][2][i]);
        }
        __syncthreads();                                                                        // st is reused by the next head
        f32x16_t t = MFMA(dcf0, pack8f(pd2), zero16());     // dv^T [column][key]
        t = MFMA(dcf1, pack8f(pd2 + 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) gv[8 * hh + e] = t[8 * hh + e];
        t = MFMA(knf0, pack8f(dsT), zero16());              // dq_scaled^T [column][query]
        t = MFMA(knf1, pack8f(dsT + 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) gq[8 * hh + e] = t[8 * hh + e] * 0.25f;
        t = MFMA(qnf0, pack8f(ds2), zero16());              // dk^T [column][key]
        t = MFMA(qnf1, pack8f(ds2 + 8), t);
#pragma unroll
        for (int e = 0; e < 8; ++e) gk[8 * hh + e] = t[8 * hh + e];
    }
    // ---- dqkv: fragments to the other waves, row-major bf16 to memory
    {
        Frag f;
        f.b = pack8<0>(gq); dqs[((2 * w) * 2 + hf) * 32 + r] = f.u;      f.b = pack8<8>(gq); dqs[((2 * w + 1) * 2 + hf) * 32 + r] = f.u;
        f.b = pack8<0>(gk); dqs[((8 + 2 * w) * 2 + hf) * 32 + r] = f.u;  f.b = pack8<8>(gk); dqs[((9 + 2 * w) * 2 + hf) * 32 + r] = f.u;
        f.b = pack8<0>(gv); dqs[((16 + 2 * w) * 2 + hf) * 32 + r] = f.u; f.b = pack8<8>(gv); dqs[((17 + 2 * w) * 2 + hf) * 32 + r] = f.u;
        if (live) {
            uint16_t* dst = p.dqkv + (tok0 + r) * 3 * E + 32 * w + 4 * hf;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *(uint2*)(dst + 8 * g4) = make_uint2(pack_bf16x2(gq[4 * g4], gq[4 * g4 + 1]), pack_bf16x2(gq[4 * g4 + 2], gq[4 * g4 + 3]));
                *(uint2*)(dst + E + 8 * g4) = make_uint2(pack_bf16x2(gk[4 * g4], gk[4 * g4 + 1]), pack_bf16x2(gk[4 * g4 + 2], gk[4 * g4 + 3]));
                *(uint2*)(dst + 2 * E + 8 * g4) = make_uint2(pack_bf16x2(gv[4 * g4], gv[4 * g4 + 1]), pack_bf16x2(gv[4 * g4 + 2], gv[4 * g4 + 3]));
            }
        }
    }
    // Wqkv^T fragments (columns in fragment order): requested before the barrier they are consumed behind
    bf16x8_t wtf[24];
    {
        const uint16_t* wt = p.WqkvT + (long)(32 * w + r) * 3 * E + 4 * hf;
#pragma unroll
        for (int kk = 0; kk < 24; ++kk) wtf[kk] = ldg_split(wt + E * (kk >> 3) + 32 * ((kk & 7) >> 1) + 16 * (kk & 1));
    }
    __syncthreads();
    // ---- dx^T [feature][token] = Wqkv^T x dqkv^T, + the residual path
    f32x16_t ax = zero16();
#pragma unroll
    for (int kk = 0; kk < 24; ++kk) {
        Frag f; f.u = dqs[(kk * 2 + hf) * 32 + r];
        ax = MFMA(wtf[kk], f.b, ax);
    }
    if constexpr (REGS) {
#pragma unroll
        for (int e = 0; e < 16; ++e) dxr[e] = live ? dpre[e] + ax[e] : 0.f;
    } else if (live) {
        float* dst = p.dx + (tok0 + r) * E + 32 * w + 4 * hf;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
            *(float4*)(dst + 8 * g4) = make_float4(dpre[4 * g4] + ax[4 * g4], dpre[4 * g4 + 1] + ax[4 * g4 + 1], dpre[4 * g4 + 2] + ax[4 * g4 + 2],
                                                   dpre[4 * g4 + 3] + ax[4 * g4 + 3]);
    }
}

}  // namespace
