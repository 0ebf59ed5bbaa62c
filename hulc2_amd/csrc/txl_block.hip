// txl_block.hip — the plan-recognition transformer trunk as ONE launch per direction (bf16 MFMA).
//
// reference arithmetic: PlanRecognitionTransformersNetwork.forward up to the sequence mean, hulc2/models/plan_encoders/plan_recognition_net.py:125-146 —
//     x0 = dropout(emb + pos[arange(S)]);  L x nn.TransformerEncoderLayer (post-norm, d_model 128, 8 heads, ReLU feed-forward, dropout p);  mean over S
// and its autograd backward.  Everything up to the mean is independent per sequence, so one workgroup (4 waves) owns one sequence for the
// whole trunk: the attention halves are the bodies of hulc_txl_attn_fwd / _bwd (txl_attn.h), the feed-forward halves are new here.
//
// Feed-forward half on 32 tokens without LDS traffic in its loop: wave w owns 32 of every 128 hidden units.  It computes the TRANSPOSED
// hidden tile z^T[hidden][token] = W1s y1^T (accumulator: lane <-> token, registers <-> hidden units), applies bias / ReLU / dropout in
// registers, and — the accumulator-as-operand trick of txl_fused.hip — feeds the packed tile straight back as the B operand of
// f^T[out][token] += W2[out][hidden] h^T[hidden][token] with W2's columns read in the register order.  Four 32-row output tiles accumulate
// over all hidden units of the wave; the four waves' partial tiles meet once per layer in LDS (fixed order), after which wave w holds
// output features 32w.. in exactly the (lane <-> token, registers <-> features) layout the residual + dropout + LayerNorm epilogue of the
// attention half uses.  Backward is the mirror image: z^T recomputed, dh^T = (W2^T df^T) * gate, dy1^T += W1^T dh^T accumulated in registers,
// h and dh stored once (bf16, token-major) as the operands of the weight-gradient products, which join the pass's grouped launch
// (wgrad_group.hip) together with the attention half's.
//
// Stages hand tensors to each other through memory that backward keeps anyway (y1, y2, ...): written and re-read by the same workgroup
// (same CU, L1 / L2 resident) with a workgroup barrier in between.
//
// One workgroup per sequence uses B of the 256 CUs, and the feed-forward half is most of the work (2 x 2048 x 128 MACs and 2048 dropout
// draws per token).  While B Q <= 256, Q = 2 or 4 workgroups SHARE a sequence: each runs the (cheap) attention half redundantly — same
// inputs, same instructions, the same bits — and takes 1 / Q of the hidden units in the feed-forward half; the Q partial output tiles
// (32 x 128 fp32) are exchanged through memory (write-through stores, one arrival counter per sequence, plain loads from a region used
// once per launch: the protocol of mlp_chain.hip) and summed in workgroup order by all Q, which then continue in lock step.  Tensors every
// member computes are stored by every member (identical values).  Like mlp_chain this needs its workgroups co-resident: the launcher picks
// Q > 1 only when the whole grid fits the device, and a member that waits too long sets bit 2 of the sticky fault word.
#include <stdio.h>
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include "txl_attn.h"
#include <stdlib.h>

namespace {

typedef hulc_txl_block_desc BlockP;
typedef hulc_txl_block_layer LayerP;

constexpr int PART_BYTES = 4 * 4 * 16 * 64 * 4;          // the four waves' partial output tiles [wave][tile][register][lane] fp32 = 64 KB
constexpr int LT_BYTES = 4 * 2 * 32 * 33 * 4;            // LayerNorm parameter-gradient transposes (as in txl_attn_bwd_body)
constexpr int BLOCK_LDS = PART_BYTES + 2 * 4 * 32 * 4;   // + red[2][4][32]
constexpr int SYNC_STRIDE = 16;                          // unsigned words between the sequences' arrival counters (64 B)
constexpr int SYNC_MAX_SEQ = 128;                        // sequences that can be shared (B Q <= 256 workgroups, Q >= 2)
constexpr long SYNC_BYTES = (long)SYNC_MAX_SEQ * SYNC_STRIDE * 4;   // FIXED size of the counter area: one workspace serves launches of any B

// kernel-side view of the launch: the C description + the sharing factor and the exchange workspace
struct BlockK {
    BlockP d;
    int Q;                      // workgroups per sequence (1, 2, 4)
    unsigned* sync;             // [B][SYNC_STRIDE]: word 0 arrivals, word 1 members that finished
    float* xpart;               // [L][B][Q][4 waves][16][64] partial tiles
    int* err_sticky;
    int dbg;                    // timing probes (HULC_TXL_DBG, results invalid): 1 skip the attention stages, 2 the feed-forward loops, 4 the exchanges
};

HULC_DEVICE TxlP attn_params(const BlockP& d, const LayerP& l) {
    TxlP p = {};
    p.x = l.x; p.Wqkv = (const uint16_t*)l.Wqkv; p.Wo = (const uint16_t*)l.Wo; p.WqkvT = (const uint16_t*)l.WqkvT; p.WoT = (const uint16_t*)l.WoT;
    p.Wqkv_lo = (const uint16_t*)l.Wqkv_lo; p.Wo_lo = (const uint16_t*)l.Wo_lo;
    p.bqkv = l.bqkv; p.bo = l.bo; p.gamma = l.g1; p.beta = l.be1; p.eps = d.eps; p.B = d.B; p.S = d.S;
    p.drop_p = d.drop_p; p.seed_attn = l.seed_attn; p.seed_ln = l.seed_ln1; p.seed_dev = d.seed_dev;
    p.y = l.y1; p.pre = l.pre1; p.mean = l.mean1; p.rstd = l.rstd1; p.ctx = (uint16_t*)l.ctx;
    p.dy = nullptr; p.dy_slab = nullptr; p.n_slab = 0; p.slab_stride = 0;
    p.dx = nullptr; p.d_o = (uint16_t*)l.d_o; p.dqkv = (uint16_t*)l.dqkv; p.ln_partial = l.lnp1;
    return p;
}

// sum of the four waves' partial tiles for THIS wave's output tile (fixed order); part: [wave][tile][register][lane]
HULC_DEVICE void exchange_tiles(const f32x16_t (&acc)[4], float* part, int w, int lane, f32x16_t& out) {
#pragma unroll
    for (int ot = 0; ot < 4; ++ot)
#pragma unroll
        for (int e = 0; e < 16; ++e) part[((w * 4 + ot) * 16 + e) * 64 + lane] = acc[ot][e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const float a = part[((0 * 4 + w) * 16 + e) * 64 + lane], b = part[((1 * 4 + w) * 16 + e) * 64 + lane];
        const float c = part[((2 * 4 + w) * 16 + e) * 64 + lane], d = part[((3 * 4 + w) * 16 + e) * 64 + lane];
        out[e] = (a + b) + (c + d);
    }
}

// sum of the Q members' partial tiles, in member order (every member ends with the same bits).  sync_no: 1, 2, ... within the launch.
HULC_DEVICE void quad_exchange(const BlockK& k, f32x16_t& o, int b, int q, int li, int sync_no) {
    if (k.Q == 1 || (k.dbg & 4)) return;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float* base = k.xpart + ((long)li * k.d.B + b) * k.Q * (4 * 16 * 64);
    float* mine = base + (long)q * (4 * 16 * 64) + (w * 16) * 64 + lane;
#pragma unroll
    for (int e = 0; e < 16; ++e) __hip_atomic_store(mine + e * 64, o[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        unsigned* ctr = k.sync + (long)b * SYNC_STRIDE;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned want = (unsigned)(k.Q * sync_no);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1L << 22)) {
                if (k.err_sticky) __hip_atomic_fetch_or(k.err_sticky, 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // bit 2 = txl_block
                break;
            }
        }
    }
    __syncthreads();
    asm volatile("" ::: "memory");
    const float* src = base + (w * 16) * 64 + lane;
    if (k.Q == 4) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const float a = src[e * 64], c = src[4096 + e * 64], f = src[2 * 4096 + e * 64], g = src[3 * 4096 + e * 64];
            o[e] = (a + c) + (f + g);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) o[e] = src[e * 64] + src[4096 + e * 64];
    }
}

// behind the last exchange of a launch: the member that gets here last puts the sequence's counters back to zero
HULC_DEVICE void quad_finish(const BlockK& k, int b) {
    if (k.Q == 1 || (k.dbg & 4) || threadIdx.x != 0) return;
    unsigned* ctr = k.sync + (long)b * SYNC_STRIDE;
    if (__hip_atomic_fetch_add(ctr + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(k.Q - 1)) {
        __hip_atomic_store(ctr, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(ctr + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// workgroup -> (sequence, member): the members of a sequence sit on one XCD (blockIdx % 8), next to each other in its dispatch order
HULC_DEVICE bool quad_ids(const BlockK& k, int& b, int& q) {
    if (k.Q == 1) { b = blockIdx.x; q = 0; return true; }
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    q = slot % k.Q;
    b = (slot / k.Q) * 8 + xcd;
    return b < k.d.B;
}

// ---------------------------------------------------------------------------------------------------------------- forward stages
// x0 = dropout(emb + pos[pos_ids[s]])   (add_pos_fwd_kernel's arithmetic and dropout stream)
HULC_DEVICE void pos_add_seq(const BlockP& d, int b) {
    const int tid = threadIdx.x, S = d.S;
    const unsigned long long seed = d.seed_pos ^ (d.seed_dev ? d.seed_dev[0] : 0ull);
    float* x0 = d.layers[0].x;
    for (int i4 = tid; i4 < S * (E / 4); i4 += 256) {
        const int s = i4 / (E / 4), c = (i4 % (E / 4)) * 4;
        const long i = ((long)b * S + s) * E + c;
        const float4 a = *(const float4*)(d.emb + i), q = *(const float4*)(d.pos + d.pos_ids[s] * E + c);
        float v[4] = {a.x + q.x, a.y + q.y, a.z + q.z, a.w + q.w};
        if (d.drop_p > 0.f) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] *= dropout_scale(seed, (uint64_t)(i + j), d.drop_p);
        }
        *(float4*)(x0 + i) = make_float4(v[0], v[1], v[2], v[3]);
    }
}

struct FfnFrags { bf16x8_t w1[8], w2[8]; };

// y2 = LayerNorm2(y1 + dropout(W2 dropout(relu(W1 y1 + b1)) + b2))  for the 32 tokens of sequence b
// X3: both products from hi / lo splits of both operands (three bf16 MFMAs each; packed weights and their remainders required)
template <bool X3>
HULC_DEVICE void ffn_fwd_seq(const BlockK& k, const LayerP& l, int b, int q, int li, char* lds, bool stash) {
    const BlockP& d = k.d;
    float* part = (float*)lds;
    float (*red)[4][32] = (float (*)[4][32])(lds + PART_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int S = d.S, FF = d.FF;
    const long tok0 = (long)b * S;
    const unsigned long long sd = d.seed_dev ? d.seed_dev[0] : 0ull;
    const unsigned long long seed_ffn = l.seed_ffn ^ sd, seed_ln = l.seed_ln2 ^ sd;
    const uint16_t* W1 = (const uint16_t*)l.W1;
    const uint16_t* W2 = (const uint16_t*)l.W2;

    bf16x8_t xf[8], xl[X3 ? 8 : 1];
    if constexpr (X3) load_x_frags_hl(xf, xl, l.y1, tok0, r, hf, S);
    else load_x_frags(xf, l.y1, tok0, r, hf, S);
    f32x16_t acc[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = zero16();

    const uint16_t* W1p = (const uint16_t*)l.W1p;
    const uint16_t* W2p = (const uint16_t*)l.W2p;
    auto load = [&](FfnFrags& f, int s) {
        const int j0 = s * 128 + 32 * w;
        if (W1p && W2p) {                                           // fragment-packed copies: 1 KB contiguous per load instruction
            const long u = ((long)(j0 / 32) * 8 * 64 + lane) * 8;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { f.w1[q8] = ldg16(W1p + u + q8 * 512); f.w2[q8] = ldg16(W2p + u + q8 * 512); }
            return;
        }
        const uint16_t* a = W1 + (long)(j0 + r) * E + hf * 8;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) f.w1[ks] = ldg16(a + ks * 16);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot)
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) f.w2[ot * 2 + kk] = ldg_split(W2 + (long)(ot * 32 + r) * FF + j0 + 16 * kk + 4 * hf);
    };
    auto compute = [&](const FfnFrags& f, int s) {
        const int j0 = s * 128 + 32 * w;
        f32x16_t zT = zero16();                                   // [hidden j0 + arow][token r]
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) zT = MFMA(f.w1[ks], xf[ks], zT);
        FfnFrags fl;                                                // (X3) the weights' remainders of this slice
        if constexpr (X3) {
            const long u = ((long)(j0 / 32) * 8 * 64 + lane) * 8;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) { fl.w1[q8] = ldg16((const uint16_t*)l.W1p_lo + u + q8 * 512); fl.w2[q8] = ldg16((const uint16_t*)l.W2p_lo + u + q8 * 512); }
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { zT = MFMA(f.w1[ks], xl[ks], zT); zT = MFMA(fl.w1[ks], xf[ks], zT); }
        }
        add_row_vec(zT, l.b1 + j0, hf, 1.f);
        float hv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                            // registers 4 g4 .. 4 g4 + 3 = four consecutive hidden units: one draw
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (d.drop_p > 0.f) dropout_scale4(seed_ffn, (uint64_t)(tok0 + r) * (uint64_t)FF + (uint64_t)(j0 + 8 * g4 + 4 * hf), d.drop_p, keep);
#pragma unroll
            for (int i = 0; i < 4; ++i) hv[4 * g4 + i] = fmaxf(zT[4 * g4 + i], 0.f) * keep[i];
        }
        if constexpr (X3) {
            bf16x8_t h0, h0l, h1, h1l;
            pack8f_hl(hv, h0, h0l); pack8f_hl(hv + 8, h1, h1l);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                acc[ot] = MFMA(f.w2[ot * 2], h0, acc[ot]); acc[ot] = MFMA(f.w2[ot * 2], h0l, acc[ot]); acc[ot] = MFMA(fl.w2[ot * 2], h0, acc[ot]);
                acc[ot] = MFMA(f.w2[ot * 2 + 1], h1, acc[ot]); acc[ot] = MFMA(f.w2[ot * 2 + 1], h1l, acc[ot]); acc[ot] = MFMA(fl.w2[ot * 2 + 1], h1, acc[ot]);
            }
        } else {
            const bf16x8_t h0 = pack8f(hv), h1 = pack8f(hv + 8);
#pragma unroll
            for (int ot = 0; ot < 4; ++ot) {
                acc[ot] = MFMA(f.w2[ot * 2], h0, acc[ot]);
                acc[ot] = MFMA(f.w2[ot * 2 + 1], h1, acc[ot]);
            }
        }
    };
    const int NS = (k.dbg & 2) ? 0 : FF / 128 / k.Q, s0 = q * NS, s1 = s0 + NS;     // this member's hidden slices
    FfnFrags fa, fb;
    if (NS) load(fa, s0);
    for (int s = s0; s < s1; s += 2) {                             // the next slice's weights are in flight under this slice's products
        if (s + 1 < s1) load(fb, s + 1);
        compute(fa, s);
        if (s + 2 < s1) load(fa, s + 2);
        if (s + 1 < s1) compute(fb, s + 1);
    }
    // epilogue operands, requested before the exchange
    float4 xres[4], b2v[4], gmv[4], btv[4];
    {
        const float* xr0 = l.y1 + (tok0 + (r < S ? r : 0)) * E + 32 * w + 4 * hf;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            xres[g] = *(const float4*)(xr0 + 8 * g); b2v[g] = *(const float4*)(l.b2 + 32 * w + 4 * hf + 8 * g);
            gmv[g] = *(const float4*)(l.g2 + 32 * w + 4 * hf + 8 * g); btv[g] = *(const float4*)(l.be2 + 32 * w + 4 * hf + 8 * g);
        }
    }
    f32x16_t o;
    exchange_tiles(acc, part, w, lane, o);
    quad_exchange(k, o, b, q, li, li + 1);
    // residual + dropout + LayerNorm2 over the 128 features of token r (4 waves x 2 lane halves x 16 registers)
    float pre[16];
    float sum1 = 0.f;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const float4 xv = xres[g], bv = b2v[g];
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ba[4] = {bv.x, bv.y, bv.z, bv.w};
        float keep[4] = {1.f, 1.f, 1.f, 1.f};
        if (d.drop_p > 0.f) dropout_scale4(seed_ln, (uint64_t)((tok0 + r) * E + 32 * w + 8 * g + 4 * hf), d.drop_p, keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int e = 4 * g + i;
            pre[e] = xa[i] + (o[e] + ba[i]) * keep[i];
            sum1 += pre[e];
        }
    }
    sum1 += __shfl_xor(sum1, 32, 64);
    if (hf == 0) red[0][w][r] = sum1;
    __syncthreads();
    const float mean = ((red[0][0][r] + red[0][1][r]) + (red[0][2][r] + red[0][3][r])) * (1.0f / E);
    float s2 = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float dl = pre[e] - mean; s2 += dl * dl; }
    s2 += __shfl_xor(s2, 32, 64);
    if (hf == 0) red[1][w][r] = s2;
    __syncthreads();
    const float rstd = rsqrtf(((red[1][0][r] + red[1][1][r]) + (red[1][2][r] + red[1][3][r])) * (1.0f / E) + d.eps);
    if (r < S) {
        float* yr = l.y2 + (tok0 + r) * E + 32 * w + 4 * hf;
        float* pr = l.pre2 ? l.pre2 + (tok0 + r) * E + 32 * w + 4 * hf : nullptr;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float4 gv = gmv[g], bt = btv[g];
            float4 out;
            out.x = (pre[4 * g] - mean) * rstd * gv.x + bt.x; out.y = (pre[4 * g + 1] - mean) * rstd * gv.y + bt.y;
            out.z = (pre[4 * g + 2] - mean) * rstd * gv.z + bt.z; out.w = (pre[4 * g + 3] - mean) * rstd * gv.w + bt.w;
            *(float4*)(yr + 8 * g) = out;
            if (pr) *(float4*)(pr + 8 * g) = make_float4(pre[4 * g], pre[4 * g + 1], pre[4 * g + 2], pre[4 * g + 3]);
            if (stash) *(float4*)(part + r * E + 32 * w + 4 * hf + 8 * g) = out;   // the last layer's rows stay in LDS for the sequence mean
        }
        if (w == 0 && hf == 0 && l.mean2) { l.mean2[tok0 + r] = mean; l.rstd2[tok0 + r] = rstd; }
    }
}

template <bool X3>
__global__ __launch_bounds__(256) void txl_block_fwd_kernel(BlockK k) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const BlockP& d = k.d;
    const int tid = threadIdx.x;
    int b, q;
    if (!quad_ids(k, b, q)) return;
    pos_add_seq(d, b);
    __syncthreads();
    for (int li = 0; li < d.L; ++li) {
        const LayerP& l = d.layers[li];
        const TxlP p = attn_params(d, l);
        // (the sequence index is made opaque per layer: otherwise the compiler hoists every layer-invariant piece of the dropout hashes and
        // address products out of this loop — 60+ registers live across both stages, spilled)
        int bb = b;
        asm volatile("" : "+s"(bb));
        if (!(k.dbg & 1)) txl_attn_fwd_body<X3>(p, bb, lds);
        __syncthreads();
        ffn_fwd_seq<X3>(k, l, bb, q, li, lds, li + 1 == d.L);
        __syncthreads();
    }
    quad_finish(k, b);
    if (d.pooled && tid < E) {                                      // seq_mean_fwd_kernel's summation order, rows from LDS
        const float* xb = (const float*)lds + tid;
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int t = 0;
        for (; t + 3 < d.S; t += 4) {
            const float a = xb[(long)t * E], c = xb[(long)(t + 1) * E], e = xb[(long)(t + 2) * E], f = xb[(long)(t + 3) * E];
            s0 += a; s1 += c; s2 += e; s3 += f;
        }
        for (; t < d.S; ++t) s0 += xb[(long)t * E];
        d.pooled[(long)b * E + tid] = ((s0 + s1) + (s2 + s3)) / d.S;
    }
}

// ---------------------------------------------------------------------------------------------------------------- backward stages
// LayerNorm backward in the (lane <-> token r, registers <-> features 32w + arow(e, hf)) layout (the LayerNorm1 section of txl_attn_bwd_body):
// dyv -> dpre, and the sequence's {dgamma, dbeta} partials
HULC_DEVICE void ln_bwd_regs(const float (&dyv)[16], const float* pre, const float* meanp, const float* rstdp, const float* gamma, long tok, int b,
                             float (*lt)[2][32][33], float (*red)[4][32], float* ln_partial, float (&dpre)[16]) {
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const long off = tok * E + 32 * w + 4 * hf;
    const float mean = meanp[tok], rstd = rstdp[tok];
    float g[16], xh[16];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) {
        const float4 pv = *(const float4*)(pre + off + 8 * q4), gm = *(const float4*)(gamma + 32 * w + 4 * hf + 8 * q4);
        const float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gm.x, gm.y, gm.z, gm.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = 4 * q4 + q;
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
#pragma unroll
    for (int e = 0; e < 16; ++e) dpre[e] = rstd * (g[e] - s1 - xh[e] * s2);
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < 32; ++i) acc += lt[w][hf][r][i];
    ln_partial[((long)b * 2 + hf) * E + 32 * w + r] = acc;
}

// LayerNorm2 backward + feed-forward backward of sequence b: dyv = gradient of y2 (registers) -> dy1 (memory), h / dh / df (bf16, memory)
HULC_DEVICE void ffn_bwd_seq(const BlockK& k, const LayerP& l, int b, int q, int li, const float (&dyv)[16], float (&dy1)[16], char* lds) {
    const BlockP& d = k.d;
    // LDS: [dfs 8 KB | lt 33 KB] during the LayerNorm part and the fragment exchange, then the 64 KB of partial tiles over both
    uint4* dfs = (uint4*)lds;
    float (*lt)[2][32][33] = (float (*)[2][32][33])(lds + 8 * 2 * 32 * 16);
    float* part = (float*)lds;
    float (*red)[4][32] = (float (*)[4][32])(lds + PART_BYTES);
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    const int S = d.S, FF = d.FF;
    const long tok0 = (long)b * S;
    const bool live = r < S;
    const long tok = tok0 + (live ? r : 0);
    const unsigned long long sd = d.seed_dev ? d.seed_dev[0] : 0ull;
    const unsigned long long seed_ffn = l.seed_ffn ^ sd, seed_ln = l.seed_ln2 ^ sd;
    const uint16_t* W1 = (const uint16_t*)l.W1;
    const uint16_t* W1T = (const uint16_t*)l.W1T;
    const uint16_t* W2T = (const uint16_t*)l.W2T;

    float dpre[16], df[16];
    ln_bwd_regs(dyv, l.pre2, l.mean2, l.rstd2, l.g2, tok, b, lt, red, l.lnp2, dpre);
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        float keep[4] = {1.f, 1.f, 1.f, 1.f};
        if (d.drop_p > 0.f) dropout_scale4(seed_ln, (uint64_t)((tok0 + r) * E + 32 * w + 8 * g4 + 4 * hf), d.drop_p, keep);
#pragma unroll
        for (int i = 0; i < 4; ++i) df[4 * g4 + i] = dpre[4 * g4 + i] * keep[i];
    }
    { Frag f; f.b = pack8f(df); dfs[((2 * w) * 2 + hf) * 32 + r] = f.u; f.b = pack8f(df + 8); dfs[((2 * w + 1) * 2 + hf) * 32 + r] = f.u; }
    if (live) {                                                     // row-major bf16: the left operand of dW2 = df^T h (its row sums = db2)
        uint16_t* dst = (uint16_t*)l.df + (tok0 + r) * E + 32 * w + 4 * hf;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4)
            *(uint2*)(dst + 8 * q4) = make_uint2(pack_bf16x2(df[4 * q4], df[4 * q4 + 1]), pack_bf16x2(df[4 * q4 + 2], df[4 * q4 + 3]));
    }
    bf16x8_t xf[8];
    load_x_frags(xf, l.y1, tok0, r, hf, S);
    __syncthreads();
    bf16x8_t dff[8];                                                // df^T fragments: k = out feature (register order of the producing tiles), n = token
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) { Frag f; f.u = dfs[(kk * 2 + hf) * 32 + r]; dff[kk] = f.b; }
    f32x16_t acc[4];
#pragma unroll
    for (int ot = 0; ot < 4; ++ot) acc[ot] = zero16();
    uint16_t* hrow = (uint16_t*)l.h + (tok0 + r) * (long)FF + 4 * hf;
    uint16_t* dhrow = (uint16_t*)l.dh + (tok0 + r) * (long)FF + 4 * hf;
    const int NS = (k.dbg & 2) ? 0 : FF / 128 / k.Q;
    for (int s = q * NS; s < (q + 1) * NS; ++s) {                   // this member's hidden slices
        const int j0 = s * 128 + 32 * w;
        bf16x8_t w1f[8], w2t[8], w1t[8];
        if (l.W1p && l.W2Tp && l.W1Tp) {                            // fragment-packed copies: 1 KB contiguous per load instruction
            const long u = ((long)(j0 / 32) * 8 * 64 + lane) * 8;
#pragma unroll
            for (int q8 = 0; q8 < 8; ++q8) {
                w1f[q8] = ldg16((const uint16_t*)l.W1p + u + q8 * 512); w2t[q8] = ldg16((const uint16_t*)l.W2Tp + u + q8 * 512);
                w1t[q8] = ldg16((const uint16_t*)l.W1Tp + u + q8 * 512);
            }
        } else {
            const uint16_t* a = W1 + (long)(j0 + r) * E + hf * 8;
            const uint16_t* c = W2T + (long)(j0 + r) * E + 4 * hf;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) { w1f[ks] = ldg16(a + ks * 16); w2t[ks] = ldg_split(c + 32 * (ks >> 1) + 16 * (ks & 1)); }
#pragma unroll
            for (int ot = 0; ot < 4; ++ot)
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) w1t[ot * 2 + kk] = ldg_split(W1T + (long)(ot * 32 + r) * FF + j0 + 16 * kk + 4 * hf);
        }
        f32x16_t zT = zero16(), dT = zero16();                      // [hidden j0 + arow][token r]
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) { zT = MFMA(w1f[ks], xf[ks], zT); dT = MFMA(w2t[ks], dff[ks], dT); }
        add_row_vec(zT, l.b1 + j0, hf, 1.f);
        float hv[16], dv[16];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            float keep[4] = {1.f, 1.f, 1.f, 1.f};
            if (d.drop_p > 0.f) dropout_scale4(seed_ffn, (uint64_t)(tok0 + r) * (uint64_t)FF + (uint64_t)(j0 + 8 * g4 + 4 * hf), d.drop_p, keep);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * g4 + i;
                const bool on = zT[e] > 0.f && live;
                hv[e] = on ? zT[e] * keep[i] : 0.f;
                dv[e] = on ? dT[e] * keep[i] : 0.f;
            }
        }
        if (live) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                *(uint2*)(hrow + j0 + 8 * g4) = make_uint2(pack_bf16x2(hv[4 * g4], hv[4 * g4 + 1]), pack_bf16x2(hv[4 * g4 + 2], hv[4 * g4 + 3]));
                *(uint2*)(dhrow + j0 + 8 * g4) = make_uint2(pack_bf16x2(dv[4 * g4], dv[4 * g4 + 1]), pack_bf16x2(dv[4 * g4 + 2], dv[4 * g4 + 3]));
            }
        }
        const bf16x8_t d0 = pack8f(dv), d1 = pack8f(dv + 8);
#pragma unroll
        for (int ot = 0; ot < 4; ++ot) {
            acc[ot] = MFMA(w1t[ot * 2], d0, acc[ot]);
            acc[ot] = MFMA(w1t[ot * 2 + 1], d1, acc[ot]);
        }
    }
    __syncthreads();                                                // dfs / lt are dead: the partial tiles take their place
    f32x16_t ax;
    exchange_tiles(acc, part, w, lane, ax);
    quad_exchange(k, ax, b, q, li, d.L - li);
#pragma unroll
    for (int e = 0; e < 16; ++e) dy1[e] = live ? dpre[e] + ax[e] : 0.f;
}

__global__ __launch_bounds__(256) void txl_block_bwd_kernel(BlockK k) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const BlockP& d = k.d;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, r = lane & 31, hf = lane >> 5;
    int b, q;
    if (!quad_ids(k, b, q)) return;
    const int S = d.S;
    const long tok0 = (long)b * S;
    const bool live = r < S;
    const long off = (tok0 + (live ? r : 0)) * E + 32 * w + 4 * hf;
    float dyv[16];                                                  // gradient of the current layer's output: registers from stage to stage
    {                                                               // mean over the sequence: every token receives dpooled / S
        const float* src = d.dpooled + (long)b * E + 32 * w + 4 * hf;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const float4 a = *(const float4*)(src + 8 * q4);
            dyv[4 * q4] = a.x / S; dyv[4 * q4 + 1] = a.y / S; dyv[4 * q4 + 2] = a.z / S; dyv[4 * q4 + 3] = a.w / S;
        }
        if (!live) {
#pragma unroll
            for (int e = 0; e < 16; ++e) dyv[e] = 0.f;
        }
    }
    for (int li = d.L - 1; li >= 0; --li) {
        const LayerP& l = d.layers[li];
        int bb = b;                                                 // (opaque per layer, see the forward kernel)
        asm volatile("" : "+s"(bb));
        float dy1[16];
        ffn_bwd_seq(k, l, bb, q, li, dyv, dy1, lds);
        __syncthreads();
        const TxlP p = attn_params(d, l);
        asm volatile("" : "+s"(bb));
        if (!(k.dbg & 1)) txl_attn_bwd_body<true>(p, bb, lds, dy1, dyv);
        __syncthreads();
    }
    quad_finish(k, b);
    if (live) {                                                     // dropout of the position-embedded input (dropout_bwd_kernel's stream)
        const unsigned long long seed = d.seed_pos ^ (d.seed_dev ? d.seed_dev[0] : 0ull);
        float* dst = d.demb + off;
        float keep[16];
        if (d.drop_p > 0.f) dropout_scale_acc16(seed, (uint64_t)((tok0 + r) * E + 32 * w), hf, d.drop_p, keep);
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            float4 a = make_float4(dyv[4 * q4], dyv[4 * q4 + 1], dyv[4 * q4 + 2], dyv[4 * q4 + 3]);
            if (d.drop_p > 0.f) { a.x *= keep[4 * q4]; a.y *= keep[4 * q4 + 1]; a.z *= keep[4 * q4 + 2]; a.w *= keep[4 * q4 + 3]; }
            *(float4*)(dst + 8 * q4) = a;
        }
    }
}

int block_check(const hulc_txl_block_desc* d, bool bwd, const char* who) {
    if (!d || !d->emb || !d->pos || !d->pos_ids) return hulc_fail(-1, who);
    if (d->E != E || d->H != NH || d->S < 1 || d->S > SMAX || d->B < 1 || d->L < 1 || d->L > HULC_TXL_MAX_LAYERS || d->FF < 128 || d->FF % 128)
        return hulc_fail(-2, who);
    for (int i = 0; i < d->L; ++i) {
        const hulc_txl_block_layer& l = d->layers[i];
        if (!l.Wqkv || !l.Wo || !l.W1 || !l.W2 || !l.bqkv || !l.bo || !l.b1 || !l.b2 || !l.g1 || !l.be1 || !l.g2 || !l.be2 || !l.x || !l.y1 || !l.y2)
            return hulc_fail(-1, who);
        if ((l.pre1 != nullptr) != (l.mean1 != nullptr) || (l.mean1 != nullptr) != (l.rstd1 != nullptr) ||
            (l.pre2 != nullptr) != (l.mean2 != nullptr) || (l.mean2 != nullptr) != (l.rstd2 != nullptr))
            return hulc_fail(-3, "hulc_txl_block: pre / mean / rstd are kept together or not at all");
        if (i + 1 < d->L && d->layers[i + 1].x != l.y2) return hulc_fail(-3, "hulc_txl_block: layer l+1's x is layer l's y2");
        if (bwd && (!l.WqkvT || !l.WoT || !l.W1T || !l.W2T || !l.pre1 || !l.mean1 || !l.rstd1 || !l.ctx || !l.pre2 || !l.mean2 || !l.rstd2 || !l.d_o ||
                    !l.dqkv || !l.df || !l.h || !l.dh || !l.lnp1 || !l.lnp2))
            return hulc_fail(-1, who);
    }
    return 0;
}

int block_lds(const void* fn) {
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, BLOCK_LDS) != hipSuccess)
        return hulc_fail(-8, "hulc_txl_block: could not raise the dynamic LDS limit");
    return 0;
}

int device_cus() {
    static int n = -1;
    if (n < 0) {
        int dev = 0;
        hipDeviceProp_t pr;
        n = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&pr, dev) == hipSuccess) ? pr.multiProcessorCount : 0;
    }
    return n;
}

// workgroups per sequence: as many as divide the hidden slices while the whole grid is co-resident (one workgroup per CU)
int block_share(const hulc_txl_block_desc* d) {
    static const int forced = getenv("HULC_TXL_SHARE") ? atoi(getenv("HULC_TXL_SHARE")) : 0;
    if (!d->ws || d->exclusive == 0) return 1;
    const int ns = d->FF / 128, groups = (d->B + 7) / 8 * 8;
    for (int q = 4; q > 1; q >>= 1)
        if ((forced == 0 || q <= forced) && ns % q == 0 && groups * q <= device_cus() / hulc_coop_share() && d->B <= SYNC_MAX_SEQ) return q;
    return 1;
}

BlockK block_kernel_params(const hulc_txl_block_desc* d, int Q) {
    BlockK k = {};
    static const int dbg = getenv("HULC_TXL_DBG") ? atoi(getenv("HULC_TXL_DBG")) : 0;
    k.d = *d; k.Q = Q; k.err_sticky = d->err_sticky; k.dbg = dbg;
    k.sync = (unsigned*)d->ws;
    k.xpart = (float*)((char*)d->ws + SYNC_BYTES);
    return k;
}

static_assert(BLOCK_LDS >= TXL_BWD_LDS && BLOCK_LDS >= TXL_FWD_LDS_X3, "the attention stages fit the block's LDS");
static_assert(8 * 2 * 32 * 16 + LT_BYTES <= PART_BYTES, "fragment exchange + LayerNorm transposes fit under the partial tiles");

}  // namespace

// see include/hulc2_amd.h
extern "C" long hulc_txl_block_workspace(int B, int L) {
    return SYNC_BYTES + (long)L * B * 4 * (4 * 16 * 64) * 4;
}

extern "C" int hulc_txl_block_fwd(const hulc_txl_block_desc* d, void* stream) {
    if (int rc = block_check(d, false, "hulc_txl_block_fwd: needs d_model 128, 8 heads, 1 <= S <= 32, 1 <= L <= 4, FF a multiple of 128 and non-null operands")) return rc;
    static bool attr = false;
    if (!attr) {
        if (int rc = block_lds((const void*)txl_block_fwd_kernel<false>)) return rc;
        if (int rc = block_lds((const void*)txl_block_fwd_kernel<true>)) return rc;
        attr = true;
    }
    int nlo = 0;
    for (int i = 0; i < d->L; ++i) {
        const hulc_txl_block_layer& l = d->layers[i];
        const int have = (l.Wqkv_lo != nullptr) + (l.Wo_lo != nullptr) + (l.W1p_lo != nullptr) + (l.W2p_lo != nullptr);
        if (have != 0 && (have != 4 || !l.W1p || !l.W2p))
            return hulc_fail(-3, "hulc_txl_block_fwd: the split-operand forward needs all four remainder arrays and the packed W1p / W2p of a layer");
        nlo += have == 4;
    }
    if (nlo != 0 && nlo != d->L) return hulc_fail(-3, "hulc_txl_block_fwd: remainder arrays on every layer or on none");
    const int Q = block_share(d);
    const unsigned grid = Q == 1 ? (unsigned)d->B : (unsigned)((d->B + 7) / 8 * 8 * Q);
    if (nlo) txl_block_fwd_kernel<true><<<grid, 256, BLOCK_LDS, (hipStream_t)stream>>>(block_kernel_params(d, Q));
    else txl_block_fwd_kernel<false><<<grid, 256, BLOCK_LDS, (hipStream_t)stream>>>(block_kernel_params(d, Q));
    return hulc_check_launch("hulc_txl_block_fwd");
}

extern "C" int hulc_txl_block_bwd(const hulc_txl_block_desc* d, void* stream) {
    if (int rc = block_check(d, true, "hulc_txl_block_bwd: needs d_model 128, 8 heads, 1 <= S <= 32, 1 <= L <= 4, FF a multiple of 128 and non-null operands")) return rc;
    if (!d->dpooled || !d->demb) return hulc_fail(-1, "hulc_txl_block_bwd: null pointer");
    static bool attr = false;
    if (!attr) { if (int rc = block_lds((const void*)txl_block_bwd_kernel)) return rc; attr = true; }
    const int Q = block_share(d);
    const unsigned grid = Q == 1 ? (unsigned)d->B : (unsigned)((d->B + 7) / 8 * 8 * Q);
    txl_block_bwd_kernel<<<grid, 256, BLOCK_LDS, (hipStream_t)stream>>>(block_kernel_params(d, Q));
    return hulc_check_launch("hulc_txl_block_bwd");
}
