// lang_encoder.hip — the pieces of the MiniLM-L3 sentence encoder around its dense layers.
//
// reference behaviour: SentenceTransformer("paraphrase-MiniLM-L3-v2") as hulc2/affordance/models/language_encoders/
// sbert_lang_encoder.py:13-71 calls it (frozen, eval): transformers' BertEmbeddings / BertSelfAttention / BertSelfOutput /
// BertIntermediate / BertOutput followed by sentence_transformers' mean Pooling.  Those packages are un-vendored dependencies
// (requirements.txt:19): the arithmetic is restated from their published definition and checked against transformers' BertModel with
// seeded weights (oracle/gen_golden.py minilm).  Inference only, a few dozen sentences of a few dozen tokens per call: latency-bound,
// one wave per row / head, fp32.
#include "hulc_common.h"
#include "hulc_abi_internal.h"

namespace {

// LayerNorm of a row held as v[16] per lane (D <= 1024): y = (v - mean) * rstd * gamma + beta, biased variance as torch
HULC_DEVICE void ln_row_store(const float (&v)[16], int D, int lane, float eps, const float* gamma, const float* beta, float* y) {
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = lane + q * 64; if (c < D) s += v[q]; }
    const float mean = wave_sum(s) / D;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = lane + q * 64; if (c < D) { const float d = v[q] - mean; ss += d * d; } }
    const float rstd = rsqrtf(wave_sum(ss) / D + eps);
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = lane + q * 64; if (c < D) y[c] = (v[q] - mean) * rstd * gamma[c] + beta[c]; }
}

__global__ __launch_bounds__(256) void embed_ln_kernel(const long* __restrict__ ids, const float* __restrict__ word, const float* __restrict__ pos,
                                                       const float* __restrict__ type0, const float* __restrict__ gamma,
                                                       const float* __restrict__ beta, float eps, int T, int S, int D, float* __restrict__ out) {
    const int lane = threadIdx.x & 63, t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= T) return;
    const float* w = word + ids[t] * D;
    const float* pp = pos + (long)(t % S) * D;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = lane + q * 64; v[q] = c < D ? w[c] + type0[c] + pp[c] : 0.f; }   // BertEmbeddings: (word + type) + position
    ln_row_store(v, D, lane, eps, gamma, beta, out + (long)t * D);
}

__global__ __launch_bounds__(256) void ln_wide_kernel(const float* __restrict__ x, const float* __restrict__ add, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, float eps, int R, int D, float* __restrict__ y) {
    const int lane = threadIdx.x & 63, r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) { const int c = lane + q * 64; v[q] = c < D ? x[(long)r * D + c] + (add ? add[(long)r * D + c] : 0.f) : 0.f; }
    ln_row_store(v, D, lane, eps, gamma, beta, y + (long)r * D);
}

// one workgroup per (sentence, head): K and V of the head in LDS, thread = query token
__global__ __launch_bounds__(128) void mha_masked_kernel(const float* __restrict__ qkv, const int* __restrict__ mask, int S, int nhead, int hd,
                                                         float* __restrict__ out) {
    extern __shared__ float sm[];                           // K [S][hd + 1] | V [S][hd + 1] | additive mask [S]
    const int b = blockIdx.x / nhead, hI = blockIdx.x % nhead, D = nhead * hd, ldq = 3 * D, hp = hd + 1;
    float* Ks = sm; float* Vs = sm + S * hp; float* am = Vs + S * hp;
    for (int i = threadIdx.x; i < S * hd; i += blockDim.x) {
        const int s = i / hd, c = i % hd;
        const float* row = qkv + (long)(b * S + s) * ldq + hI * hd + c;
        Ks[s * hp + c] = row[D]; Vs[s * hp + c] = row[2 * D];
    }
    for (int s = threadIdx.x; s < S; s += blockDim.x) am[s] = mask[b * S + s] ? 0.f : -3.4028234663852886e38f;   // (1 - mask) * finfo.min
    __syncthreads();
    const int s = threadIdx.x;
    if (s >= S) return;
    float q[64];
    const float* qrow = qkv + (long)(b * S + s) * ldq + hI * hd;
#pragma unroll 8
    for (int c = 0; c < hd; ++c) q[c] = qrow[c];
    const float scale = rsqrtf((float)hd);
    float mx = -INFINITY;
    for (int k = 0; k < S; ++k) {
        float d = 0.f;
        for (int c = 0; c < hd; ++c) d += q[c] * Ks[k * hp + c];
        d = d * scale + am[k];                              // -inf if the sum overflows, exactly as the fp32 reference
        mx = fmaxf(mx, d);
    }
    float den = 0.f, acc[64];
    for (int c = 0; c < hd; ++c) acc[c] = 0.f;
    for (int k = 0; k < S; ++k) {
        float d = 0.f;
        for (int c = 0; c < hd; ++c) d += q[c] * Ks[k * hp + c];
        const float e = expf(d * scale + am[k] - mx);
        den += e;
        for (int c = 0; c < hd; ++c) acc[c] += e * Vs[k * hp + c];
    }
    float* o = out + (long)(b * S + s) * D + hI * hd;
    for (int c = 0; c < hd; ++c) o[c] = acc[c] / den;
}

__global__ void masked_mean_kernel(const float* __restrict__ x, const int* __restrict__ mask, int S, int D, float* __restrict__ out) {
    const int b = blockIdx.x;
    float cnt = 0.f;
    for (int s = 0; s < S; ++s) cnt += mask[b * S + s] ? 1.f : 0.f;
    cnt = fmaxf(cnt, 1e-9f);
    for (int c = threadIdx.x; c < D; c += blockDim.x) {
        float a = 0.f;
        for (int s = 0; s < S; ++s) if (mask[b * S + s]) a += x[((long)b * S + s) * D + c];
        out[(long)b * D + c] = a / cnt;
    }
}

}  // namespace

extern "C" int hulc_embed_ln_fwd(const long* ids, const float* word, const float* pos, const float* type0, const float* gamma, const float* beta,
                                 float eps, int T, int S, int D, float* out, void* stream) {
    if (!ids || !word || !pos || !type0 || !gamma || !beta || !out) return hulc_fail(-1, "hulc_embed_ln_fwd: null pointer");
    if (T <= 0 || S <= 0 || D <= 0 || D > 1024) return hulc_fail(-2, "hulc_embed_ln_fwd: needs T, S > 0 and 0 < D <= 1024");
    embed_ln_kernel<<<(T + 3) / 4, 256, 0, (hipStream_t)stream>>>(ids, word, pos, type0, gamma, beta, eps, T, S, D, out);
    return hulc_check_launch("hulc_embed_ln_fwd");
}

extern "C" int hulc_ln_wide_fwd(const float* x, const float* add, const float* gamma, const float* beta, float eps, int R, int D, float* y, void* stream) {
    if (!x || !gamma || !beta || !y) return hulc_fail(-1, "hulc_ln_wide_fwd: null pointer");
    if (R <= 0 || D <= 0 || D > 1024) return hulc_fail(-2, "hulc_ln_wide_fwd: needs R > 0 and 0 < D <= 1024");
    ln_wide_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, add, gamma, beta, eps, R, D, y);
    return hulc_check_launch("hulc_ln_wide_fwd");
}

extern "C" int hulc_mha_masked_fwd(const float* qkv, const int* mask, int B, int S, int nhead, int hd, float* out, void* stream) {
    if (!qkv || !mask || !out) return hulc_fail(-1, "hulc_mha_masked_fwd: null pointer");
    if (B <= 0 || S <= 0 || S > 128 || nhead <= 0 || hd <= 0 || hd > 64) return hulc_fail(-2, "hulc_mha_masked_fwd: needs S <= 128 and head dim <= 64");
    const size_t lds = (size_t)(2 * S * (hd + 1) + S) * sizeof(float);
    mha_masked_kernel<<<B * nhead, 128, lds, (hipStream_t)stream>>>(qkv, mask, S, nhead, hd, out);
    return hulc_check_launch("hulc_mha_masked_fwd");
}

extern "C" int hulc_masked_mean_fwd(const float* x, const int* mask, int B, int S, int D, float* out, void* stream) {
    if (!x || !mask || !out) return hulc_fail(-1, "hulc_masked_mean_fwd: null pointer");
    if (B <= 0 || S <= 0 || D <= 0) return hulc_fail(-2, "hulc_masked_mean_fwd: bad shape");
    masked_mean_kernel<<<B, 128, 0, (hipStream_t)stream>>>(x, mask, S, D, out);
    return hulc_check_launch("hulc_masked_mean_fwd");
}
