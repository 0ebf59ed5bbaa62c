// losses.hip — the scalar heads of the HULC++ training step, forward and analytic backward.
//
//   discretised logistic mixture NLL + gripper cross-entropy
//       reference: LogisticDecoderRNN._loss/_logistic_loss, hulc2/models/decoders/logistic_decoder_rnn.py:133-152,181-228
//   KL-balanced categorical KL between plan recognition (posterior) and plan proposal (prior)
//       reference: Hulc2.compute_kl_loss, hulc2/models/hulc2.py:444-466 (+ torch.distributions categorical KL)
//   straight-through one-hot sample of the 32x32 latent plan
//       reference: hulc2/utils/distributions.py:23-27 + hulc2.py:235-237
//   CLIP-style symmetric contrastive loss on projected features
//       reference: Hulc2.clip_auxiliary_loss, hulc2.py:472-508
//   world -> tcp frame change of the relative actions
//       reference: hulc2/models/decoders/utils/gripper_control.py:16-36 (pytorch3d XYZ euler maths)
//
// These are tiny, reduction-shaped problems (<= 2048 tokens): each is one workgroup (deterministic
// summation order) or one thread per token; fp32 VALU throughout, libm-accurate exp/log.
#include "hulc_common.h"
#include "hulc_abi_internal.h"

namespace {

HULC_DEVICE float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); }   // torch threshold = 20
HULC_DEVICE float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }

// deterministic block-wide sum (blockDim.x <= 1024); result valid on every thread
HULC_DEVICE float block_sum(float v, float* sh) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    float t = 0.f;
    for (int w = 0; w < nw; ++w) t += sh[w];
    return t;
}

struct MixP {
    const float* y; long ld;        // [T][>=182] head outputs: logit_probs | means | log_scales | gripper(2)
    const float* act;               // [T][A+1]
    const float* amin; const float* amax;   // [A]
    int T, A, NM, num_classes;
    float log_scale_min, gripper_alpha;
    int tm_B, nseg;                 // tm_B > 0: rows are time-major (row = step * tm_B + batch row) and a segment is a block of tm_B / nseg batch rows
};

// physical row of token j (0 .. seg_tokens) of segment seg
HULC_DEVICE int mix_row(const MixP& p, int seg, int j, int seg_tokens) {
    if (p.tm_B == 0) return seg * seg_tokens + j;
    const int bs = p.tm_B / p.nseg;
    return (j / bs) * p.tm_B + seg * bs + j % bs;
}
HULC_DEVICE int mix_seg_of_row(const MixP& p, int t, int seg_tokens) { return p.tm_B == 0 ? t / seg_tokens : (t % p.tm_B) / (p.tm_B / p.nseg); }

// per (token, action dim): NLL and, when G != nullptr, gradients w.r.t. the 3*NM head outputs
// per (token, action dim): NLL and, when GRAD, gradients w.r.t. the 3*NM head outputs — one LANE per mixture component: an item is a
// group of 16 lanes (NM <= 16), the two log-sum-exps are butterfly reductions inside the group.  An item-per-thread form is one long dependent chain of ~40 transcendentals per thread and
// only (T * 7) / 64 = 224 waves for the benchmark's 2048 tokens: 23 us forward, 30 us backward; spread over 16x the lanes it is latency-hidden.
HULC_DEVICE float grp16_max(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
HULC_DEVICE float grp16_sum(float v) {
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
template <bool GRAD>
HULC_DEVICE float mix_nll_lanes(const MixP& p, int t, int d, int i, float gscale, float* dy_row) {
    const float* row = p.y + (long)t * p.ld;
    const int NM = p.NM;
    const bool on = i < NM;
    const int ii = on ? i : 0;
    const float a = p.act[(long)t * (p.A + 1) + d];
    const float lo = p.amin[d], hi = p.amax[d];
    const float half = (hi - lo) * 0.5f / (float)(p.num_classes - 1);
    const float logc = logf((float)(p.num_classes - 1) * 0.5f);
    const float logit = row[d * NM + ii];
    const float lmax = grp16_max(on ? logit : -INFINITY);
    const float llse = lmax + logf(grp16_sum(on ? expf(logit - lmax) : 0.f));
    const float mu = row[p.A * NM + d * NM + ii];
    const float raw = row[2 * p.A * NM + d * NM + ii];
    const float ls = fmaxf(raw, p.log_scale_min);
    const float inv = expf(-ls), c = a - mu;
    const float plus = inv * (c + half), minn = inv * (c - half), mid = inv * c;
    float v, gmu, gls;
    if (a < lo + 1e-3f) {
        v = plus - softplus_t(plus);
        const float k = 1.f - sigmoid_f(plus);
        gmu = -inv * k; gls = -plus * k;
    } else if (a > hi - 1e-3f) {
        v = -softplus_t(minn);
        const float k = -sigmoid_f(minn);
        gmu = -inv * k; gls = -minn * k;
    } else {
        const float sp = sigmoid_f(plus), sm = sigmoid_f(minn), delta = sp - sm;
        if (delta > 1e-5f) {
            v = logf(fmaxf(delta, 1e-12f));
            const float dp = sp * (1.f - sp), dm = sm * (1.f - sm);
            gmu = (-inv * dp + inv * dm) / delta;
            gls = (-plus * dp + minn * dm) / delta;
        } else {
            v = mid - ls - 2.f * softplus_t(mid) - logc;
            const float k = 1.f - 2.f * sigmoid_f(mid);
            gmu = -inv * k; gls = -mid * k - 1.f;
        }
    }
    if (raw < p.log_scale_min) gls = 0.f;
    const float lp = v + (logit - llse);
    const float m = grp16_max(on ? lp : -INFINITY);
    const float lse = m + logf(grp16_sum(on ? expf(lp - m) : 0.f));
    if (GRAD && on) {
        const float w = expf(lp - lse), pi = expf(logit - llse);
        dy_row[d * NM + i] = -gscale * (w - pi);
        dy_row[p.A * NM + d * NM + i] = -gscale * w * gmu;
        dy_row[2 * p.A * NM + d * NM + i] = -gscale * w * gls;
    }
    return -lse;
}

// Tokens are split into `nseg` equal segments (one per modality when both are batched through the decoder); every
// segment gets its own mean.  Pass 1: one (token, dim) item per thread, per-workgroup (nll, ce) partials; a workgroup
// never straddles a segment (items per segment are padded to the workgroup size).  Pass 2: one workgroup per segment
// adds the partials in a fixed order.  out (3, nseg) planar = totals | nll means | ce means.
__global__ __launch_bounds__(256) void mix_loss_partial_kernel(MixP p, int seg_tokens, int blocks_per_seg, float* __restrict__ partial) {
    __shared__ float sh[16];
    const int seg = blockIdx.x / blocks_per_seg, bl = blockIdx.x % blocks_per_seg;
    const int w = bl * 16 + (threadIdx.x >> 4), i = threadIdx.x & 15;     // item inside the segment (16 per workgroup), mixture lane
    float nll = 0.f, ce = 0.f;
    if (w < seg_tokens * (p.A + 1)) {
        const int t = mix_row(p, seg, w / (p.A + 1), seg_tokens), d = w % (p.A + 1);
        if (d < p.A) {
            const float v = mix_nll_lanes<false>(p, t, d, i, 0.f, nullptr);
            nll = i == 0 ? v : 0.f;
        } else if (i == 0) {
            const float* g = p.y + (long)t * p.ld + 3 * p.A * p.NM;
            const float a = p.act[(long)t * (p.A + 1) + p.A];
            const int lbl = (a == -1.f) ? 0 : (int)a;
            const float mx = fmaxf(g[0], g[1]);
            ce = mx + logf(expf(g[0] - mx) + expf(g[1] - mx)) - g[lbl];
        }
    }
    nll = block_sum(nll, sh);
    ce = block_sum(ce, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = nll; partial[2 * blockIdx.x + 1] = ce; }
}

__global__ __launch_bounds__(64) void mix_loss_final_kernel(const float* __restrict__ partial, int blocks_per_seg, int seg_tokens,
                                                            float gripper_alpha, float* __restrict__ out) {
    const int seg = blockIdx.x;
    float nll = 0.f, ce = 0.f;                              // lane-strided partial sums, then one butterfly: a fixed order
    for (int b = threadIdx.x; b < blocks_per_seg; b += 64) { nll += partial[2 * (seg * blocks_per_seg + b)]; ce += partial[2 * (seg * blocks_per_seg + b) + 1]; }
    nll = wave_sum(nll); ce = wave_sum(ce);
    if (threadIdx.x != 0) return;
    nll /= seg_tokens; ce /= seg_tokens;
    // planar: out[0 .. nseg) = totals (what the step's loss tail reads, contiguous), then the nll means, then the ce means
    out[seg] = nll + gripper_alpha * ce; out[gridDim.x + seg] = nll; out[2 * gridDim.x + seg] = ce;
}

// gout[seg] scales segment seg's tokens
__global__ __launch_bounds__(256) void mix_loss_bwd_kernel(MixP p, int seg_tokens, const float* __restrict__ gout, float* __restrict__ dy, long ld_dy) {
    const int w = blockIdx.x * 16 + (threadIdx.x >> 4), i = threadIdx.x & 15;
    if (w >= p.T * (p.A + 1)) return;                      // whole 16-lane groups leave together
    const int t = w / (p.A + 1), d = w % (p.A + 1);
    const float g = gout[mix_seg_of_row(p, t, seg_tokens)] / seg_tokens;
    float* drow = dy + (long)t * ld_dy;
    if (d < p.A) mix_nll_lanes<true>(p, t, d, i, g, drow);
    else if (i == 0) {
        for (int c = 3 * p.A * p.NM + 2; c < ld_dy; ++c) drow[c] = 0.f;      // pad columns of the fused head output carry no gradient
        const float* gl = p.y + (long)t * p.ld + 3 * p.A * p.NM;
        const float a = p.act[(long)t * (p.A + 1) + p.A];
        const int lbl = (a == -1.f) ? 0 : (int)a;
        const float mx = fmaxf(gl[0], gl[1]);
        const float e0 = expf(gl[0] - mx), e1 = expf(gl[1] - mx), inv = 1.f / (e0 + e1);
        drow[3 * p.A * p.NM + 0] = g * p.gripper_alpha * (e0 * inv - (lbl == 0 ? 1.f : 0.f));
        drow[3 * p.A * p.NM + 1] = g * p.gripper_alpha * (e1 * inv - (lbl == 1 ? 1.f : 0.f));
    }
}

// ------------------------------------------------------------------------------------------------
// categorical latent plan: groups of CLS logits; 32-lane sub-wave per group
// ------------------------------------------------------------------------------------------------
HULC_DEVICE float sub32_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
HULC_DEVICE float sub32_max(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// out[0] = beta * (mix*KL + (1-mix)*KL) = beta*KL_mean (value), CLS == 32: one 32-lane sub-wave per (row, category) group,
// then a single workgroup sums the B*G group values in a fixed order (deterministic).
__global__ __launch_bounds__(256) void cat_kl_group_kernel(const float* __restrict__ pp, const float* __restrict__ pr, int NG,
                                                           float* __restrict__ kl_group) {
    const int lane = threadIdx.x & 31;
    const int g = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (g >= NG) return;
    const float a = pr[(long)g * 32 + lane], b = pp[(long)g * 32 + lane];
    const float ma = sub32_max(a), mb = sub32_max(b);
    const float lp = a - (ma + logf(sub32_sum(expf(a - ma))));
    const float lq = b - (mb + logf(sub32_sum(expf(b - mb))));
    const float kl = sub32_sum(expf(lp) * (lp - lq));
    if (lane == 0) kl_group[g] = kl;
}
// one workgroup per segment (a modality of the batch): out[seg] = beta * mean over the segment's rows
__global__ __launch_bounds__(1024) void cat_kl_sum_kernel(const float* __restrict__ kl_group, int NG_seg, int B_seg, float beta, float* __restrict__ out) {
    __shared__ float sh[16];
    const float* kg = kl_group + (long)blockIdx.x * NG_seg;
    float acc = 0.f;
    for (int g = threadIdx.x; g < NG_seg; g += blockDim.x) acc += kg[g];
    acc = block_sum(acc, sh);
    if (threadIdx.x == 0) out[blockIdx.x] = beta * acc / B_seg;
}

__global__ __launch_bounds__(256) void cat_kl_bwd_kernel(const float* __restrict__ pp, const float* __restrict__ pr,
                                                         const float* __restrict__ kl_group, int B, int G, float beta, float mix,
                                                         const float* __restrict__ gout, int nseg, float* __restrict__ dpp, float* __restrict__ dpr) {
    const int lane = threadIdx.x & 31;
    const int g = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (g >= B * G) return;
    const int Bs = B / nseg;                                      // rows per segment; gout[seg] is that segment's upstream gradient
    gout += (g / G) / Bs;
    B = Bs;
    const float a = pr[(long)g * 32 + lane], b = pp[(long)g * 32 + lane];
    const float ma = sub32_max(a), mb = sub32_max(b);
    const float lp = a - (ma + logf(sub32_sum(expf(a - ma))));
    const float lq = b - (mb + logf(sub32_sum(expf(b - mb))));
    const float p = expf(lp), q = expf(lq);
    const float s = gout[0] * beta / B;
    dpp[(long)g * 32 + lane] = s * mix * (q - p);
    dpr[(long)g * 32 + lane] = s * (1.f - mix) * p * (lp - lq - kl_group[g]);
}

// one-hot sample per group; idx_in (optional) injects the class indices (parity tests)
__global__ __launch_bounds__(256) void plan_sample_kernel(const float* __restrict__ logits, const long* __restrict__ idx_in,
                                                          unsigned long long seed, const unsigned long long* __restrict__ seed_dev, int NG,
                                                          long* __restrict__ idx_out, float* __restrict__ plan) {
    const int lane = threadIdx.x & 31;
    if (seed_dev) seed ^= seed_dev[0];
    const int g = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (g >= NG) return;
    int idx;
    if (idx_in) idx = (int)idx_in[g];
    else {
        const float a = logits[(long)g * 32 + lane];
        const float e = expf(a - sub32_max(a));
        float c = e;                                           // inclusive prefix sum over the 32 classes
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) { const float n = __shfl_up(c, o, 32); if (lane >= o) c += n; }
        const float total = __shfl(c, 31, 32);
        const float u = hulc_uniform01(seed, (uint64_t)g) * total;
        const unsigned long long ball = __ballot(c > u);
        const unsigned int mine = (unsigned int)(ball >> (threadIdx.x & 32));   // this 32-lane half of the wave
        idx = mine ? __ffs((int)mine) - 1 : 31;
    }
    if (lane == 0 && idx_out) idx_out[g] = idx;
    plan[(long)g * 32 + lane] = lane == idx ? 1.f : 0.f;
}

// straight-through estimator: d logits = softmax jacobian^T * d plan
__global__ __launch_bounds__(256) void plan_sample_bwd_kernel(const float* __restrict__ logits, const float* __restrict__ dplan, int NG,
                                                              float* __restrict__ dlogits, int accumulate) {
    const int lane = threadIdx.x & 31;
    const int g = blockIdx.x * (blockDim.x >> 5) + (threadIdx.x >> 5);
    if (g >= NG) return;
    const float a = logits[(long)g * 32 + lane];
    const float e = expf(a - sub32_max(a));
    const float p = e / sub32_sum(e);
    const float gd = dplan[(long)g * 32 + lane];
    const float dot = sub32_sum(p * gd);
    const float v = p * (gd - dot);
    const long i = (long)g * 32 + lane;
    dlogits[i] = accumulate ? dlogits[i] + v : v;
}

// ------------------------------------------------------------------------------------------------
// CLIP-style contrastive loss, single workgroup, M <= 128 rows of D = 32 features
// ------------------------------------------------------------------------------------------------
#define CLIP_MAXM 128
#define CLIP_D 32
template <bool GRAD>
__global__ __launch_bounds__(256) void clip_loss_kernel(const float* __restrict__ im, const float* __restrict__ tx,
                                                        const unsigned char* __restrict__ use, int row0, const float* __restrict__ logit_scale, int M,
                                                        float* __restrict__ out, const float* __restrict__ gout, float* __restrict__ dim_,
                                                        float* __restrict__ dtx, float* __restrict__ dscale) {
    // one workgroup; the operands are staged into LDS with coalesced loads and every later phase works from LDS (the earlier form looped
    // over global memory: 64 dependent loads per thread for the norms, M byte loads for the `use` count, and the projection re-read its own
    // global stores).  Rows are padded by one float because a lane walks a ROW index in the logit / gradient loops (stride 32 floats would
    // put every lane on one bank).  Measured: 18 / 25 us per launch against 19 / 27 before — the phases are short dependent chains behind
    // seven workgroup barriers on ONE CU, not memory time; left there (1 % of the step).
    constexpr int DP = CLIP_D + 1;
    const int LP = M + 1;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* n = sm;                       // [M][DP] image features, normalised in place
    float* t = n + M * DP;               // [M][DP]
    float* L = t + M * DP;               // [M][LP] logits, later dL
    float* ni = L + M * LP;              // [M] norms
    float* ti = ni + M;                  // [M]
    float* rl = ti + M;                  // [M] row lse
    float* cl = rl + M;                  // [M] col lse
    float* uf = cl + M;                  // [M] use flags as 0 / 1
    float* dn = uf + M;                  // [M][DP] gradients w.r.t. the normalised features (GRAD)
    float* dt = dn + M * DP;             // [M][DP]
    __shared__ float sh[16];
    const int tid = threadIdx.x;
    const float s = expf(logit_scale[0]);
    // rows below row0 never take part (another modality's rows): the kernel works on rows row0 .. only (its cost is quadratic in the rows);
    // the backward writes zeros for the rest
    if (GRAD) for (int i = tid; i < row0 * CLIP_D; i += blockDim.x) { dim_[i] = 0.f; dtx[i] = 0.f; }
    im += (long)row0 * CLIP_D; tx += (long)row0 * CLIP_D;
    if (GRAD) { dim_ += (long)row0 * CLIP_D; dtx += (long)row0 * CLIP_D; }
    for (int i = tid; i < M * CLIP_D; i += blockDim.x) { const int r = i / CLIP_D, d = i % CLIP_D; n[r * DP + d] = im[i]; t[r * DP + d] = tx[i]; }
    for (int r = tid; r < M; r += blockDim.x) uf[r] = use[r] ? 1.f : 0.f;
    __syncthreads();
    for (int r = tid; r < M; r += blockDim.x) {
        float a = 0.f, b = 0.f;
        for (int d = 0; d < CLIP_D; ++d) { a += n[r * DP + d] * n[r * DP + d]; b += t[r * DP + d] * t[r * DP + d]; }
        ni[r] = sqrtf(a); ti[r] = sqrtf(b);
    }
    __syncthreads();
    for (int i = tid; i < M * CLIP_D; i += blockDim.x) { const int r = i / CLIP_D, q = r * DP + i % CLIP_D; n[q] = n[q] / ni[r]; t[q] = t[q] / ti[r]; }
    __syncthreads();
    for (int i = tid; i < M * M; i += blockDim.x) {
        const int r = i / M, c = i % M;
        float a = 0.f;
        for (int d = 0; d < CLIP_D; ++d) a += n[r * DP + d] * t[c * DP + d];
        L[r * LP + c] = s * a;
    }
    __syncthreads();
    float cnt = 0.f;
    for (int r = 0; r < M; ++r) cnt += uf[r];
    // row and column log-sum-exp: 2 M tasks over the threads (task = r for rows, M + r for columns)
    for (int q = tid; q < 2 * M; q += blockDim.x) {
        const bool col = q >= M;
        const int r = col ? q - M : q;
        float m1 = -INFINITY;
        for (int c = 0; c < M; ++c) if (uf[c] != 0.f) m1 = fmaxf(m1, col ? L[c * LP + r] : L[r * LP + c]);
        float s1 = 0.f;
        for (int c = 0; c < M; ++c) if (uf[c] != 0.f) s1 += expf((col ? L[c * LP + r] : L[r * LP + c]) - m1);
        (col ? cl : rl)[r] = m1 + logf(s1);
    }
    __syncthreads();
    if (!GRAD) {
        float acc = 0.f;
        for (int r = tid; r < M; r += blockDim.x) if (uf[r] != 0.f) acc += (rl[r] - L[r * LP + r]) + (cl[r] - L[r * LP + r]);
        acc = block_sum(acc, sh);
        if (tid == 0) {
            out[0] = cnt > 0.f ? acc / (2.f * cnt) : 0.f;
            out[1] = cnt > 0.f ? cnt : 1.f;      // batch_size["aux_lang"] of hulc2.py:391-394: the masked-in rows, 1 when there are none
        }
        return;
    }
    const float g = cnt > 0.f ? gout[0] / (2.f * cnt) : 0.f;
    float ds = 0.f;
    for (int i = tid; i < M * M; i += blockDim.x) {
        const int r = i / M, c = i % M;
        float dl = 0.f;
        if (uf[r] != 0.f && uf[c] != 0.f) {
            const float lv = L[r * LP + c];
            dl = g * (expf(lv - rl[r]) + expf(lv - cl[c]) - (r == c ? 2.f : 0.f));
            ds += dl * lv;                       // d/d logit_scale of s*dot = L
        }
        L[r * LP + c] = dl;
    }
    ds = block_sum(ds, sh);                      // contains a __syncthreads: all dL written
    if (tid == 0) dscale[0] = ds;
    for (int i = tid; i < M * CLIP_D; i += blockDim.x) {
        const int r = i / CLIP_D, d = i % CLIP_D;
        float a = 0.f, b = 0.f;
        for (int c = 0; c < M; ++c) { a += L[r * LP + c] * t[c * DP + d]; b += L[c * LP + r] * n[c * DP + d]; }
        dn[r * DP + d] = s * a;                  // gradient w.r.t. the normalised feature (projected below)
        dt[r * DP + d] = s * b;
    }
    __syncthreads();
    // projection onto the tangent of the unit sphere, one (row, column) per thread; the row's dot product is recomputed per element from LDS
    for (int i = tid; i < M * CLIP_D; i += blockDim.x) {
        const int r = i / CLIP_D, q = r * DP + i % CLIP_D;
        float pa = 0.f, pb = 0.f;
        for (int d = 0; d < CLIP_D; ++d) { pa += dn[r * DP + d] * n[r * DP + d]; pb += dt[r * DP + d] * t[r * DP + d]; }
        dim_[i] = (dn[q] - n[q] * pa) / ni[r];
        dtx[i] = (dt[q] - t[q] * pb) / ti[r];
    }
}

// ------------------------------------------------------------------------------------------------
// world -> tcp frame (one thread per (b, s)); R = Rx(a) Ry(b) Rz(c)
// ------------------------------------------------------------------------------------------------
HULC_DEVICE void euler_xyz(float a, float b, float c, float (&R)[3][3]) {
    const float ca = cosf(a), sa = sinf(a), cb = cosf(b), sb = sinf(b), cc = cosf(c), sc = sinf(c);
    R[0][0] = cb * cc;                R[0][1] = -cb * sc;               R[0][2] = sb;
    R[1][0] = sa * sb * cc + ca * sc; R[1][1] = -sa * sb * sc + ca * cc; R[1][2] = -sa * cb;
    R[2][0] = -ca * sb * cc + sa * sc; R[2][1] = ca * sb * sc + sa * cc; R[2][2] = ca * cb;
}
HULC_DEVICE void world_to_tcp_row(const float* a, const float* o, float* y) {
    float R[3][3], Rn[3][3];
    euler_xyz(o[3], o[4], o[5], R);
    euler_xyz(o[3] + 0.01f * a[3], o[4] + 0.01f * a[4], o[5] + 0.01f * a[5], Rn);
    for (int r = 0; r < 3; ++r) y[r] = R[0][r] * a[0] + R[1][r] * a[1] + R[2][r] * a[2];      // R^-1 p = R^T p
    float M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = Rn[0][r] * R[0][c] + Rn[1][r] * R[1][c] + Rn[2][r] * R[2][c];   // Rn^T R
    float e[3] = {atan2f(-M[1][2], M[2][2]), asinf(fminf(fmaxf(M[0][2], -1.f), 1.f)), atan2f(-M[0][1], M[0][0])};
    const float pi = 3.14159265358979323846f;
    for (int r = 0; r < 3; ++r) {
        if (e[r] < -pi) e[r] += 2 * pi;
        if (e[r] > pi) e[r] -= 2 * pi;
        y[3 + r] = e[r] * 100.f;
    }
    y[6] = a[6];
}
__global__ void world_to_tcp_kernel(const float* __restrict__ act, const float* __restrict__ obs, int n, int obs_dim, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    world_to_tcp_row(act + (long)i * 7, obs + (long)i * obs_dim, out + (long)i * 7);
}
// The decoder's target actions for a step, in the row order the recurrent kernel leaves its outputs in: nseg modality batches (B, S, 7)
// [+ their robot_obs (B, S, obs_dim)] -> out row (s * nseg * B + seg * B + b), world -> tcp frame applied on the way (to_tcp) or copied.
// One launch instead of two concatenations, the frame change and a transposing copy.
struct ActSegP { const float* act[4]; const float* obs[4]; };
__global__ void actions_time_major_kernel(ActSegP p, int nseg, int B, int S, int obs_dim, int to_tcp, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nseg * B * S) return;
    const int st = i / (nseg * B), rem = i % (nseg * B), seg = rem / B, b = rem % B;
    const float* a = p.act[seg] + ((long)b * S + st) * 7;
    float* y = out + (long)i * 7;
    if (to_tcp) world_to_tcp_row(a, p.obs[seg] + ((long)b * S + st) * obs_dim, y);
    else for (int r = 0; r < 7; ++r) y[r] = a[r];
}

// tcp -> world frame (gripper_control.py:39-63): pos_w = R p, R_new = R * R(0.01*orn)^-1, orn_w = euler(R_new) - euler_obs, wrapped, x100
__global__ void tcp_to_world_kernel(const float* __restrict__ act, const float* __restrict__ obs, int n, int obs_dim, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float* a = act + (long)i * 7;
    const float* o = obs + (long)i * obs_dim;
    float R[3][3], Q[3][3];
    euler_xyz(o[3], o[4], o[5], R);
    euler_xyz(0.01f * a[3], 0.01f * a[4], 0.01f * a[5], Q);
    float* y = out + (long)i * 7;
    for (int r = 0; r < 3; ++r) y[r] = R[r][0] * a[0] + R[r][1] * a[1] + R[r][2] * a[2];
    float M[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) M[r][c] = R[r][0] * Q[c][0] + R[r][1] * Q[c][1] + R[r][2] * Q[c][2];       // R Q^T (Q^-1 = Q^T)
    float e[3] = {atan2f(-M[1][2], M[2][2]), asinf(fminf(fmaxf(M[0][2], -1.f), 1.f)), atan2f(-M[0][1], M[0][0])};
    const float pi = 3.14159265358979323846f;
    for (int r = 0; r < 3; ++r) {
        float d = e[r] - o[3 + r];
        if (d < -pi) d += 2 * pi;
        if (d > pi) d -= 2 * pi;
        y[3 + r] = d * 100.f;
    }
    y[6] = a[6];
}

// ------------------------------------------------------------------------------------------------
// sampling from the logistic mixture (logistic_decoder_rnn.py:231-255): Gumbel-max over the mixtures, inverse-CDF draw from the
// selected logistic, gripper = bounds[argmax].  One thread per (token, action dimension).  u_mix / u_inv (optional) inject the
// raw uniforms torch.rand would have produced (parity tests); otherwise the counter RNG supplies them.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mix_sample_kernel(MixP p, const float* __restrict__ u_mix, const float* __restrict__ u_inv,
                                                         unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                         const float* __restrict__ gripper_bounds, float* __restrict__ act_out,
                                                         long* __restrict__ idx_out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)p.T * (p.A + 1)) return;
    if (seed_dev) seed ^= seed_dev[0];
    const int t = (int)(i / (p.A + 1)), a = (int)(i % (p.A + 1));
    const float* row = p.y + (long)t * p.ld;
    const int n = p.A * p.NM;
    if (a == p.A) {                                                  // gripper command: first maximum of the two logits
        const int g = row[3 * n + 1] > row[3 * n] ? 1 : 0;
        act_out[(long)t * (p.A + 1) + p.A] = gripper_bounds[g];
        return;
    }
    const float c = (float)(1e-5 - (1.0 - 1e-5)), r2 = (float)(1.0 - 1e-5);   // u = (r1 - r2) * rand + r2, as separate mul and add
    int best = 0; float best_v = -INFINITY;
    for (int k = 0; k < p.NM; ++k) {
        const float raw = u_mix ? u_mix[((long)t * p.A + a) * p.NM + k] : hulc_uniform01(seed, (uint64_t)(((long)t * p.A + a) * p.NM + k));
        const float u = __fadd_rn(__fmul_rn(c, raw), r2);
        const float v = row[a * p.NM + k] - logf(-logf(u));
        if (v > best_v) { best_v = v; best = k; }
    }
    const float mean = row[n + a * p.NM + best];
    const float ls = fmaxf(row[2 * n + a * p.NM + best], p.log_scale_min);
    const float raw = u_inv ? u_inv[(long)t * p.A + a] : hulc_uniform01(seed ^ 0x9E3779B97F4A7C15ull, (uint64_t)((long)t * p.A + a));
    const float u = __fadd_rn(__fmul_rn(c, raw), r2);
    act_out[(long)t * (p.A + 1) + a] = mean + expf(ls) * (logf(u) - logf(1.0f - u));
    if (idx_out) idx_out[(long)t * p.A + a] = best;
}

MixP make_mix(const hulc_mix_desc* d, const float* y, const float* act) {
    MixP p;
    p.y = y; p.ld = d->ld; p.act = act; p.amin = d->act_min; p.amax = d->act_max;
    p.T = d->T; p.A = d->A; p.NM = d->n_mix; p.num_classes = d->num_classes;
    p.log_scale_min = d->log_scale_min; p.gripper_alpha = d->gripper_alpha;
    p.tm_B = d->time_major_B; p.nseg = d->nseg < 1 ? 1 : d->nseg;
    return p;
}

}  // namespace

static int mix_check(const hulc_mix_desc* d, const char* who) {
    if (d->n_mix > 16 || d->n_mix <= 0) return hulc_fail(-2, "hulc_mix_loss: n_mix must be in 1..16");
    if (d->nseg < 1 || d->T % d->nseg != 0) return hulc_fail(-3, "hulc_mix_loss: T must split into nseg equal segments");
    if (d->time_major_B < 0 || (d->time_major_B > 0 && (d->T % d->time_major_B || d->time_major_B % d->nseg)))
        return hulc_fail(-3, "hulc_mix_loss: time-major rows need T % B == 0 and B % nseg == 0");
    (void)who;
    return 0;
}
extern "C" long hulc_mix_loss_workspace(const hulc_mix_desc* d) {
    if (!d || d->nseg < 1) return -1;
    const int items = (d->T / d->nseg) * (d->A + 1);
    return (long)d->nseg * ((items + 15) / 16) * 2 * (long)sizeof(float);     // 16 items (of 16 mixture lanes) per workgroup
}
extern "C" int hulc_mix_loss_fwd(const hulc_mix_desc* d, const float* y, const float* act, float* out, void* ws, void* stream) {
    if (!d || !y || !act || !out || !ws || !d->act_min || !d->act_max) return hulc_fail(-1, "hulc_mix_loss_fwd: null pointer");
    int rc = mix_check(d, "fwd"); if (rc) return rc;
    const int seg_tokens = d->T / d->nseg, bps = (seg_tokens * (d->A + 1) + 15) / 16;
    hipStream_t s = (hipStream_t)stream;
    mix_loss_partial_kernel<<<d->nseg * bps, 256, 0, s>>>(make_mix(d, y, act), seg_tokens, bps, (float*)ws);
    mix_loss_final_kernel<<<d->nseg, 64, 0, s>>>((const float*)ws, bps, seg_tokens, d->gripper_alpha, out);
    return hulc_check_launch("hulc_mix_loss_fwd");
}
extern "C" int hulc_mix_loss_bwd(const hulc_mix_desc* d, const float* y, const float* act, const float* gout, float* dy, long ld_dy,
                                 void* stream) {
    if (!d || !y || !act || !gout || !dy) return hulc_fail(-1, "hulc_mix_loss_bwd: null pointer");
    int rc = mix_check(d, "bwd"); if (rc) return rc;
    const int n = d->T * (d->A + 1);
    mix_loss_bwd_kernel<<<(n + 15) / 16, 256, 0, (hipStream_t)stream>>>(make_mix(d, y, act), d->T / d->nseg, gout, dy, ld_dy);
    return hulc_check_launch("hulc_mix_loss_bwd");
}

extern "C" int hulc_cat_kl_fwd(const float* pp, const float* pr, int B, int G, int CLS, float beta, int nseg, float* out, float* kl_group, void* stream) {
    if (!pp || !pr || !out || !kl_group) return hulc_fail(-1, "hulc_cat_kl_fwd: null pointer");
    if (CLS != 32) return hulc_fail(-2, "hulc_cat_kl_fwd: class_size must be 32 (one 32-lane sub-wave per category)");
    if (nseg < 1 || B % nseg) return hulc_fail(-2, "hulc_cat_kl_fwd: the batch must split evenly into nseg segments");
    cat_kl_group_kernel<<<(B * G + 7) / 8, 256, 0, (hipStream_t)stream>>>(pp, pr, B * G, kl_group);
    cat_kl_sum_kernel<<<nseg, 1024, 0, (hipStream_t)stream>>>(kl_group, B / nseg * G, B / nseg, beta, out);
    return hulc_check_launch("hulc_cat_kl_fwd");
}
extern "C" int hulc_cat_kl_bwd(const float* pp, const float* pr, const float* kl_group, int B, int G, int CLS, float beta, float mix,
                               const float* gout, int nseg, float* dpp, float* dpr, void* stream) {
    if (!pp || !pr || !kl_group || !gout || !dpp || !dpr) return hulc_fail(-1, "hulc_cat_kl_bwd: null pointer");
    if (CLS != 32) return hulc_fail(-2, "hulc_cat_kl_bwd: class_size must be 32");
    if (nseg < 1 || B % nseg) return hulc_fail(-2, "hulc_cat_kl_bwd: the batch must split evenly into nseg segments");
    cat_kl_bwd_kernel<<<(B * G + 7) / 8, 256, 0, (hipStream_t)stream>>>(pp, pr, kl_group, B, G, beta, mix, gout, nseg, dpp, dpr);
    return hulc_check_launch("hulc_cat_kl_bwd");
}
extern "C" int hulc_plan_sample_fwd(const float* logits, const long* idx_in, unsigned long long seed, const unsigned long long* seed_dev,
                                    int NG, int CLS, long* idx_out, float* plan, void* stream) {
    if (!logits || !plan) return hulc_fail(-1, "hulc_plan_sample_fwd: null pointer");
    if (CLS != 32) return hulc_fail(-2, "hulc_plan_sample_fwd: class_size must be 32");
    plan_sample_kernel<<<(NG + 7) / 8, 256, 0, (hipStream_t)stream>>>(logits, idx_in, seed, seed_dev, NG, idx_out, plan);
    return hulc_check_launch("hulc_plan_sample_fwd");
}
extern "C" int hulc_plan_sample_bwd(const float* logits, const float* dplan, int NG, int CLS, float* dlogits, int accumulate, void* stream) {
    if (!logits || !dplan || !dlogits) return hulc_fail(-1, "hulc_plan_sample_bwd: null pointer");
    if (CLS != 32) return hulc_fail(-2, "hulc_plan_sample_bwd: class_size must be 32");
    plan_sample_bwd_kernel<<<(NG + 7) / 8, 256, 0, (hipStream_t)stream>>>(logits, dplan, NG, dlogits, accumulate);
    return hulc_check_launch("hulc_plan_sample_bwd");
}

static size_t clip_smem(int M) { return ((size_t)4 * M * (CLIP_D + 1) + (size_t)M * (M + 1) + 5 * (size_t)M) * sizeof(float); }
template <bool G> static int clip_lds_ok() {        // up to 132 KB at M = 128: above the default dynamic-LDS limit
    static int rc = hipFuncSetAttribute((const void*)clip_loss_kernel<G>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)clip_smem(CLIP_MAXM)) == hipSuccess ? 0 : -8;
    return rc;
}

extern "C" int hulc_clip_loss_fwd(const float* im, const float* tx, const unsigned char* use, int row0, const float* logit_scale, int M, int D,
                                  float* out, void* stream) {
    if (!im || !tx || !use || !logit_scale || !out) return hulc_fail(-1, "hulc_clip_loss_fwd: null pointer");
    if (M > CLIP_MAXM || M <= 0 || D != CLIP_D || row0 < 0 || row0 >= M) return hulc_fail(-2, "hulc_clip_loss_fwd: needs M <= 128, D == 32, 0 <= row0 < M");
    if (clip_lds_ok<false>()) return hulc_fail(-8, "hulc_clip_loss_fwd: could not raise the dynamic LDS limit");
    clip_loss_kernel<false><<<1, 256, clip_smem(M - row0), (hipStream_t)stream>>>(im, tx, use, row0, logit_scale, M - row0, out, nullptr, nullptr, nullptr, nullptr);
    return hulc_check_launch("hulc_clip_loss_fwd");
}
extern "C" int hulc_clip_loss_bwd(const float* im, const float* tx, const unsigned char* use, int row0, const float* logit_scale, int M, int D,
                                  const float* gout, float* dim, float* dtx, float* dscale, void* stream) {
    if (!im || !tx || !use || !logit_scale || !gout || !dim || !dtx || !dscale) return hulc_fail(-1, "hulc_clip_loss_bwd: null pointer");
    if (M > CLIP_MAXM || M <= 0 || D != CLIP_D || row0 < 0 || row0 >= M) return hulc_fail(-2, "hulc_clip_loss_bwd: needs M <= 128, D == 32, 0 <= row0 < M");
    if (clip_lds_ok<true>()) return hulc_fail(-8, "hulc_clip_loss_bwd: could not raise the dynamic LDS limit");
    clip_loss_kernel<true><<<1, 256, clip_smem(M - row0), (hipStream_t)stream>>>(im, tx, use, row0, logit_scale, M - row0, nullptr, gout, dim, dtx, dscale);
    return hulc_check_launch("hulc_clip_loss_bwd");
}

// total = (sum_m act[m] + sum_m kl[m]) / n + beta * clip  — the scalar tail of Hulc2.training_step (hulc2.py:400-430) as one launch per
// direction (a dozen 0-dim torch kernels otherwise).  out = {total, kl mean, action mean, beta * clip, per-modality act + kl ...}
__global__ void loss_combine_fwd_kernel(const float* __restrict__ kls, const float* __restrict__ acts, const float* __restrict__ clip, int n, float beta,
                                        float* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float sk = 0.f, sa = 0.f, st = 0.f;
    for (int m = 0; m < n; ++m) { const float t = acts[m] + kls[m]; out[4 + m] = t; sk += kls[m]; sa += acts[m]; st += t; }
    float total = st / n;
    const float wc = clip ? beta * clip[0] : 0.f;
    if (clip) total = total + wc;
    out[0] = total; out[1] = sk / n; out[2] = sa / n; out[3] = wc;
}
__global__ void loss_combine_bwd_kernel(const float* __restrict__ g, int n, float beta, float* __restrict__ dkls, float* __restrict__ dacts,
                                        float* __restrict__ dclip) {
    const int m = threadIdx.x;
    if (m < n) { dkls[m] = g[0] / n; dacts[m] = g[0] / n; }
    if (m == 0 && dclip) dclip[0] = g[0] * beta;
}
extern "C" int hulc_loss_combine_fwd(const float* kls, const float* acts, const float* clip, int n, float beta, float* out, void* stream) {
    if (!kls || !acts || !out) return hulc_fail(-1, "hulc_loss_combine_fwd: null pointer");
    if (n < 1 || n > 64) return hulc_fail(-2, "hulc_loss_combine_fwd: 1..64 modalities");
    loss_combine_fwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(kls, acts, clip, n, beta, out);
    return hulc_check_launch("hulc_loss_combine_fwd");
}
extern "C" int hulc_loss_combine_bwd(const float* g, int n, float beta, float* dkls, float* dacts, float* dclip, void* stream) {
    if (!g || !dkls || !dacts) return hulc_fail(-1, "hulc_loss_combine_bwd: null pointer");
    if (n < 1 || n > 64) return hulc_fail(-2, "hulc_loss_combine_bwd: 1..64 modalities");
    loss_combine_bwd_kernel<<<1, 64, 0, (hipStream_t)stream>>>(g, n, beta, dkls, dacts, dclip);
    return hulc_check_launch("hulc_loss_combine_bwd");
}

extern "C" int hulc_actions_time_major(const float* const* act, const float* const* robot_obs, int nseg, int B, int S, int obs_dim, int to_tcp,
                                       float* out, void* stream) {
    if (!act || !out || (to_tcp && !robot_obs)) return hulc_fail(-1, "hulc_actions_time_major: null pointer");
    if (nseg < 1 || nseg > 4 || B < 1 || S < 1 || (to_tcp && obs_dim < 6)) return hulc_fail(-2, "hulc_actions_time_major: 1..4 segments, robot_obs with the euler angles in columns 3:6");
    ActSegP p{};
    for (int i = 0; i < nseg; ++i) {
        if (!act[i] || (to_tcp && !robot_obs[i])) return hulc_fail(-1, "hulc_actions_time_major: null segment pointer");
        p.act[i] = act[i];
        p.obs[i] = to_tcp ? robot_obs[i] : nullptr;
    }
    const int n = nseg * B * S;
    actions_time_major_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(p, nseg, B, S, obs_dim, to_tcp, out);
    return hulc_check_launch("hulc_actions_time_major");
}

extern "C" int hulc_world_to_tcp(const float* act, const float* robot_obs, int n, int obs_dim, float* out, void* stream) {
    if (!act || !robot_obs || !out) return hulc_fail(-1, "hulc_world_to_tcp: null pointer");
    if (obs_dim < 6) return hulc_fail(-2, "hulc_world_to_tcp: robot_obs needs the euler angles in columns 3:6");
    world_to_tcp_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(act, robot_obs, n, obs_dim, out);
    return hulc_check_launch("hulc_world_to_tcp");
}

extern "C" int hulc_tcp_to_world(const float* act, const float* robot_obs, int n, int obs_dim, float* out, void* stream) {
    if (!act || !robot_obs || !out) return hulc_fail(-1, "hulc_tcp_to_world: null pointer");
    if (obs_dim < 6) return hulc_fail(-2, "hulc_tcp_to_world: robot_obs needs the euler angles in columns 3:6");
    tcp_to_world_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(act, robot_obs, n, obs_dim, out);
    return hulc_check_launch("hulc_tcp_to_world");
}

extern "C" int hulc_mix_sample(const hulc_mix_desc* d, const float* y, const float* u_mix, const float* u_inv, unsigned long long seed,
                               const unsigned long long* seed_dev, const float* gripper_bounds, float* act_out, long* idx_out, void* stream) {
    if (!d || !y || !gripper_bounds || !act_out) return hulc_fail(-1, "hulc_mix_sample: null pointer");
    if (d->T <= 0 || d->A <= 0 || d->n_mix <= 0 || d->ld < 3L * d->A * d->n_mix + 2) return hulc_fail(-2, "hulc_mix_sample: bad geometry");
    MixP p = make_mix(d, y, nullptr);
    const long n = (long)d->T * (d->A + 1);
    mix_sample_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(p, u_mix, u_inv, seed, seed_dev, gripper_bounds, act_out, idx_out);
    return hulc_check_launch("hulc_mix_sample");
}
