// ffn_fused.hip — feed-forward block of the plan-recognition transformer layer, fused (bf16 compute).
//
// reference arithmetic: nn.TransformerEncoderLayer's linear1 -> ReLU -> dropout -> linear2 (d_model 128, dim_feedforward 2048),
// hulc2/models/plan_encoders/plan_recognition_net.py:108-117, and its autograd backward.
//
// As four GEMM launches per layer and direction the block materialises the (tokens x 2048) hidden activation in HBM (fp32), re-reads
// it three times and pays a launch floor per GEMM.  Here the hidden activation never leaves the CU:
//   forward : workgroup = 64 tokens x 128 hidden units.  z = x W1s^T (+b1, ReLU, dropout) is computed by MFMA straight into LDS as
//             bf16 and immediately consumed as the A operand of the second product f_partial = h W2s^T; weights go global ->
//             registers as B fragments (each wave needs only its own 32 rows).  The 16 hidden slices leave fp32 partials that one
//             fixed-order pass sums (b2 rides in slice 0).
//   backward: same decomposition, the hidden activation is RECOMPUTED (1/3 more FLOPs, zero HBM bytes): z, dh = (df W2s) * gate,
//             the input-gradient partial dh W1s, and the two weight-gradient tiles dh^T x and df^T h accumulated in registers over
//             the token tiles a workgroup walks; transposed copies of x, df, dh, h are written to LDS so that every MFMA operand
//             is a 16-byte LDS read.  Partials are combined by fixed-order passes (bit-reproducible).
// Dropout uses the same counter RNG stream as the unfused path (element index = token * 2048 + hidden unit), so both give the
// same masks.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

constexpr int D = 128;          // model width
constexpr int HS = 128;         // hidden units per workgroup
constexpr int TT = 64;          // tokens per tile
constexpr int ROW = D * 2 + 16; // LDS row stride of a [*][128] bf16 tile (272 B: conflict-free ds_read_b128)
constexpr int ROWT = TT * 2 + 16;   // LDS row stride of a transposed [128][64 tokens] bf16 tile (144 B)

struct FfnP {
    const float* x; const float* df;                 // (T, 128) fp32
    const uint16_t *W1, *W2, *W1T, *W2T;             // bf16: W1 [FF][128], W2 [128][FF], W1T [128][FF], W2T [FF][128]
    const float *b1, *b2;
    int T, FF;
    float drop_p; unsigned long long seed; const unsigned long long* seed_dev;
    float* f_slab;                                   // fwd: [FF/HS][T][128]
    float* dx_slab;                                  // bwd: [FF/HS][T][128]
    float* dw1_slab; float* dw2_slab; float* db1_slab;   // bwd: [G][FF/HS][128][128] x2, [G][FF]
};

HULC_DEVICE bf16x8_t ldg_frag(const uint16_t* p) { union { uint4 u; bf16x8_t b; } x; x.u = *(const uint4*)p; return x.b; }

// stage a (TT x 128) fp32 token tile as bf16 rows (row stride ROW); rows beyond T are zero
HULC_DEVICE void stage_rows(char* dst, const float* src, int t0, int T, int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int id = tid + q * 512, row = id >> 4, ch = id & 15;
        const bool ok = t0 + row < T;
        const float4* p = (const float4*)(src + (long)(ok ? t0 + row : 0) * D + ch * 8);
        const float4 a = p[0], b = p[1];
        uint4 v = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
        if (!ok) v = make_uint4(0u, 0u, 0u, 0u);
        *(uint4*)(dst + row * ROW + ch * 16) = v;
    }
}
// the same tile transposed: [feature][token] bf16 (row stride ROWT)
HULC_DEVICE void stage_transposed(char* dst, const float* src, int t0, int T, int tid) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int id = tid + q * 512, row = id & 63, ch = id >> 6;          // consecutive lanes -> consecutive tokens (2-byte column writes)
        const bool ok = t0 + row < T;
        const float4* p = (const float4*)(src + (long)(ok ? t0 + row : 0) * D + ch * 8);
        const float4 a = p[0], b = p[1];
        const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) *(uint16_t*)(dst + (ch * 8 + j) * ROWT + row * 2) = ok ? f32_to_bf16_bits(v[j]) : (uint16_t)0;
    }
}

// ---------------------------------------------------------------- forward
__global__ __launch_bounds__(512) void ffn_fwd_kernel(FfnP p) {
    __shared__ __attribute__((aligned(16))) char xs[TT * ROW];
    __shared__ __attribute__((aligned(16))) char hs[TT * ROW];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int t0 = blockIdx.x * TT, slice = blockIdx.y, j0 = slice * HS;
    const int mt = wave & 1, ct = wave >> 1;                                 // token tile (32 rows), column tile (32 of 128)
    const unsigned long long seed = p.seed ^ (p.seed_dev ? p.seed_dev[0] : 0ull);

    bf16x8_t w[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w[ks] = ldg_frag(p.W1 + (long)(j0 + ct * 32 + r) * D + ks * 16 + h * 8);
    stage_rows(xs, p.x, t0, p.T, tid);
    __syncthreads();
    f32x16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const bf16x8_t a = *(const bf16x8_t*)(xs + (mt * 32 + r) * ROW + (ks * 16 + h * 8) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w[ks], acc, 0, 0, 0);           // D[token][hidden]
    }
    {   // bias, ReLU, dropout -> bf16 hidden tile in LDS
        const int j = j0 + ct * 32 + r;
        const float bj = p.b1[j];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = mt * 32 + acc_row(e, lane);
            float v = fmaxf(acc[e] + bj, 0.f);
            if (p.drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)(t0 + row) * (uint64_t)p.FF + j, p.drop_p);
            *(uint16_t*)(hs + row * ROW + (ct * 32 + r) * 2) = f32_to_bf16_bits(v);
        }
    }
    // second product: B fragments = W2 rows (output features), k = this slice's hidden units
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) w[ks] = ldg_frag(p.W2 + (long)(ct * 32 + r) * p.FF + j0 + ks * 16 + h * 8);
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const bf16x8_t a = *(const bf16x8_t*)(hs + (mt * 32 + r) * ROW + (ks * 16 + h * 8) * 2);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w[ks], acc, 0, 0, 0);           // D[token][out feature]
    }
    const int n = ct * 32 + r;
    const float bn = slice == 0 ? p.b2[n] : 0.f;
    float* out = p.f_slab + (long)slice * p.T * D;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = t0 + mt * 32 + acc_row(e, lane);
        if (t < p.T) out[(long)t * D + n] = acc[e] + bn;
    }
}

// ---------------------------------------------------------------- backward
__global__ __launch_bounds__(512) void ffn_bwd_kernel(FfnP p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* xs = smem;                     // [64][128]  x rows            (A of z)
    char* dfs = xs + TT * ROW;           // [64][128]  df rows           (A of dhd)
    char* dhs = dfs + TT * ROW;          // [64][128]  dh rows           (A of dx)
    char* xT = dhs + TT * ROW;           // [128][64]  x^T               (B of dW1)
    char* dfT = xT + D * ROWT;           // [128][64]  df^T              (A of dW2)
    char* dhT = dfT + D * ROWT;          // [128][64]  dh^T              (A of dW1)
    char* hdT = dhT + HS * ROWT;         // [128][64]  h^T               (B of dW2)
    float* bred = (float*)(hdT + HS * ROWT);   // [2][128] bias-gradient partials of the two token halves
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int slice = blockIdx.y, j0 = slice * HS, G = gridDim.x;
    const int mt = wave & 1, ct = wave >> 1;
    const unsigned long long seed = p.seed ^ (p.seed_dev ? p.seed_dev[0] : 0ull);

    // weight fragments are (re)loaded per token tile just ahead of their product: they are L2-resident (1.5 MB for the whole
    // block) and keeping all three sets live next to the 64 weight-gradient accumulators would spill
    const float b1j = p.b1[j0 + ct * 32 + r];
    // weight-gradient accumulators: wave (jt = wave >> 1) owns rows jt*32.., two column tiles 2*(wave & 1) + {0, 1}
    f32x16_t gw1[2], gw2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) { gw1[i][e] = 0.f; gw2[i][e] = 0.f; }
    float gb1 = 0.f;

    for (int tile = blockIdx.x; tile < ntiles; tile += G) {
        const int t0 = tile * TT;
        __syncthreads();                                                     // previous tile's LDS reads are done
        stage_rows(xs, p.x, t0, p.T, tid);
        stage_rows(dfs, p.df, t0, p.T, tid);
        stage_transposed(xT, p.x, t0, p.T, tid);
        stage_transposed(dfT, p.df, t0, p.T, tid);
        bf16x8_t w1[8], w2t[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            w1[ks] = ldg_frag(p.W1 + (long)(j0 + ct * 32 + r) * D + ks * 16 + h * 8);        // rows = hidden, k = in features
            w2t[ks] = ldg_frag(p.W2T + (long)(j0 + ct * 32 + r) * D + ks * 16 + h * 8);      // rows = hidden, k = out features
        }
        __syncthreads();
        // ---- z = x W1s^T and dhd = df W2s for this wave's (token tile, hidden tile)
        f32x16_t az, ad;
#pragma unroll
        for (int e = 0; e < 16; ++e) { az[e] = 0.f; ad[e] = 0.f; }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const bf16x8_t ax = *(const bf16x8_t*)(xs + (mt * 32 + r) * ROW + (ks * 16 + h * 8) * 2);
            const bf16x8_t ag = *(const bf16x8_t*)(dfs + (mt * 32 + r) * ROW + (ks * 16 + h * 8) * 2);
            az = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ax, w1[ks], az, 0, 0, 0);
            ad = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ag, w2t[ks], ad, 0, 0, 0);
        }
        bf16x8_t w1t[8];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) w1t[ks] = ldg_frag(p.W1T + (long)(ct * 32 + r) * p.FF + j0 + ks * 16 + h * 8);   // rows = in features, k = hidden
        {   // gate, dh, h: row-major dh for the input gradient, transposed dh / h for the weight gradients
            const int jl = ct * 32 + r, j = j0 + jl;
            float bsum = 0.f;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                float hv[4], dv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int e = g4 * 4 + q, row = mt * 32 + acc_row(e, lane);
                    const float z = az[e] + b1j;
                    const float keep = p.drop_p > 0.f ? dropout_scale(seed, (uint64_t)(t0 + row) * (uint64_t)p.FF + j, p.drop_p) : 1.f;
                    const bool live = z > 0.f && t0 + row < p.T;
                    hv[q] = live ? z * keep : 0.f;
                    dv[q] = live ? ad[e] * keep : 0.f;
                    bsum += dv[q];
                    *(uint16_t*)(dhs + row * ROW + jl * 2) = f32_to_bf16_bits(dv[q]);
                }
                const int tok = mt * 32 + 8 * g4 + 4 * h;                   // the 4 registers of a group are 4 consecutive tokens
                *(uint2*)(dhT + jl * ROWT + tok * 2) = make_uint2(pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3]));
                *(uint2*)(hdT + jl * ROWT + tok * 2) = make_uint2(pack_bf16x2(hv[0], hv[1]), pack_bf16x2(hv[2], hv[3]));
            }
            bsum += __shfl_xor(bsum, 32, 64);
            if (h == 0) bred[mt * HS + jl] = bsum;
        }
        __syncthreads();
        if (tid < HS) gb1 += bred[tid] + bred[HS + tid];
        // ---- input gradient partial: dx = dh W1s  (token tile mt, in-feature tile ct)
        {
            f32x16_t ax;
#pragma unroll
            for (int e = 0; e < 16; ++e) ax[e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 8; ++ks) {
                const bf16x8_t a = *(const bf16x8_t*)(dhs + (mt * 32 + r) * ROW + (ks * 16 + h * 8) * 2);
                ax = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, w1t[ks], ax, 0, 0, 0);
            }
            float* out = p.dx_slab + (long)slice * p.T * D;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int t = t0 + mt * 32 + acc_row(e, lane);
                if (t < p.T) out[(long)t * D + ct * 32 + r] = ax[e];
            }
        }
        // ---- weight gradients over this tile's 64 tokens: dW1s += dh^T x, dW2s += df^T h   (rows tile jt, column tiles 2*(wave&1)+i)
        {
            const int jt = wave >> 1;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ko = (ks * 16 + h * 8) * 2;
                const bf16x8_t a1 = *(const bf16x8_t*)(dhT + (jt * 32 + r) * ROWT + ko);      // rows = hidden
                const bf16x8_t a2 = *(const bf16x8_t*)(dfT + (jt * 32 + r) * ROWT + ko);      // rows = out features
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int cto = (2 * (wave & 1) + i) * 32 + r;
                    const bf16x8_t bx = *(const bf16x8_t*)(xT + cto * ROWT + ko);             // cols = in features
                    const bf16x8_t bh = *(const bf16x8_t*)(hdT + cto * ROWT + ko);            // cols = hidden
                    gw1[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bx, gw1[i], 0, 0, 0);
                    gw2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, bh, gw2[i], 0, 0, 0);
                }
            }
        }
    }
    // ---- partial weight gradients of this workgroup (summed over token groups by ffn_wgrad_reduce_kernel)
    const long slab = ((long)blockIdx.x * gridDim.y + slice) * HS * D;
    const int jt = wave >> 1;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int col = (2 * (wave & 1) + i) * 32 + r;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int row = jt * 32 + acc_row(e, lane);
            p.dw1_slab[slab + (long)row * D + col] = gw1[i][e];            // [hidden local][in feature]
            p.dw2_slab[slab + (long)row * HS + col] = gw2[i][e];           // [out feature][hidden local]
        }
    }
    if (tid < HS) p.db1_slab[(long)blockIdx.x * p.FF + j0 + tid] = gb1;
}

// dW1 [FF][128], dW2 [128][FF], db1 [FF] (+)= sum over the G token groups, fixed order
// sum of G slabs at one offset, four independent loads per trip
HULC_DEVICE float slab_sum(const float* __restrict__ base, long stride, int G) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int g = 0;
    for (; g + 3 < G; g += 4) {
        const float a = base[(long)g * stride], b = base[(long)(g + 1) * stride], c = base[(long)(g + 2) * stride], d = base[(long)(g + 3) * stride];
        s0 += a; s1 += b; s2 += c; s3 += d;
    }
    for (; g < G; ++g) s0 += base[(long)g * stride];
    return (s0 + s1) + (s2 + s3);
}

__global__ __launch_bounds__(256) void ffn_wgrad_reduce_kernel(FfnP p, int G, float* dW1, float* dW2, float* db1, int accumulate) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long n1 = (long)p.FF * D, nslab = (long)(p.FF / HS) * HS * D;
    if (i < n1) {                                                            // dW1[(s*128 + j)][c]: slab index s*16384 + j*128 + c = i
        const float v = slab_sum(p.dw1_slab + i, nslab, G);
        dW1[i] = accumulate ? dW1[i] + v : v;
    } else if (i < 2 * n1) {                                                 // dW2[n][s*128 + j]
        const long o = i - n1;
        const int n = (int)(o / p.FF), jj = (int)(o % p.FF), s = jj / HS, j = jj % HS;
        const long src = (long)s * HS * D + (long)n * HS + j;
        const float v = slab_sum(p.dw2_slab + src, nslab, G);
        dW2[o] = accumulate ? dW2[o] + v : v;
    } else if (i < 2 * n1 + p.FF) {
        const long o = i - 2 * n1;
        const float v = slab_sum(p.db1_slab + o, p.FF, G);
        db1[o] = accumulate ? db1[o] + v : v;
    }
}

// out[i] (+)= sum_s slab[s][i]   (the 16 hidden slices of f or dx)
__global__ __launch_bounds__(256) void ffn_slice_sum_kernel(const float* __restrict__ slab, float* __restrict__ out, int nslab, long n, int accumulate) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int s = 0; s < nslab; ++s) {
        const float4 a = *(const float4*)(slab + (long)s * n + i);
        v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w;
    }
    if (accumulate) { const float4 o = *(const float4*)(out + i); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
    *(float4*)(out + i) = v;
}

int ffn_groups(int T) {
    static const int gmax = getenv("HULC_FFN_GROUPS") ? atoi(getenv("HULC_FFN_GROUPS")) : 16;   // 16 groups x 16 hidden slices = one workgroup per CU; 32 groups doubled the partial slabs (0.140 vs 0.104 ms per 2 launches)
    const int nt = (T + TT - 1) / TT;
    return nt < gmax ? nt : gmax;
}

}  // namespace

extern "C" long hulc_ffn_workspace(int T, int FF) {
    const long ns = FF / HS;
    const long fwd = ns * T * D * 4;
    const long bwd = ns * T * D * 4 + (long)ffn_groups(T) * (2 * ns * HS * D + FF) * 4;
    return fwd > bwd ? fwd : bwd;
}

static int ffn_check(int T, int Dm, int FF, const char* who) {
    if (Dm != D || FF % HS || FF <= 0 || T <= 0) return hulc_fail(-2, who);
    return 0;
}

// see include/hulc2_amd.h
extern "C" int hulc_ffn_fwd(const float* x, const void* W1, const float* b1, const void* W2, const float* b2, int T, int Dm, int FF, float drop_p,
                            unsigned long long seed, const unsigned long long* seed_dev, float* f, void* ws, void* stream) {
    if (!x || !W1 || !b1 || !W2 || !b2 || !ws) return hulc_fail(-1, "hulc_ffn_fwd: null pointer");
    if (int rc = ffn_check(T, Dm, FF, "hulc_ffn_fwd: needs d_model 128 and dim_feedforward a multiple of 128")) return rc;
    FfnP p = {};
    p.x = x; p.W1 = (const uint16_t*)W1; p.W2 = (const uint16_t*)W2; p.b1 = b1; p.b2 = b2; p.T = T; p.FF = FF;
    p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev; p.f_slab = (float*)ws;
    hipStream_t s = (hipStream_t)stream;
    ffn_fwd_kernel<<<dim3((T + TT - 1) / TT, FF / HS), 512, 0, s>>>(p);
    const long n = (long)T * D;
    if (f) ffn_slice_sum_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, s>>>(p.f_slab, f, FF / HS, n, 0);
    return hulc_check_launch("hulc_ffn_fwd");
}

extern "C" int hulc_ffn_bwd(const float* x, const float* df, const void* W1, const float* b1, const void* W1T, const void* W2T, int T, int Dm, int FF,
                            float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* dx, int dx_accumulate,
                            float* dW1, float* db1, float* dW2, int accumulate_params, void* ws, void* stream) {
    if (!x || !df || !W1 || !b1 || !W1T || !W2T || !dW1 || !db1 || !dW2 || !ws) return hulc_fail(-1, "hulc_ffn_bwd: null pointer");
    if (int rc = ffn_check(T, Dm, FF, "hulc_ffn_bwd: needs d_model 128 and dim_feedforward a multiple of 128")) return rc;
    const int ns = FF / HS, G = ffn_groups(T), ntiles = (T + TT - 1) / TT;
    FfnP p = {};
    p.x = x; p.df = df; p.W1 = (const uint16_t*)W1; p.W1T = (const uint16_t*)W1T; p.W2T = (const uint16_t*)W2T; p.b1 = b1; p.T = T; p.FF = FF;
    p.drop_p = drop_p; p.seed = seed; p.seed_dev = seed_dev;
    p.dx_slab = (float*)ws;
    p.dw1_slab = p.dx_slab + (long)ns * T * D;
    p.dw2_slab = p.dw1_slab + (long)G * ns * HS * D;
    p.db1_slab = p.dw2_slab + (long)G * ns * HS * D;
    hipStream_t s = (hipStream_t)stream;
    const size_t lds = (size_t)3 * TT * ROW + 2 * D * ROWT + 2 * HS * ROWT + 2 * HS * 4;
    static bool attr = false;
    if (!attr) {
        if (hipFuncSetAttribute((const void*)ffn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
            return hulc_fail(-8, "hulc_ffn_bwd: could not raise the dynamic LDS limit");
        attr = true;
    }
    ffn_bwd_kernel<<<dim3(G, ns), 512, lds, s>>>(p, ntiles);
    const long n = (long)T * D;
    if (dx) ffn_slice_sum_kernel<<<(unsigned)((n / 4 + 255) / 256), 256, 0, s>>>(p.dx_slab, dx, ns, n, dx_accumulate);
    const long nr = 2L * FF * D + FF;
    ffn_wgrad_reduce_kernel<<<(unsigned)((nr + 255) / 256), 256, 0, s>>>(p, G, dW1, dW2, db1, accumulate_params);
    return hulc_check_launch("hulc_ffn_bwd");
}
