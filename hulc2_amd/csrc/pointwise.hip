// pointwise.hip — HBM-bound reductions of the perceptual encoders and the plan-recognition transformer,
// written as 64-lane wavefront kernels (lane = channel / feature column, no cross-lane traffic where the
// layout allows it).
//
//   spatial softmax   : hulc2/models/perceptual_encoders/vision_network.py:100-108
//   (residual+dropout+) LayerNorm : nn.LayerNorm in vision_network.py:53, goal_encoders.py:28,61 and the
//                       post-norm nn.TransformerEncoderLayer of plan_recognition_net.py:115-117
//   column sums       : bias gradients of every Linear
//   sequence mean, positional-embedding add : plan_recognition_net.py:133-136,145
//   self-attention    : nn.MultiheadAttention inside the encoder layer (S <= 32, head_dim 16)
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

// ------------------------------------------------------------------------------------------------
// spatial softmax, NHWC input [N][HW][C], C <= 64 (lane = channel)
// ------------------------------------------------------------------------------------------------
// One workgroup (4 waves) per frame, lane = channel; each wave walks a quarter of the HW positions and the four
// (max, sum, sum*x, sum*y) partials are merged through LDS in a fixed order.
__global__ __launch_bounds__(256) void spatial_softmax_fwd_kernel(const void* __restrict__ x, int x_dtype, int N, int HW, int C,
                                                                  const float* __restrict__ xmap, const float* __restrict__ ymap,
                                                                  const float* __restrict__ temperature, float* __restrict__ out,
                                                                  float* __restrict__ stats) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x;
    const bool act = lane < C;
    const float invT = 1.0f / temperature[0];
    const long base = (long)n * HW * C + lane;
    const int per = (HW + 3) / 4, p0 = wave * per, p1 = p0 + per < HW ? p0 + per : HW;
    float m = -INFINITY;
    if (act)
        for (int p = p0; p < p1; ++p) m = fmaxf(m, load_elem(x, x_dtype, base + (long)p * C) * invT);
    red[wave][0][lane] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0][0][lane], red[1][0][lane]), fmaxf(red[2][0][lane], red[3][0][lane]));
    float s = 0.f, sx = 0.f, sy = 0.f;
    if (act)
        for (int p = p0; p < p1; ++p) {
            const float e = __expf(load_elem(x, x_dtype, base + (long)p * C) * invT - m);
            s += e; sx += e * xmap[p]; sy += e * ymap[p];
        }
    red[wave][1][lane] = s; red[wave][2][lane] = sx; red[wave][3][lane] = sy;
    __syncthreads();
    if (wave == 0 && act) {
        s = red[0][1][lane] + red[1][1][lane] + red[2][1][lane] + red[3][1][lane];
        sx = red[0][2][lane] + red[1][2][lane] + red[2][2][lane] + red[3][2][lane];
        sy = red[0][3][lane] + red[1][3][lane] + red[2][3][lane] + red[3][3][lane];
        const float inv = 1.0f / s;
        out[(long)n * 2 * C + 2 * lane] = sx * inv;
        out[(long)n * 2 * C + 2 * lane + 1] = sy * inv;
        stats[((long)n * C + lane) * 2] = m;
        stats[((long)n * C + lane) * 2 + 1] = s;
    }
}

// dz[n][p][c] = (x > 0) * (1/T) * softmax_p * (gx*(xmap_p - ex) + gy*(ymap_p - ey)); 4 waves split the positions
__global__ __launch_bounds__(256) void spatial_softmax_bwd_kernel(const void* __restrict__ x, int x_dtype, int N, int HW, int C,
                                                                  const float* __restrict__ xmap, const float* __restrict__ ymap,
                                                                  const float* __restrict__ temperature, const float* __restrict__ out,
                                                                  const float* __restrict__ stats, const float* __restrict__ dout,
                                                                  void* __restrict__ dx, int dx_dtype, int relu_mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = blockIdx.x;
    if (lane >= C) return;
    const float invT = 1.0f / temperature[0];
    const long base = (long)n * HW * C + lane;
    const float m = stats[((long)n * C + lane) * 2], inv = 1.0f / stats[((long)n * C + lane) * 2 + 1];
    const float ex = out[(long)n * 2 * C + 2 * lane], ey = out[(long)n * 2 * C + 2 * lane + 1];
    const float gx = dout[(long)n * 2 * C + 2 * lane], gy = dout[(long)n * 2 * C + 2 * lane + 1];
    const int per = (HW + 3) / 4, p0 = wave * per, p1 = p0 + per < HW ? p0 + per : HW;
    for (int p = p0; p < p1; ++p) {
        const float xv = load_elem(x, x_dtype, base + (long)p * C);
        const float pr = __expf(xv * invT - m) * inv;
        float g = invT * pr * (gx * (xmap[p] - ex) + gy * (ymap[p] - ey));
        if (relu_mask && !(xv > 0.f)) g = 0.f;
        store_elem(dx, dx_dtype, base + (long)p * C, g);
    }
}

// ---- C == 64 fast path: a lane owns 8 channels of one position (one 16-byte bf16 / two 16-byte fp32 loads), 8 lanes cover a
// position, a wave 8 positions per step; ONE pass over the frame with an online (running-max) softmax, partials merged across
// the 8 position lanes by shuffles and across the 4 waves through LDS in a fixed order.
struct SsmAcc { float m, s, sx, sy; };
HULC_DEVICE void ssm_merge(SsmAcc& a, float m2, float s2, float sx2, float sy2) {
    const float mn = fmaxf(a.m, m2);
    const float c1 = a.m == -INFINITY ? 0.f : __expf(a.m - mn), c2 = m2 == -INFINITY ? 0.f : __expf(m2 - mn);   // empty partials weigh 0
    a.s = a.s * c1 + s2 * c2; a.sx = a.sx * c1 + sx2 * c2; a.sy = a.sy * c1 + sy2 * c2; a.m = mn;
}
__global__ __launch_bounds__(256) void spatial_softmax_fwd64_kernel(const void* __restrict__ x, int x_dtype, int HW,
                                                                    const float* __restrict__ xmap, const float* __restrict__ ymap,
                                                                    const float* __restrict__ temperature, float* __restrict__ out,
                                                                    float* __restrict__ stats) {
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 7, pl = lane >> 3;
    const int n = blockIdx.x;
    const float invT = 1.0f / temperature[0];
    const long base = (long)n * HW * 64 + cg * 8;
    SsmAcc a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j].m = -INFINITY; a[j].s = 0.f; a[j].sx = 0.f; a[j].sy = 0.f; }
    for (int p = wave * 8 + pl; p < HW; p += 32) {
        Chunk8 c; chunk_load_contig(c, x, x_dtype, base + (long)p * 64);
        const float xm = xmap[p], ym = ymap[p];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = c.v[j] * invT;
            const float mn = fmaxf(a[j].m, v);
            const float sc = __expf(a[j].m - mn), e = __expf(v - mn);
            a[j].s = a[j].s * sc + e; a[j].sx = a[j].sx * sc + e * xm; a[j].sy = a[j].sy * sc + e * ym; a[j].m = mn;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
            ssm_merge(a[j], __shfl_xor(a[j].m, o, 64), __shfl_xor(a[j].s, o, 64), __shfl_xor(a[j].sx, o, 64), __shfl_xor(a[j].sy, o, 64));
        if (pl == 0) { red[wave][0][cg * 8 + j] = a[j].m; red[wave][1][cg * 8 + j] = a[j].s; red[wave][2][cg * 8 + j] = a[j].sx; red[wave][3][cg * 8 + j] = a[j].sy; }
    }
    __syncthreads();
    if (wave == 0) {
        SsmAcc t; t.m = red[0][0][lane]; t.s = red[0][1][lane]; t.sx = red[0][2][lane]; t.sy = red[0][3][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) ssm_merge(t, red[w][0][lane], red[w][1][lane], red[w][2][lane], red[w][3][lane]);
        const float inv = 1.0f / t.s;
        out[(long)n * 128 + 2 * lane] = t.sx * inv;
        out[(long)n * 128 + 2 * lane + 1] = t.sy * inv;
        stats[((long)n * 64 + lane) * 2] = t.m;
        stats[((long)n * 64 + lane) * 2 + 1] = t.s;
    }
}
// HW <= 448 (the 21 x 21 map of the static camera): a lane's 14 positions fit in registers, so the maximum is taken first and every element
// costs ONE exponential instead of the two of the online form above (which is VALU-bound: 2.6 TB/s).  Same partials, same merge tree.
template <int DT>       // storage type of the map: HULC_F32 / HULC_BF16 / HULC_F16
__global__ __launch_bounds__(256) void spatial_softmax_fwd64_regs_kernel(const void* __restrict__ x, int HW, const float* __restrict__ xmap,
                                                                         const float* __restrict__ ymap, const float* __restrict__ temperature,
                                                                         float* __restrict__ out, float* __restrict__ stats) {
    constexpr int IT = 14;
    __shared__ float red[4][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 7, pl = lane >> 3;
    const int n = blockIdx.x;
    const float invT = 1.0f / temperature[0];
    const long base = (long)n * HW * 64 + cg * 8;
    float v[IT][8];
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int p = wave * 8 + pl + 32 * k;
        const bool on = p < HW;
        const long off = base + (long)(on ? p : 0) * 64;
        float c[8];
        if (DT == HULC_BF16) {
            const uint4 r = *(const uint4*)((const uint16_t*)x + off);
            const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) { c[2 * j] = __uint_as_float(w[j] << 16); c[2 * j + 1] = __uint_as_float(w[j] & 0xffff0000u); }
        } else if (DT == HULC_F16) {
            union { uint4 u; _Float16 hh[8]; } r; r.u = *(const uint4*)((const uint16_t*)x + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) c[j] = (float)r.hh[j];
        } else {
            const float4* q = (const float4*)((const float*)x + off);
            const float4 a0 = q[0], a1 = q[1];
            c[0] = a0.x; c[1] = a0.y; c[2] = a0.z; c[3] = a0.w; c[4] = a1.x; c[5] = a1.y; c[6] = a1.z; c[7] = a1.w;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) v[k][j] = on ? c[j] * invT : -INFINITY;
    }
    SsmAcc a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float m = v[0][j];                                   // position k = 0 exists for every lane (HW >= 32)
#pragma unroll
        for (int k = 1; k < IT; ++k) m = fmaxf(m, v[k][j]);
        a[j].m = m; a[j].s = 0.f; a[j].sx = 0.f; a[j].sy = 0.f;
    }
#pragma unroll
    for (int k = 0; k < IT; ++k) {
        const int p = wave * 8 + pl + 32 * k;
        const int pc = p < HW ? p : 0;
        const float xm = xmap[pc], ym = ymap[pc];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float e = __expf(v[k][j] - a[j].m);        // exp(-inf) = 0 for the positions past HW
            a[j].s += e; a[j].sx += e * xm; a[j].sy += e * ym;
        }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
            ssm_merge(a[j], __shfl_xor(a[j].m, o, 64), __shfl_xor(a[j].s, o, 64), __shfl_xor(a[j].sx, o, 64), __shfl_xor(a[j].sy, o, 64));
        if (pl == 0) { red[wave][0][cg * 8 + j] = a[j].m; red[wave][1][cg * 8 + j] = a[j].s; red[wave][2][cg * 8 + j] = a[j].sx; red[wave][3][cg * 8 + j] = a[j].sy; }
    }
    __syncthreads();
    if (wave == 0) {
        SsmAcc t; t.m = red[0][0][lane]; t.s = red[0][1][lane]; t.sx = red[0][2][lane]; t.sy = red[0][3][lane];
#pragma unroll
        for (int w = 1; w < 4; ++w) ssm_merge(t, red[w][0][lane], red[w][1][lane], red[w][2][lane], red[w][3][lane]);
        const float inv = 1.0f / t.s;
        out[(long)n * 128 + 2 * lane] = t.sx * inv;
        out[(long)n * 128 + 2 * lane + 1] = t.sy * inv;
        stats[((long)n * 64 + lane) * 2] = t.m;
        stats[((long)n * 64 + lane) * 2 + 1] = t.s;
    }
}
__global__ __launch_bounds__(256) void spatial_softmax_bwd64_kernel(const void* __restrict__ x, int x_dtype, int HW,
                                                                    const float* __restrict__ xmap, const float* __restrict__ ymap,
                                                                    const float* __restrict__ temperature, const float* __restrict__ out,
                                                                    const float* __restrict__ stats, const float* __restrict__ dout,
                                                                    void* __restrict__ dx, int dx_dtype, int relu_mask) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int cg = lane & 7, pl = lane >> 3;
    const int n = blockIdx.x;
    const float invT = 1.0f / temperature[0];
    const long base = (long)n * HW * 64 + cg * 8;
    float m[8], k[8], ex[8], ey[8], gx[8], gy[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = cg * 8 + j;
        m[j] = stats[((long)n * 64 + c) * 2]; k[j] = invT / stats[((long)n * 64 + c) * 2 + 1];
        ex[j] = out[(long)n * 128 + 2 * c]; ey[j] = out[(long)n * 128 + 2 * c + 1];
        gx[j] = dout[(long)n * 128 + 2 * c]; gy[j] = dout[(long)n * 128 + 2 * c + 1];
    }
    for (int p = wave * 8 + pl; p < HW; p += 32) {
        Chunk8 c; chunk_load_contig(c, x, x_dtype, base + (long)p * 64);
        const float xm = xmap[p], ym = ymap[p];
        float g[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            g[j] = k[j] * __expf(c.v[j] * invT - m[j]) * (gx[j] * (xm - ex[j]) + gy[j] * (ym - ey[j]));
            if (relu_mask && !(c.v[j] > 0.f)) g[j] = 0.f;
        }
        const long o = base + (long)p * 64;
        if (dx_dtype == HULC_BF16) *(uint4*)((uint16_t*)dx + o) = make_uint4(pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]), pack_bf16x2(g[4], g[5]), pack_bf16x2(g[6], g[7]));
        else { ((float4*)((float*)dx + o))[0] = make_float4(g[0], g[1], g[2], g[3]); ((float4*)((float*)dx + o))[1] = make_float4(g[4], g[5], g[6], g[7]); }
    }
}


// ------------------------------------------------------------------------------------------------
// LayerNorm over the last dimension D <= 256 (one wave per row, up to 4 elements per lane)
//   pre = x + dropout(o)   (o optional)   y = (pre - mean) * rstd * gamma + beta
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ o, float drop_p,
                                                            unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                            const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps, int R, int D,
                                                            float* __restrict__ pre_out, float* __restrict__ y,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                            int n_o = 1, long o_stride = 0, long ld_y = 0) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
    if (seed_dev) seed ^= seed_dev[0];
    float v[4];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + q * 64;
        v[q] = 0.f;
        if (c < D) {
            const long i = (long)r * D + c;
            float t = x[i];
            if (o) {
                float ov = o[i];
                {   // o given as partial slabs (fused feed-forward slices), fixed order; four independent loads per trip
                    int sl = 1;
                    for (; sl + 3 < n_o; sl += 4) {
                        const float a0 = o[(long)sl * o_stride + i], a1 = o[(long)(sl + 1) * o_stride + i], a2 = o[(long)(sl + 2) * o_stride + i],
                                    a3 = o[(long)(sl + 3) * o_stride + i];
                        ov += a0; ov += a1; ov += a2; ov += a3;
                    }
                    for (; sl < n_o; ++sl) ov += o[(long)sl * o_stride + i];
                }
                if (drop_p > 0.f) ov *= dropout_scale(seed, (uint64_t)i, drop_p);
                t += ov;
            }
            v[q] = t;
            if (pre_out) pre_out[i] = t;
            s += t;
        }
    }
    const float mean = wave_sum(s) / D;
    float ss = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + q * 64;
        if (c < D) { const float dlt = v[q] - mean; ss += dlt * dlt; }
    }
    const float rstd = rsqrtf(wave_sum(ss) / D + eps);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int c = lane + q * 64;
        if (c < D) y[(long)r * (ld_y ? ld_y : D) + c] = (v[q] - mean) * rstd * gamma[c] + beta[c];
    }
    if (lane == 0) { mean_out[r] = mean; rstd_out[r] = rstd; }
}

// dpre = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma;  partial dgamma/dbeta per block.
// do_out (optional) = dpre * dropout keep-scale (the gradient of the dropped branch o).
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ pre,
                                                            const float* __restrict__ mean_in, const float* __restrict__ rstd_in,
                                                            const float* __restrict__ gamma, int R, int D, int rows_per_block,
                                                            float* __restrict__ dpre, float* __restrict__ do_out, float drop_p,
                                                            unsigned long long seed, const unsigned long long* __restrict__ seed_dev,
                                                            float* __restrict__ partial, long ld_dy = 0) {
    __shared__ float red[4][2][256];
    if (seed_dev) seed ^= seed_dev[0];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[4] = {0.f, 0.f, 0.f, 0.f}, db[4] = {0.f, 0.f, 0.f, 0.f};
    const int r0 = blockIdx.x * rows_per_block;
    for (int r = r0 + wave; r < r0 + rows_per_block && r < R; r += 4) {
        const float mean = mean_in[r], rstd = rstd_in[r];
        float g[4], xh[4], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane + q * 64;
            g[q] = 0.f; xh[q] = 0.f;
            if (c < D) {
                const long i = (long)r * D + c;
                const float d = dy[ld_dy ? (long)r * ld_dy + c : i];
                xh[q] = (pre[i] - mean) * rstd;
                g[q] = d * gamma[c];
                dg[q] += d * xh[q]; db[q] += d;
                s1 += g[q]; s2 += g[q] * xh[q];
            }
        }
        s1 = wave_sum(s1) / D; s2 = wave_sum(s2) / D;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = lane + q * 64;
            if (c < D) {
                const long i = (long)r * D + c;
                const float v = rstd * (g[q] - s1 - xh[q] * s2);
                dpre[i] = v;
                if (do_out) do_out[i] = drop_p > 0.f ? v * dropout_scale(seed, (uint64_t)i, drop_p) : v;
            }
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) { red[wave][0][lane + q * 64] = dg[q]; red[wave][1][lane + q * 64] = db[q]; }
    __syncthreads();
    for (int c = threadIdx.x; c < D; c += 256) {
        float a = 0.f, b = 0.f;
        for (int w = 0; w < 4; ++w) { a += red[w][0][c]; b += red[w][1][c]; }
        partial[(long)blockIdx.x * 2 * D + c] = a;
        partial[(long)blockIdx.x * 2 * D + D + c] = b;
    }
}

// ------------------------------------------------------------------------------------------------
// column sums: out[n] = sum_m x[m][n]   (bias gradients).  HBM-bound: M*N*sizeof(x) bytes read once.
// Workgroup = 4 waves; a lane owns 4 adjacent columns (one 16-byte / 8-byte load per row), the waves interleave rows,
// 4 rows are in flight per lane.  Row blocks are sized so that >= ~1024 workgroups exist; partials are summed in a
// fixed order by reduce_rows_wide_kernel (deterministic).
// ------------------------------------------------------------------------------------------------
__device__ inline float4 load4(const void* x, int dt, long off) {
    if (dt == HULC_F32) return *(const float4*)((const float*)x + off);
    const uint2 v = *(const uint2*)((const uint16_t*)x + off);
    return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__global__ __launch_bounds__(256) void colsum_kernel(const void* __restrict__ x, int x_dtype, long M, int N, long ld, long rows_per_block,
                                                     float* __restrict__ partial, float* __restrict__ direct_out, int accumulate, int vec) {
    __shared__ float4 red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = (blockIdx.x * 64 + lane) * 4;
    const long r0 = (long)blockIdx.y * rows_per_block;
    long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    if (vec) {
        if (n < N) {
            long r = r0 + wave;
            float4 a0 = s, a1 = s, a2 = s, a3 = s;
            for (; r + 12 < r1; r += 16) {
                const float4 v0 = load4(x, x_dtype, r * ld + n), v1 = load4(x, x_dtype, (r + 4) * ld + n);
                const float4 v2 = load4(x, x_dtype, (r + 8) * ld + n), v3 = load4(x, x_dtype, (r + 12) * ld + n);
                a0.x += v0.x; a0.y += v0.y; a0.z += v0.z; a0.w += v0.w;
                a1.x += v1.x; a1.y += v1.y; a1.z += v1.z; a1.w += v1.w;
                a2.x += v2.x; a2.y += v2.y; a2.z += v2.z; a2.w += v2.w;
                a3.x += v3.x; a3.y += v3.y; a3.z += v3.z; a3.w += v3.w;
            }
            for (; r < r1; r += 4) { const float4 v = load4(x, x_dtype, r * ld + n); a0.x += v.x; a0.y += v.y; a0.z += v.z; a0.w += v.w; }
            s.x = (a0.x + a1.x) + (a2.x + a3.x); s.y = (a0.y + a1.y) + (a2.y + a3.y);
            s.z = (a0.z + a1.z) + (a2.z + a3.z); s.w = (a0.w + a1.w) + (a2.w + a3.w);
        }
    } else {
        float* sp = (float*)&s;
        for (int j = 0; j < 4; ++j)
            if (n + j < N)
                for (long r = r0 + wave; r < r1; r += 4) sp[j] += load_elem(x, x_dtype, r * ld + n + j);
    }
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        const float4 a = red[0][lane], b = red[1][lane], c = red[2][lane], d = red[3][lane];
        const float v[4] = {(a.x + b.x) + (c.x + d.x), (a.y + b.y) + (c.y + d.y), (a.z + b.z) + (c.z + d.z), (a.w + b.w) + (c.w + d.w)};
        for (int j = 0; j < 4; ++j) {
            if (n + j >= N) break;
            if (direct_out) direct_out[n + j] = accumulate ? direct_out[n + j] + v[j] : v[j];      // single row block: no second pass
            else partial[(long)blockIdx.y * N + n + j] = v[j];
        }
    }
}
// out[r] (+)= sum_q partial[q*ld + r]: 64 columns x 16 row slices per workgroup, fixed summation order
// blockIdx.y = 1 (optional second half of a launch): the same sum for partial + off1 into out1 — LayerNorm's dgamma | dbeta in one launch
__global__ __launch_bounds__(1024) void reduce_rows_wide_kernel(const float* __restrict__ partial, float* __restrict__ out, int P, long R, long ld,
                                                                int accumulate, long off1 = 0, float* __restrict__ out1 = nullptr) {
    __shared__ float red[16][64];
    if (blockIdx.y == 1) { partial += off1; out = out1; }
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long rr = (long)blockIdx.x * 64 + lane;
    const int per = (P + 15) / 16;
    const int q0 = sl * per, q1 = q0 + per < P ? q0 + per : P;
    float s = 0.f;
    if (rr < R)
        for (int q = q0; q < q1; ++q) s += partial[(long)q * ld + rr];
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && rr < R) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w][lane];
        out[rr] = accumulate ? out[rr] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------
// sequence mean and positional embedding
// ------------------------------------------------------------------------------------------------
__global__ void seq_mean_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int S, int D, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * D) return;
    const int b = (int)(i / D), d = (int)(i % D);
    // four independent loads per trip (a single accumulator chain waits for every load's round trip in turn)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const float* xb = x + (long)b * S * D + d;
    int t = 0;
    for (; t + 3 < S; t += 4) {
        const float a = xb[(long)t * D], c = xb[(long)(t + 1) * D], e = xb[(long)(t + 2) * D], f = xb[(long)(t + 3) * D];
        s0 += a; s1 += c; s2 += e; s3 += f;
    }
    for (; t < S; ++t) s0 += xb[(long)t * D];
    const float s = (s0 + s1) + (s2 + s3);
    y[i] = scale * (s / S);
}
// y[b][d] = scale * sum_s x[b*stride_b + s*stride_s + d]  (any layout: time-major recurrent buffers, strided halves)
__global__ void strided_seq_sum_kernel(const void* __restrict__ x, int x_dtype, float* __restrict__ y, int B, int S, int D, long stride_b,
                                       long stride_s, long ldy, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * D) return;
    const int b = (int)(i / D), d = (int)(i % D);
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    const long o = (long)b * stride_b + d;
    int t = 0;
    for (; t + 3 < S; t += 4) {
        const float a = load_elem(x, x_dtype, o + (long)t * stride_s), c = load_elem(x, x_dtype, o + (long)(t + 1) * stride_s),
                    e = load_elem(x, x_dtype, o + (long)(t + 2) * stride_s), f = load_elem(x, x_dtype, o + (long)(t + 3) * stride_s);
        s0 += a; s1 += c; s2 += e; s3 += f;
    }
    for (; t < S; ++t) s0 += load_elem(x, x_dtype, o + (long)t * stride_s);
    y[(long)b * ldy + d] = scale * ((s0 + s1) + (s2 + s3));
}

__global__ void seq_mean_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, int B, int S, int D) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * D) return;
    const int d = (int)(i % D); const long bs = i / D; const int b = (int)(bs / S);
    dx[i] = dy[(long)b * D + d] / S;
}
// y[b][s][:] = dropout(x[b][s][:] + pos[pos_ids[s]][:])
__global__ void add_pos_fwd_kernel(const float* __restrict__ x, const float* __restrict__ pos, const long* __restrict__ pos_ids,
                                   float* __restrict__ y, int B, int S, int D, float drop_p, unsigned long long seed,
                                   const unsigned long long* __restrict__ seed_dev) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * D) return;
    if (seed_dev) seed ^= seed_dev[0];
    const int d = (int)(i % D); const int s = (int)((i / D) % S);
    float v = x[i] + pos[pos_ids[s] * D + d];
    if (drop_p > 0.f) v *= dropout_scale(seed, (uint64_t)i, drop_p);
    y[i] = v;
}
__global__ void dropout_bwd_kernel(const float* __restrict__ dy, float* __restrict__ dx, long n, float drop_p, unsigned long long seed,
                                   const unsigned long long* __restrict__ seed_dev) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (seed_dev) seed ^= seed_dev[0];
    dx[i] = dy[i] * dropout_scale(seed, (uint64_t)i, drop_p);
}

// dx = dy * (y > 0) * scale   (gradient through ReLU [+ inverted dropout] given the saved output y)
__global__ void relu_bwd_kernel(const float* __restrict__ dy, const void* __restrict__ y, int y_dtype, float* __restrict__ dx, long n, float scale) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    dx[i] = load_elem(y, y_dtype, i) > 0.f ? dy[i] * scale : 0.f;
}

// ------------------------------------------------------------------------------------------------
// multi-head self-attention for short sequences: one wave per (batch, head); S <= 32, head_dim = 16.
// qkv: [B*S][3E] rows = token b*S+s, columns [q | k | v], head h owns columns h*16..h*16+15 of each.
// ------------------------------------------------------------------------------------------------
#define ATT_DH 16
__global__ __launch_bounds__(64) void attention_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ out, float* __restrict__ probs,
                                                           int B, int S, int H, float scale, float drop_p, unsigned long long seed,
                                                           const unsigned long long* __restrict__ seed_dev) {
    __shared__ float q[32][ATT_DH + 1], k[32][ATT_DH + 1], v[32][ATT_DH + 1];
    if (seed_dev) seed ^= seed_dev[0];
    const int lane = threadIdx.x, b = blockIdx.x / H, h = blockIdx.x % H, E = H * ATT_DH;
    for (int idx = lane; idx < S * ATT_DH; idx += 64) {
        const int s = idx / ATT_DH, d = idx % ATT_DH;
        const long row = ((long)b * S + s) * 3 * E + h * ATT_DH + d;
        q[s][d] = qkv[row]; k[s][d] = qkv[row + E]; v[s][d] = qkv[row + 2 * E];
    }
    __syncthreads();
    const int i = lane & 31, half = lane >> 5;
    float sc[16];
    float m = -INFINITY;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = half * 16 + jj;
        float a = -INFINITY;
        if (i < S && j < S) {
            a = 0.f;
#pragma unroll
            for (int d = 0; d < ATT_DH; ++d) a += q[i][d] * k[j][d];
            a *= scale;
        }
        sc[jj] = a; m = fmaxf(m, a);
    }
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) { sc[jj] = (i < S && half * 16 + jj < S) ? __expf(sc[jj] - m) : 0.f; sum += sc[jj]; }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = i < S ? 1.0f / sum : 0.f;
    float o[ATT_DH];
#pragma unroll
    for (int d = 0; d < ATT_DH; ++d) o[d] = 0.f;
    const long pbase = (((long)b * H + h) * S + i) * S;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = half * 16 + jj;
        if (i < S && j < S) {
            float p = sc[jj] * inv;
            if (drop_p > 0.f) p *= dropout_scale(seed, (uint64_t)(pbase + j), drop_p);
            probs[pbase + j] = p;           // post-dropout probabilities (what multiplies V)
#pragma unroll
            for (int d = 0; d < ATT_DH; ++d) o[d] += p * v[j][d];
        }
    }
#pragma unroll
    for (int d = 0; d < ATT_DH; ++d) o[d] += __shfl_xor(o[d], 32, 64);
    if (i < S) {
        float* dst = out + ((long)b * S + i) * E + h * ATT_DH + half * 8;
#pragma unroll
        for (int d = 0; d < 8; ++d) dst[d] = o[half * 8 + d];
    }
}

// probs holds the post-dropout probabilities P'; the pre-dropout softmax is P = P' / keepscale where
// kept, and dropped entries contribute no gradient (their dP' is multiplied by 0).
__global__ __launch_bounds__(64) void attention_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ probs,
                                                           const float* __restrict__ dout, float* __restrict__ dqkv, int B, int S, int H,
                                                           float scale, float drop_p, unsigned long long seed,
                                                           const unsigned long long* __restrict__ seed_dev) {
    __shared__ float q[32][ATT_DH + 1], k[32][ATT_DH + 1], v[32][ATT_DH + 1], go[32][ATT_DH + 1];
    if (seed_dev) seed ^= seed_dev[0];
    __shared__ float ds[32][33], pp[32][33];
    const int lane = threadIdx.x, b = blockIdx.x / H, h = blockIdx.x % H, E = H * ATT_DH;
    for (int idx = lane; idx < S * ATT_DH; idx += 64) {
        const int s = idx / ATT_DH, d = idx % ATT_DH;
        const long row = ((long)b * S + s) * 3 * E + h * ATT_DH + d;
        q[s][d] = qkv[row]; k[s][d] = qkv[row + E]; v[s][d] = qkv[row + 2 * E];
        go[s][d] = dout[((long)b * S + s) * E + h * ATT_DH + d];
    }
    for (int idx = lane; idx < 32 * 33; idx += 64) { (&ds[0][0])[idx] = 0.f; (&pp[0][0])[idx] = 0.f; }
    __syncthreads();
    const int i = lane & 31, half = lane >> 5;
    const long pbase = (((long)b * H + h) * S + i) * S;
    float dp[16], pr[16];
    float dot = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = half * 16 + jj;
        dp[jj] = 0.f; pr[jj] = 0.f;
        if (i < S && j < S) {
            const float pd = probs[pbase + j];                   // P' (post-dropout)
            float ks = 1.f;
            if (drop_p > 0.f) ks = dropout_scale(seed, (uint64_t)(pbase + j), drop_p);
            float a = 0.f;
#pragma unroll
            for (int d = 0; d < ATT_DH; ++d) a += go[i][d] * v[j][d];   // dP'
            const float p = ks > 0.f ? pd / ks : 0.f;                    // P where kept (dropped: gradient is 0 anyway)
            dp[jj] = a * ks;                                             // dP
            pr[jj] = p;
            dot += dp[jj] * p;
            pp[i][j] = pd;
        }
    }
    // note: for dropped entries p is unknown (set 0) but dP = 0 there, so dot and dS are exact except
    // that dS for a dropped entry must still be -P*dot; recover P from a second softmax pass below.
    dot += __shfl_xor(dot, 32, 64);
    if (drop_p > 0.f) {
        // recompute the exact pre-dropout softmax row (cheap: S <= 32)
        float sc[16], m = -INFINITY, sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
            const int j = half * 16 + jj;
            float a = -INFINITY;
            if (i < S && j < S) {
                a = 0.f;
#pragma unroll
                for (int d = 0; d < ATT_DH; ++d) a += q[i][d] * k[j][d];
                a *= scale;
            }
            sc[jj] = a; m = fmaxf(m, a);
        }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) { sc[jj] = (i < S && half * 16 + jj < S) ? __expf(sc[jj] - m) : 0.f; sum += sc[jj]; }
        sum += __shfl_xor(sum, 32, 64);
        dot = 0.f;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) { pr[jj] = i < S ? sc[jj] / sum : 0.f; dot += dp[jj] * pr[jj]; }
        dot += __shfl_xor(dot, 32, 64);
    }
    float dq[ATT_DH];
#pragma unroll
    for (int d = 0; d < ATT_DH; ++d) dq[d] = 0.f;
#pragma unroll
    for (int jj = 0; jj < 16; ++jj) {
        const int j = half * 16 + jj;
        if (i < S && j < S) {
            const float dsv = pr[jj] * (dp[jj] - dot) * scale;
            ds[i][j] = dsv;
#pragma unroll
            for (int d = 0; d < ATT_DH; ++d) dq[d] += dsv * k[j][d];
        }
    }
#pragma unroll
    for (int d = 0; d < ATT_DH; ++d) dq[d] += __shfl_xor(dq[d], 32, 64);
    if (i < S) {
        float* dst = dqkv + ((long)b * S + i) * 3 * E + h * ATT_DH + half * 8;
#pragma unroll
        for (int d = 0; d < 8; ++d) dst[d] = dq[half * 8 + d];
    }
    __syncthreads();
    // key/value side: lane owns key j = lane & 31 and 8 of the 16 head dims
    const int j = i;
    if (j < S) {
        float dk[8], dv[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
        for (int ii = 0; ii < S; ++ii) {
            const float a = ds[ii][j], pv = pp[ii][j];
#pragma unroll
            for (int d = 0; d < 8; ++d) { dk[d] += a * q[ii][half * 8 + d]; dv[d] += pv * go[ii][half * 8 + d]; }
        }
        float* dst = dqkv + ((long)b * S + j) * 3 * E + h * ATT_DH + half * 8;
#pragma unroll
        for (int d = 0; d < 8; ++d) { dst[E + d] = dk[d]; dst[2 * E + d] = dv[d]; }
    }
}

}  // namespace

extern "C" int hulc_spatial_softmax_fwd(const void* x, int x_dtype, int N, int HW, int C, const float* xmap, const float* ymap,
                                        const float* temperature, float* out, float* stats, void* stream) {
    if (!x || !xmap || !ymap || !temperature || !out || !stats) return hulc_fail(-1, "hulc_spatial_softmax_fwd: null pointer");
    if (C > 64 || C <= 0) return hulc_fail(-2, "hulc_spatial_softmax_fwd: C must be in 1..64 (lane = channel)");
    if (x_dtype == HULC_F16 && (C != 64 || ((uintptr_t)x % 16))) return hulc_fail(-4, "hulc_spatial_softmax_fwd: an fp16 map has 64 channels, 16-byte aligned");
    const int esz = x_dtype == HULC_F32 ? 4 : 2;
    if (C == 64 && ((uintptr_t)x % (8 * esz)) == 0 && HW >= 32 && HW <= 448 && !getenv("HULC_SSM_ONLINE")) {
        if (x_dtype == HULC_BF16) spatial_softmax_fwd64_regs_kernel<HULC_BF16><<<N, 256, 0, (hipStream_t)stream>>>(x, HW, xmap, ymap, temperature, out, stats);
        else if (x_dtype == HULC_F16) spatial_softmax_fwd64_regs_kernel<HULC_F16><<<N, 256, 0, (hipStream_t)stream>>>(x, HW, xmap, ymap, temperature, out, stats);
        else spatial_softmax_fwd64_regs_kernel<HULC_F32><<<N, 256, 0, (hipStream_t)stream>>>(x, HW, xmap, ymap, temperature, out, stats);
    } else if (C == 64 && ((uintptr_t)x % (8 * esz)) == 0)
        spatial_softmax_fwd64_kernel<<<N, 256, 0, (hipStream_t)stream>>>(x, x_dtype, HW, xmap, ymap, temperature, out, stats);
    else
        spatial_softmax_fwd_kernel<<<N, 256, 0, (hipStream_t)stream>>>(x, x_dtype, N, HW, C, xmap, ymap, temperature, out, stats);
    return hulc_check_launch("hulc_spatial_softmax_fwd");
}

extern "C" int hulc_spatial_softmax_bwd(const void* x, int x_dtype, int N, int HW, int C, const float* xmap, const float* ymap,
                                        const float* temperature, const float* out, const float* stats, const float* dout,
                                        void* dx, int dx_dtype, int relu_mask, void* stream) {
    if (!x || !out || !stats || !dout || !dx) return hulc_fail(-1, "hulc_spatial_softmax_bwd: null pointer");
    if (C > 64 || C <= 0) return hulc_fail(-2, "hulc_spatial_softmax_bwd: C must be in 1..64");
    const int esz = x_dtype == HULC_F32 ? 4 : 2, dsz = dx_dtype == HULC_F32 ? 4 : 2;
    if (C == 64 && ((uintptr_t)x % (8 * esz)) == 0 && ((uintptr_t)dx % (8 * dsz)) == 0)
        spatial_softmax_bwd64_kernel<<<N, 256, 0, (hipStream_t)stream>>>(x, x_dtype, HW, xmap, ymap, temperature, out, stats, dout, dx, dx_dtype,
                                                                      relu_mask);
    else
        spatial_softmax_bwd_kernel<<<N, 256, 0, (hipStream_t)stream>>>(x, x_dtype, N, HW, C, xmap, ymap, temperature, out, stats,
                                                                             dout, dx, dx_dtype, relu_mask);
    return hulc_check_launch("hulc_spatial_softmax_bwd");
}

extern "C" int hulc_layernorm_fwd(const float* x, const float* o, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                                  const float* gamma, const float* beta, float eps, int R, int D, float* pre_out, float* y, float* mean,
                                  float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return hulc_fail(-1, "hulc_layernorm_fwd: null pointer");
    if (D > 256 || D <= 0) return hulc_fail(-2, "hulc_layernorm_fwd: D must be in 1..256");
    if (o && !pre_out) return hulc_fail(-3, "hulc_layernorm_fwd: pre_out required with a residual branch");
    layernorm_fwd_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, o, drop_p, seed, seed_dev, gamma, beta, eps, R, D, pre_out, y, mean, rstd);
    return hulc_check_launch("hulc_layernorm_fwd");
}

extern "C" int hulc_layernorm_slab_fwd(const float* x, const float* o, int n_o, long o_stride, float drop_p, unsigned long long seed,
                                       const unsigned long long* seed_dev, const float* gamma, const float* beta, float eps, int R, int D,
                                       float* pre_out, float* y, float* mean, float* rstd, void* stream) {
    if (!x || !o || !gamma || !beta || !y || !mean || !rstd || !pre_out) return hulc_fail(-1, "hulc_layernorm_slab_fwd: null pointer");
    if (D > 256 || D <= 0 || n_o < 1) return hulc_fail(-2, "hulc_layernorm_slab_fwd: D must be in 1..256, n_o >= 1");
    layernorm_fwd_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, o, drop_p, seed, seed_dev, gamma, beta, eps, R, D, pre_out, y, mean, rstd,
                                                                        n_o, o_stride);
    return hulc_check_launch("hulc_layernorm_slab_fwd");
}

// the same LayerNorm writing its rows ld_y floats apart: several LayerNorms fill disjoint column / row blocks of ONE tensor (the two camera
// encoders' 64 + 64 halves of the perceptual embedding, concat_encoders.py:96-107; the per-modality goal encoders' rows) without a concat copy
extern "C" int hulc_layernorm_fwd_ld(const float* x, const float* gamma, const float* beta, float eps, int R, int D, float* y, long ld_y,
                                     float* mean, float* rstd, void* stream) {
    if (!x || !gamma || !beta || !y || !mean || !rstd) return hulc_fail(-1, "hulc_layernorm_fwd_ld: null pointer");
    if (D > 256 || D <= 0 || ld_y < D) return hulc_fail(-2, "hulc_layernorm_fwd_ld: D must be in 1..256, ld_y >= D");
    layernorm_fwd_kernel<<<(R + 3) / 4, 256, 0, (hipStream_t)stream>>>(x, nullptr, 0.f, 0ull, nullptr, gamma, beta, eps, R, D, nullptr, y, mean, rstd, 1, 0, ld_y);
    return hulc_check_launch("hulc_layernorm_fwd_ld");
}

// rows per workgroup: ~256 workgroups (one per CU) once there are enough rows, never fewer than 4 rows (one per wave)
static int ln_bwd_rows_per_block(int R) { const int r = (R + 255) / 256; return r < 4 ? 4 : r; }
extern "C" long hulc_layernorm_bwd_workspace(int R, int D) {
    const int rpb = ln_bwd_rows_per_block(R);
    return (long)((R + rpb - 1) / rpb) * 2 * D * (long)sizeof(float);
}

extern "C" int hulc_layernorm_bwd(const float* dy, const float* pre, const float* mean, const float* rstd, const float* gamma, int R,
                                  int D, float* dpre, float* do_out, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                                  float* dgamma, float* dbeta, int accumulate_params, void* ws, void* stream) {
    if (!dy || !pre || !mean || !rstd || !gamma || !dpre || !dgamma || !dbeta || !ws) return hulc_fail(-1, "hulc_layernorm_bwd: null pointer");
    if (D > 256 || D <= 0) return hulc_fail(-2, "hulc_layernorm_bwd: D must be in 1..256");
    const int rpb = ln_bwd_rows_per_block(R), nb = (R + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    layernorm_bwd_kernel<<<nb, 256, 0, s>>>(dy, pre, mean, rstd, gamma, R, D, rpb, dpre, do_out, drop_p, seed, seed_dev, (float*)ws);
    // partial rows are [dgamma | dbeta]; each half is summed (fixed order) into its own output
    reduce_rows_wide_kernel<<<dim3((D + 63) / 64, 2), 1024, 0, s>>>((const float*)ws, dgamma, nb, D, 2 * D, accumulate_params, (long)D, dbeta);
    return hulc_check_launch("hulc_layernorm_bwd");
}

// backward of hulc_layernorm_fwd_ld: the incoming gradient is the block of a wider tensor (rows ld_dy floats apart), read in place
extern "C" int hulc_layernorm_bwd_ld(const float* dy, long ld_dy, const float* pre, const float* mean, const float* rstd, const float* gamma, int R,
                                     int D, float* dpre, float* dgamma, float* dbeta, int accumulate_params, void* ws, void* stream) {
    if (!dy || !pre || !mean || !rstd || !gamma || !dpre || !dgamma || !dbeta || !ws) return hulc_fail(-1, "hulc_layernorm_bwd_ld: null pointer");
    if (D > 256 || D <= 0 || ld_dy < D) return hulc_fail(-2, "hulc_layernorm_bwd_ld: D must be in 1..256, ld_dy >= D");
    const int rpb = ln_bwd_rows_per_block(R), nb = (R + rpb - 1) / rpb;
    hipStream_t s = (hipStream_t)stream;
    layernorm_bwd_kernel<<<nb, 256, 0, s>>>(dy, pre, mean, rstd, gamma, R, D, rpb, dpre, nullptr, 0.f, 0ull, nullptr, (float*)ws, ld_dy);
    reduce_rows_wide_kernel<<<dim3((D + 63) / 64, 2), 1024, 0, s>>>((const float*)ws, dgamma, nb, D, 2 * D, accumulate_params, (long)D, dbeta);
    return hulc_check_launch("hulc_layernorm_bwd_ld");
}

// ------------------------------------------------------------------------------------------------
// fan-out of the perceptual embedding emb (N, S, D) to its four consumers in Hulc2.training_step (hulc2.py:380-387 / :228-231): the
// prior sees emb[:, 0], the visual goal encoder emb[:n_last, -1], the posterior all of it, the action decoder the column slice [lo, hi)
// — handed over time-major (S, N, hi - lo), the order the recurrent kernel consumes.  Forward: the three small gathers in one launch.
// Backward: the four gradients merged in one launch (autograd's select / slice backward + three accumulate adds were nine).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emb_fanout_fwd_kernel(const float* __restrict__ emb, int N, int S, int D, int n_last, int lo, int hi,
                                                             float* __restrict__ e0, float* __restrict__ elast, float* __restrict__ edec_t) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int E = hi - lo;
    const long n_dec = (long)S * N * E, n0 = (long)N * D, nl = (long)n_last * D;
    if (i < n_dec) {
        const int c = (int)(i % E); const long sn = i / E; const int n = (int)(sn % N), st = (int)(sn / N);
        edec_t[i] = emb[((long)n * S + st) * D + lo + c];
    } else if (i < n_dec + n0) {
        const long j = i - n_dec; const int c = (int)(j % D), n = (int)(j / D);
        e0[j] = emb[((long)n * S) * D + c];
    } else if (i < n_dec + n0 + nl) {
        const long j = i - n_dec - n0; const int c = (int)(j % D), n = (int)(j / D);
        elast[j] = emb[((long)n * S + S - 1) * D + c];
    }
}
__global__ __launch_bounds__(256) void emb_fanin_bwd_kernel(const float* __restrict__ g_rec, const float* __restrict__ g0, const float* __restrict__ g_last,
                                                            const float* __restrict__ g_dec_t, int N, int S, int D, int n_last, int lo, int hi,
                                                            float* __restrict__ demb) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)N * S * D) return;
    const int c = (int)(i % D); const long ns = i / D; const int st = (int)(ns % S), n = (int)(ns / S);
    float v = g_rec ? g_rec[i] : 0.f;
    if (g0 && st == 0) v += g0[(long)n * D + c];
    if (g_last && st == S - 1 && n < n_last) v += g_last[(long)n * D + c];
    if (g_dec_t && c >= lo && c < hi) v += g_dec_t[((long)st * N + n) * (hi - lo) + c - lo];
    demb[i] = v;
}

extern "C" int hulc_emb_fanout_fwd(const float* emb, int N, int S, int D, int n_last, int lo, int hi, float* e0, float* elast, float* edec_t,
                                   void* stream) {
    if (!emb || !e0 || !edec_t || (n_last > 0 && !elast)) return hulc_fail(-1, "hulc_emb_fanout_fwd: null pointer");
    if (N < 1 || S < 1 || lo < 0 || hi > D || lo >= hi || n_last < 0 || n_last > N) return hulc_fail(-2, "hulc_emb_fanout_fwd: bad geometry");
    const long n = (long)S * N * (hi - lo) + (long)N * D + (long)n_last * D;
    emb_fanout_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(emb, N, S, D, n_last, lo, hi, e0, elast, edec_t);
    return hulc_check_launch("hulc_emb_fanout_fwd");
}
extern "C" int hulc_emb_fanin_bwd(const float* g_rec, const float* g0, const float* g_last, const float* g_dec_t, int N, int S, int D, int n_last,
                                  int lo, int hi, float* demb, void* stream) {
    if (!demb) return hulc_fail(-1, "hulc_emb_fanin_bwd: null pointer");
    if (N < 1 || S < 1 || lo < 0 || hi > D || lo >= hi || n_last < 0 || n_last > N) return hulc_fail(-2, "hulc_emb_fanin_bwd: bad geometry");
    const long n = (long)N * S * D;
    emb_fanin_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(g_rec, g0, g_last, g_dec_t, N, S, D, n_last, lo, hi, demb);
    return hulc_check_launch("hulc_emb_fanin_bwd");
}

// partial (P, 2, D) rows of [dgamma | dbeta] partial sums (hulc_txl_attn_bwd's ln_partial) -> dgamma, dbeta; fixed summation order
extern "C" int hulc_ln_partial_reduce(const float* partial, int P, int D, float* dgamma, float* dbeta, int accumulate, void* stream) {
    if (!partial || !dgamma || !dbeta) return hulc_fail(-1, "hulc_ln_partial_reduce: null pointer");
    if (P < 1 || D < 1) return hulc_fail(-2, "hulc_ln_partial_reduce: bad geometry");
    reduce_rows_wide_kernel<<<dim3((D + 63) / 64, 2), 1024, 0, (hipStream_t)stream>>>(partial, dgamma, P, D, 2 * D, accumulate, (long)D, dbeta);
    return hulc_check_launch("hulc_ln_partial_reduce");
}

namespace {
struct LnMultiP { float* dg[8]; float* db[8]; int acc[8]; };
// blockIdx.z = which LayerNorm; blockIdx.y = dgamma | dbeta; reduce_rows_wide_kernel's summation order
__global__ __launch_bounds__(1024) void ln_partial_multi_kernel(const float* __restrict__ partial, int P, int D, LnMultiP t) {
    __shared__ float red[16][64];
    const int which = blockIdx.z;
    const float* src = partial + (long)which * P * 2 * D + (blockIdx.y ? D : 0);
    float* out = blockIdx.y ? t.db[which] : t.dg[which];
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int rr = blockIdx.x * 64 + lane;
    const int per = (P + 15) / 16;
    const int q0 = sl * per, q1 = q0 + per < P ? q0 + per : P;
    float s = 0.f;
    if (rr < D)
        for (int q = q0; q < q1; ++q) s += src[(long)q * 2 * D + rr];
    red[sl][lane] = s;
    __syncthreads();
    if (sl == 0 && rr < D) {
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) v += red[w][lane];
        out[rr] = t.acc[which] ? out[rr] + v : v;
    }
}
}  // namespace

// see include/hulc2_amd.h
extern "C" int hulc_ln_partial_reduce_multi(const float* partial, int n, int P, int D, float* const* dgamma, float* const* dbeta, const int* accumulate,
                                            void* stream) {
    if (!partial || !dgamma || !dbeta || !accumulate) return hulc_fail(-1, "hulc_ln_partial_reduce_multi: null pointer");
    if (n < 1 || n > 8 || P < 1 || D < 1) return hulc_fail(-2, "hulc_ln_partial_reduce_multi: 1..8 LayerNorms");
    LnMultiP t = {};
    for (int i = 0; i < n; ++i) {
        if (!dgamma[i] || !dbeta[i]) return hulc_fail(-1, "hulc_ln_partial_reduce_multi: null destination");
        t.dg[i] = dgamma[i]; t.db[i] = dbeta[i]; t.acc[i] = accumulate[i];
    }
    ln_partial_multi_kernel<<<dim3((D + 63) / 64, 2, n), 1024, 0, (hipStream_t)stream>>>(partial, P, D, t);
    return hulc_check_launch("hulc_ln_partial_reduce_multi");
}

static long colsum_row_blocks(long M, int N) {
    const long gx = (N + 255) / 256;
    long rb = (1024 + gx - 1) / gx;                     // >= ~1024 workgroups ...
    const long cap = (M + 31) / 32;                     // ... of at least 32 rows each
    if (rb > cap) rb = cap;
    if (rb > 512) rb = 512;
    if (rb < 1) rb = 1;
    return rb;
}
extern "C" long hulc_colsum_workspace(long M, int N) { return colsum_row_blocks(M, N) * N * (long)sizeof(float); }

extern "C" int hulc_colsum(const void* x, int x_dtype, long M, int N, long ld, float* out, int accumulate, void* ws, void* stream) {
    if (!x || !out || !ws) return hulc_fail(-1, "hulc_colsum: null pointer");
    long rb = colsum_row_blocks(M, N);
    const long rpb = (M + rb - 1) / rb;
    rb = (M + rpb - 1) / rpb;
    const int esz = x_dtype == HULC_F32 ? 4 : 2;
    const int vec = (N % 4 == 0) && (ld % 4 == 0) && (((uintptr_t)x) % (4 * esz) == 0);
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((N + 255) / 256, (unsigned)rb);
    colsum_kernel<<<grid, 256, 0, s>>>(x, x_dtype, M, N, ld, rpb, (float*)ws, rb == 1 ? out : nullptr, accumulate, vec);
    if (rb > 1) reduce_rows_wide_kernel<<<(N + 63) / 64, 1024, 0, s>>>((const float*)ws, out, (int)rb, N, N, accumulate);
    return hulc_check_launch("hulc_colsum");
}

extern "C" int hulc_seq_mean_fwd(const float* x, float* y, int B, int S, int D, float scale, void* stream) {
    if (!x || !y) return hulc_fail(-1, "hulc_seq_mean_fwd: null pointer");
    const long n = (long)B * D;
    seq_mean_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, B, S, D, scale);
    return hulc_check_launch("hulc_seq_mean_fwd");
}
extern "C" int hulc_strided_seq_sum(const void* x, int x_dtype, float* y, int B, int S, int D, long stride_b, long stride_s, long ldy,
                                    float scale, void* stream) {
    if (!x || !y) return hulc_fail(-1, "hulc_strided_seq_sum: null pointer");
    const long n = (long)B * D;
    strided_seq_sum_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, x_dtype, y, B, S, D, stride_b, stride_s, ldy, scale);
    return hulc_check_launch("hulc_strided_seq_sum");
}

extern "C" int hulc_seq_mean_bwd(const float* dy, float* dx, int B, int S, int D, void* stream) {
    if (!dy || !dx) return hulc_fail(-1, "hulc_seq_mean_bwd: null pointer");
    const long n = (long)B * S * D;
    seq_mean_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, dx, B, S, D);
    return hulc_check_launch("hulc_seq_mean_bwd");
}
extern "C" int hulc_add_pos_fwd(const float* x, const float* pos, const long* pos_ids, float* y, int B, int S, int D, float drop_p,
                                unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
    if (!x || !pos || !pos_ids || !y) return hulc_fail(-1, "hulc_add_pos_fwd: null pointer");
    const long n = (long)B * S * D;
    add_pos_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, pos, pos_ids, y, B, S, D, drop_p, seed, seed_dev);
    return hulc_check_launch("hulc_add_pos_fwd");
}
extern "C" int hulc_dropout_bwd(const float* dy, float* dx, long n, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                                void* stream) {
    if (!dy || !dx) return hulc_fail(-1, "hulc_dropout_bwd: null pointer");
    dropout_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, dx, n, drop_p, seed, seed_dev);
    return hulc_check_launch("hulc_dropout_bwd");
}

extern "C" int hulc_relu_bwd(const float* dy, const void* y, int y_dtype, float* dx, long n, float scale, void* stream) {
    if (!dy || !y || !dx) return hulc_fail(-1, "hulc_relu_bwd: null pointer");
    relu_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(dy, y, y_dtype, dx, n, scale);
    return hulc_check_launch("hulc_relu_bwd");
}

extern "C" int hulc_attention_fwd(const float* qkv, float* out, float* probs, int B, int S, int H, int head_dim, float drop_p,
                                  unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
    if (!qkv || !out || !probs) return hulc_fail(-1, "hulc_attention_fwd: null pointer");
    if (S > 32 || S <= 0 || head_dim != ATT_DH) return hulc_fail(-2, "hulc_attention_fwd: needs S <= 32 and head_dim == 16");
    attention_fwd_kernel<<<B * H, 64, 0, (hipStream_t)stream>>>(qkv, out, probs, B, S, H, 1.0f / sqrtf((float)head_dim), drop_p, seed, seed_dev);
    return hulc_check_launch("hulc_attention_fwd");
}
extern "C" int hulc_attention_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int B, int S, int H, int head_dim,
                                  float drop_p, unsigned long long seed, const unsigned long long* seed_dev, void* stream) {
    if (!qkv || !probs || !dout || !dqkv) return hulc_fail(-1, "hulc_attention_bwd: null pointer");
    if (S > 32 || S <= 0 || head_dim != ATT_DH) return hulc_fail(-2, "hulc_attention_bwd: needs S <= 32 and head_dim == 16");
    attention_bwd_kernel<<<B * H, 64, 0, (hipStream_t)stream>>>(qkv, probs, dout, dqkv, B, S, H, 1.0f / sqrtf((float)head_dim), drop_p, seed, seed_dev);
    return hulc_check_launch("hulc_attention_bwd");
}
