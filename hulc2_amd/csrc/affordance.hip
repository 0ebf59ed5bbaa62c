// affordance.hip — the pointwise / reduction kernels of the affordance model's trainable part on padded-grid
// activations (layout: gridconv.hip).  Reference: Conv2dReLU's BatchNorm2d (batch statistics) + ReLU and DecoderBlock.forward
// (language fusion, nearest up-sampling, skip concatenation) of hulc2/affordance/models/core/unet_decoder.py:6-80 with FusionMult
// (core/fusion.py:40-47,64-73), cross_entropy_with_logits (utils/losses.py:6-13) as PixelAffLangDetector.criterion applies it
// (pixel_aff_lang_detector.py:122-145), and their autograd backward.
#include "hulc_common.h"
#include "hulc_abi_internal.h"

namespace {

HULC_DEVICE bool grid_interior(int r, int H, int W) {
    const int Wp = W + 2, PP = (H + 2) * Wp;
    const int rem = r % PP, yy = rem / Wp, xx = rem - yy * Wp;
    return yy >= 1 && yy <= H && xx >= 1 && xx <= W;
}

// ---- fixed-order sum of partial rows: out[c] = sum_b part[b][c], one thread per column, four loads in flight ------------------------------
HULC_DEVICE float sum_partials(const float* part, int nb, long ld, int c) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int b = 0;
    for (; b + 3 < nb; b += 4) {
        const float x0 = part[(long)b * ld + c], x1 = part[(long)(b + 1) * ld + c], x2 = part[(long)(b + 2) * ld + c], x3 = part[(long)(b + 3) * ld + c];
        a0 += x0; a1 += x1; a2 += x2; a3 += x3;
    }
    for (; b < nb; ++b) a0 += part[(long)b * ld + c];
    return (a0 + a1) + (a2 + a3);
}

// sums of the two halves of partial rows [nb][2][C] for 64 channels per workgroup: 16 row slices (4 loads in flight each), LDS, fixed order
HULC_DEVICE void sum_partials_2x(const float* part, int nb, int C, int c, bool cok, float (*red)[16][64], float& s1, float& s2) {
    const int lane = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const int per = (nb + 15) / 16, b0 = sl * per, b1 = min(b0 + per, nb);
    float a = 0.f, b = 0.f;
    if (cok && b0 < b1) {
        a = sum_partials(part + (long)b0 * 2 * C, b1 - b0, 2L * C, c);
        b = sum_partials(part + (long)b0 * 2 * C + C, b1 - b0, 2L * C, c);
    }
    red[0][sl][lane] = a; red[1][sl][lane] = b;
    __syncthreads();
    s1 = s2 = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) { s1 += red[0][w][lane]; s2 += red[1][w][lane]; }
}

// BatchNorm2d, training mode: partial sums (sum y, sum y^2 over the interior rows; gridconv's epilogue) -> bn[0..3][C] = mean, rstd,
// scale = gamma rstd, shift = beta - mean scale; running statistics updated with momentum 0.1 and the unbiased variance (nn.BatchNorm2d).
// grid = (channel groups of 64, G row groups): every workgroup sums its share of the partial rows (16 slices x 64 channels, fixed order) into
// mid[g][2][C]; the LAST one to arrive at the channel group's counter adds the G results in order and finalises (G = 1: directly).  A single
// workgroup per channel group took 40-80 us on the 12 700 partial rows of a 224 x 224 layer.
__global__ __launch_bounds__(1024) void grid_bn_finalize_kernel(const float* __restrict__ part, int nb, int C, float count, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, float eps, float momentum, float* __restrict__ bn,
                                                                float* __restrict__ run_mean, float* __restrict__ run_var, float* __restrict__ mid,
                                                                unsigned* __restrict__ ctr) {
    __shared__ float red[2][16][64];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 64 + lane, G = gridDim.y, g = blockIdx.y;
    const bool cok = c < C;
    const int per = (nb + G - 1) / G, b0 = g * per, b1 = min(b0 + per, nb);
    float s1, s2;
    sum_partials_2x(part + (long)b0 * 2 * C, b1 > b0 ? b1 - b0 : 0, C, c, cok, red, s1, s2);
    if (G > 1) {
        if (threadIdx.x < 64 && cok) {
            __hip_atomic_store(mid + ((long)g * 2) * C + c, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mid + ((long)g * 2 + 1) * C + c, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(G - 1);
            if (last) __hip_atomic_store(ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        s1 = s2 = 0.f;
        if (threadIdx.x < 64 && cok)
            for (int q = 0; q < G; ++q) {
                s1 += __hip_atomic_load(mid + ((long)q * 2) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(mid + ((long)q * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
    if (threadIdx.x >= 64 || !cok) return;
    const float mean = s1 / count;
    float var = s2 / count - mean * mean;
    var = var > 0.f ? var : 0.f;
    const float rstd = rsqrtf(var + eps), scale = gamma[c] * rstd;
    bn[c] = mean; bn[C + c] = rstd; bn[2 * C + c] = scale; bn[3 * C + c] = beta[c] - mean * scale;
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * mean;
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * var * (count / (count - 1.f));
}

// out = interior ? relu(y scale + shift) : 0 — a thread handles 8 channels of one grid row
__global__ void grid_bn_relu_fwd_kernel(const uint16_t* __restrict__ y, long ldy, const float* __restrict__ bn, int R, int H, int W, int C,
                                        uint16_t* __restrict__ out, long ldo) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = C / 8;
    if (i >= (long)R * cpr) return;
    const int r = (int)(i / cpr), c0 = (int)(i % cpr) * 8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (grid_interior(r, H, W)) {
        const uint4 v = *(const uint4*)(y + (long)r * ldy + c0);
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
        uint32_t q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = fmaxf(__uint_as_float(w[e] << 16) * bn[2 * C + c0 + 2 * e] + bn[3 * C + c0 + 2 * e], 0.f);
            const float b = fmaxf(__uint_as_float(w[e] & 0xffff0000u) * bn[2 * C + c0 + 2 * e + 1] + bn[3 * C + c0 + 2 * e + 1], 0.f);
            q[e] = pack_bf16x2(a, b);
        }
        o = make_uint4(q[0], q[1], q[2], q[3]);
    }
    *(uint4*)(out + (long)r * ldo + c0) = o;
}

// backward, pass 1: per channel  s1 = sum dY, s2 = sum dY xhat  with dY = dOut where out > 0 (the ReLU), xhat = (y - mean) rstd; a workgroup
// sums 256 grid rows x 64 channels: a thread = 8 channels (16-byte loads) of one of 32 row slots, 8 rows each; partial rows [block][2][C]
__global__ __launch_bounds__(256) void grid_bn_relu_bwd_reduce_kernel(const uint16_t* __restrict__ dout, long ldd, const uint16_t* __restrict__ out, long ldo,
                                                                      const uint16_t* __restrict__ y, long ldy, const float* __restrict__ bn, int R, int C,
                                                                      float* __restrict__ part) {
    __shared__ float red[2][32][64];
    const int ck = threadIdx.x & 7, slot = threadIdx.x >> 3;
    const int c0 = blockIdx.y * 64 + ck * 8;
    const bool cok = c0 < C;                                     // (C = 32: half the chunks idle)
    float mean[8], rstd[8], s1[8], s2[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { mean[e] = cok ? bn[c0 + e] : 0.f; rstd[e] = cok ? bn[C + c0 + e] : 0.f; s1[e] = 0.f; s2[e] = 0.f; }
    const int r0 = blockIdx.x * 256;
    if (cok) {
#pragma unroll 2
        for (int k = slot; k < 256; k += 32) {
            const int r = r0 + k, rc = r < R ? r : R - 1;
            const uint4 vo = *(const uint4*)(out + (long)rc * ldo + c0), vd = *(const uint4*)(dout + (long)rc * ldd + c0), vy = *(const uint4*)(y + (long)rc * ldy + c0);
            const uint32_t wo[4] = {vo.x, vo.y, vo.z, vo.w}, wd[4] = {vd.x, vd.y, vd.z, vd.w}, wy[4] = {vy.x, vy.y, vy.z, vy.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const uint32_t sh = (e & 1) ? 0u : 16u, mk = (e & 1) ? 0xffff0000u : 0xffffffffu;
                const float ov = __uint_as_float((wo[e >> 1] << sh) & (e & 1 ? mk : 0xffff0000u));
                const float dv = __uint_as_float((wd[e >> 1] << sh) & (e & 1 ? mk : 0xffff0000u));
                const float yv = __uint_as_float((wy[e >> 1] << sh) & (e & 1 ? mk : 0xffff0000u));
                const float d = (r < R && ov > 0.f) ? dv : 0.f;     // border rows: out == 0
                s1[e] += d; s2[e] += d * (yv - mean[e]) * rstd[e];
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) { red[0][slot][ck * 8 + e] = s1[e]; red[1][slot][ck * 8 + e] = s2[e]; }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int h = threadIdx.x >> 6, l = threadIdx.x & 63, c = blockIdx.y * 64 + l;
        if (c < C) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 32; ++w) t += red[h][w][l];
            part[((long)blockIdx.x * 2 + h) * C + c] = t;
        }
    }
}

// the two sums, once: sums[0][C] = s1 (= dbeta), sums[1][C] = s2 (= dgamma), also written / accumulated into the parameter gradients;
// spread over G row groups with a last-arriver like grid_bn_finalize_kernel
__global__ __launch_bounds__(1024) void grid_bn_bwd_sums_kernel(const float* __restrict__ part, int nb, int C, float* __restrict__ sums, float* __restrict__ dgamma,
                                                                float* __restrict__ dbeta, int accumulate, float* __restrict__ mid, unsigned* __restrict__ ctr) {
    __shared__ float red[2][16][64];
    __shared__ int s_last;
    const int lane = threadIdx.x & 63, c = blockIdx.x * 64 + lane, G = gridDim.y, g = blockIdx.y;
    const bool cok = c < C;
    const int per = (nb + G - 1) / G, b0 = g * per, b1 = min(b0 + per, nb);
    float s1, s2;
    sum_partials_2x(part + (long)b0 * 2 * C, b1 > b0 ? b1 - b0 : 0, C, c, cok, red, s1, s2);
    if (G > 1) {
        if (threadIdx.x < 64 && cok) {
            __hip_atomic_store(mid + ((long)g * 2) * C + c, s1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(mid + ((long)g * 2 + 1) * C + c, s2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned old = __hip_atomic_fetch_add(ctr + blockIdx.x, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = old == (unsigned)(G - 1);
            if (last) __hip_atomic_store(ctr + blockIdx.x, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = last;
        }
        __syncthreads();
        if (!s_last) return;
        s1 = s2 = 0.f;
        if (threadIdx.x < 64 && cok)
            for (int q = 0; q < G; ++q) {
                s1 += __hip_atomic_load(mid + ((long)q * 2) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s2 += __hip_atomic_load(mid + ((long)q * 2 + 1) * C + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
    }
    if (threadIdx.x >= 64 || !cok) return;
    sums[c] = s1; sums[C + c] = s2;
    if (dbeta) dbeta[c] = accumulate ? dbeta[c] + s1 : s1;
    if (dgamma) dgamma[c] = accumulate ? dgamma[c] + s2 : s2;
}

// backward, pass 2: dZ = interior ? gamma rstd (dY - s1 / M - xhat s2 / M) : 0
__global__ void grid_bn_relu_bwd_apply_kernel(const uint16_t* __restrict__ dout, long ldd, const uint16_t* __restrict__ out, long ldo,
                                              const uint16_t* __restrict__ y, long ldy, const float* __restrict__ bn, const float* __restrict__ sums,
                                              float inv_count, int R, int H, int W, int C, uint16_t* __restrict__ dz, long ldz) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = C / 8;
    if (i >= (long)R * cpr) return;
    const int r = (int)(i / cpr), c0 = (int)(i % cpr) * 8;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (grid_interior(r, H, W)) {
        const uint4 vd = *(const uint4*)(dout + (long)r * ldd + c0), vo = *(const uint4*)(out + (long)r * ldo + c0), vy = *(const uint4*)(y + (long)r * ldy + c0);
        const uint32_t wd[4] = {vd.x, vd.y, vd.z, vd.w}, wo[4] = {vo.x, vo.y, vo.z, vo.w}, wy[4] = {vy.x, vy.y, vy.z, vy.w};
        uint32_t q[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float res[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = c0 + 2 * e + h;
                const float ov = h ? __uint_as_float(wo[e] & 0xffff0000u) : __uint_as_float(wo[e] << 16);
                const float dv = h ? __uint_as_float(wd[e] & 0xffff0000u) : __uint_as_float(wd[e] << 16);
                const float yv = h ? __uint_as_float(wy[e] & 0xffff0000u) : __uint_as_float(wy[e] << 16);
                const float d = ov > 0.f ? dv : 0.f;
                const float xh = (yv - bn[c]) * bn[C + c];
                res[h] = bn[2 * C + c] * (d - sums[c] * inv_count - xh * sums[C + c] * inv_count);
            }
            q[e] = pack_bf16x2(res[0], res[1]);
        }
        o = make_uint4(q[0], q[1], q[2], q[3]);
    }
    *(uint4*)(dz + (long)r * ldz + c0) = o;
}

// ---- DecoderBlock input: [nearest-up-sampled (x * g) | skip] on the output grid --------------------------------------------------------
// x: (N, Hi, Wi, Cx) bf16 with element strides (xsn, xsy, xsx) — a grid tensor's interior or a plain NHWC trunk map; g: (N, Cx) fp32 or null
// (FusionMult: x * lang_proj(l)[:, :, None, None]); skip: (N, Ho, Wo, Cs) bf16 strided likewise or null; out: grid (N, Ho, Wo) rows of Cx + Cs
__global__ void grid_upcat_fwd_kernel(const uint16_t* __restrict__ x, long xsn, long xsy, long xsx, const float* __restrict__ g,
                                      const uint16_t* __restrict__ skip, long ssn, long ssy, long ssx, int N, int Ho, int Wo, int s, int Cx, int Cs,
                                      uint16_t* __restrict__ out) {
    const int Ct = Cx + Cs, cpr = Ct / 8;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Wp = Wo + 2, PP = (Ho + 2) * Wp;
    const long R = (long)N * PP;
    if (i >= R * cpr) return;
    const int r = (int)(i / cpr), c0 = (int)(i % cpr) * 8;
    const int n = r / PP, rem = r - n * PP, yy = rem / Wp, xx = rem - yy * Wp;
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (yy >= 1 && yy <= Ho && xx >= 1 && xx <= Wo) {
        const int yo = yy - 1, xo = xx - 1;
        if (c0 < Cx) {
            const uint4 v = *(const uint4*)(x + (long)n * xsn + (long)(yo / s) * xsy + (long)(xo / s) * xsx + c0);
            if (g) {
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
                uint32_t q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    q[e] = pack_bf16x2(__uint_as_float(w[e] << 16) * g[(long)n * Cx + c0 + 2 * e], __uint_as_float(w[e] & 0xffff0000u) * g[(long)n * Cx + c0 + 2 * e + 1]);
                o = make_uint4(q[0], q[1], q[2], q[3]);
            } else o = v;
        } else {
            o = *(const uint4*)(skip + (long)n * ssn + (long)yo * ssy + (long)xo * ssx + (c0 - Cx));
        }
    }
    *(uint4*)(out + (long)r * Ct + c0) = o;
}

// backward of the x branch: dsmall[n, yi, xi, c] = g[n, c] * sum over the s x s block of dX (grid rows, first Cx of ldd channels), written as a
// grid tensor (N, Hi, Wi) of Cx channels (pixels only: the consumer masks by its own output's border); dg[n, c] = sum over pixels of
// x[n, yi, xi, c] * block sum.  Workgroup = (n, 64 channels, pixel tile pt of npt); a thread = 8 channels (16-byte loads) of one of 32 pixel
// slots; dg (needs npt == 1) is reduced over the slots in a fixed order.
__global__ __launch_bounds__(256) void grid_upcat_bwd_kernel(const uint16_t* __restrict__ dX, long ldd, const uint16_t* __restrict__ x, long xsn, long xsy, long xsx,
                                                             const float* __restrict__ g, int Hi, int Wi, int s, int Cx, uint16_t* __restrict__ dsmall,
                                                             float* __restrict__ dg, int accumulate_dg) {
    __shared__ float red[32][64];
    const int ck = threadIdx.x & 7, slot = threadIdx.x >> 3;                 // 8 chunks x 32 slots
    const int n = blockIdx.x, c0 = blockIdx.y * 64 + ck * 8, pt = blockIdx.z, npt = gridDim.z;
    const int Ho = Hi * s, Wo = Wi * s, Wpo = Wo + 2, PPo = (Ho + 2) * Wpo, Wpi = Wi + 2, PPi = (Hi + 2) * Wpi;
    float gv[8], acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { gv[e] = g ? g[(long)n * Cx + c0 + e] : 1.f; acc[e] = 0.f; }
    const int npx = Hi * Wi, per = (npx + npt - 1) / npt, p0 = pt * per, p1 = min(p0 + per, npx);
    for (int px = p0 + slot; px < p1; px += 32) {
        const int yi = px / Wi, xi = px - yi * Wi;
        float bs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) bs[e] = 0.f;
        for (int a = 0; a < s; ++a)
            for (int b = 0; b < s; ++b) {
                const uint4 v = *(const uint4*)(dX + ((long)n * PPo + (long)(yi * s + a + 1) * Wpo + (xi * s + b + 1)) * ldd + c0);
                const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) { bs[2 * e] += __uint_as_float(w[e] << 16); bs[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u); }
            }
        if (dsmall) {
            uint32_t q[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) q[e] = pack_bf16x2(bs[2 * e] * gv[2 * e], bs[2 * e + 1] * gv[2 * e + 1]);
            *(uint4*)(dsmall + ((long)n * PPi + (long)(yi + 1) * Wpi + (xi + 1)) * Cx + c0) = make_uint4(q[0], q[1], q[2], q[3]);
        }
        if (dg) {
            const uint4 v = *(const uint4*)(x + (long)n * xsn + (long)yi * xsy + (long)xi * xsx + c0);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) { acc[2 * e] += bs[2 * e] * __uint_as_float(w[e] << 16); acc[2 * e + 1] += bs[2 * e + 1] * __uint_as_float(w[e] & 0xffff0000u); }
        }
    }
    if (dg) {
#pragma unroll
        for (int e = 0; e < 8; ++e) red[slot][ck * 8 + e] = acc[e];
        __syncthreads();
        if (threadIdx.x < 64) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < 32; ++w) t += red[w][threadIdx.x];
            float* dst = dg + (long)n * Cx + blockIdx.y * 64 + threadIdx.x;
            *dst = accumulate_dg ? *dst + t : t;
        }
    }
}

// ---- pixel cross-entropy over the H x W logits of an image -----------------------------------------------------------------------------
// logit0: fp32 per grid row (gridconv's out0).  lse[n] = log sum exp over the interior rows; loss = -(1 / (N H W)) sum_n (logit[p0_n] - lse[n])
__global__ __launch_bounds__(1024) void pixel_ce_fwd_kernel(const float* __restrict__ logit0, const int* __restrict__ p0, int H, int W, float* __restrict__ lse,
                                                            float* __restrict__ picked) {
    __shared__ float red[16];
    const int n = blockIdx.x, tid = threadIdx.x, Wp = W + 2, PP = (H + 2) * Wp;
    const float* lg = logit0 + (long)n * PP;
    float m = -3.0e38f;
    for (int i = tid; i < H * W; i += 1024) { const int y = i / W, x = i - y * W; m = fmaxf(m, lg[(y + 1) * Wp + x + 1]); }
    m = wave_max(m);
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    float mm = red[0];
#pragma unroll
    for (int w = 1; w < 16; ++w) mm = fmaxf(mm, red[w]);
    __syncthreads();
    float s = 0.f;
    for (int i = tid; i < H * W; i += 1024) { const int y = i / W, x = i - y * W; s += __expf(lg[(y + 1) * Wp + x + 1] - mm); }
    s = wave_sum(s);
    if ((tid & 63) == 0) red[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
        float t = 0.f;
#pragma unroll
        for (int w = 0; w < 16; ++w) t += red[w];
        lse[n] = mm + logf(t);
        picked[n] = lg[(p0[2 * n] + 1) * Wp + p0[2 * n + 1] + 1];
    }
}
// dlogit = upstream * (softmax - onehot) / (N H W) into channel 0 of a grid tensor of C channels (the others and the borders zero)
__global__ void pixel_ce_bwd_kernel(const float* __restrict__ logit0, const int* __restrict__ p0, const float* __restrict__ lse, const float* __restrict__ upstream,
                                    int N, int H, int W, int C, uint16_t* __restrict__ dz) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Wp = W + 2, PP = (H + 2) * Wp, cpr = C / 8;
    if (i >= (long)N * PP * cpr) return;
    const int r = (int)(i / cpr), ck = (int)(i % cpr);
    uint4 o = make_uint4(0u, 0u, 0u, 0u);
    if (ck == 0) {
        const int n = r / PP, rem = r - n * PP, yy = rem / Wp, xx = rem - yy * Wp;
        if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
            float d = __expf(logit0[r] - lse[n]);
            if (yy - 1 == p0[2 * n] && xx - 1 == p0[2 * n + 1]) d -= 1.f;
            d *= upstream[0] / ((float)N * H * W);
            o.x = pack_bf16x2(d, 0.f);
        }
    }
    *(uint4*)(dz + (long)r * C + ck * 8) = o;
}

// ---- the one-channel segmentation head (r3m_rn18.py:64-69: nn.Conv2d(32, 1, 3, padding = 1)) without matrix cores --------------------------
// As a 32 -> 32 gridconv with 31 zero output channels the head and its two gradients moved ~100 MB each for 9 x 32 useful weights and took
// 68 + 68 + 85 us at 32 images; written directly they are streaming kernels: a thread owns 8 channels of a grid row (4 threads per row,
// 16 rows = 1 KB per wave load), the 9 x C weights sit in LDS as fp32.
HULC_DEVICE bool grid_interior(int r, int R, int H, int W) {
    const int Wp = W + 2, PP = (H + 2) * Wp;
    const int rem = r % PP, yy = rem / Wp, xx = rem - yy * Wp;
    return r < R && yy >= 1 && yy <= H && xx >= 1 && xx <= W;
}
HULC_DEVICE void bf16x8_to_f32(const uint4& v, float (&f)[8]) {
    f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u); f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
    f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u); f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}
// out0[r] = bias + sum_{t, ci} x[r + off_t][ci] w[ci][t] on the pixels, 0 on the border; w = the parameter (1, C, 3, 3) fp32
template <int CPR>      // 16-byte chunks per row = threads per row (C = 8 CPR)
__global__ __launch_bounds__(256) void head_conv_fwd_kernel(const uint16_t* __restrict__ x, long ldx, const float* __restrict__ w, const float* __restrict__ bias,
                                                            int R, int H, int W, float* __restrict__ out0) {
    constexpr int C = CPR * 8;
    __shared__ float wl[9 * C];                                     // [tap][ci]
    for (int i = threadIdx.x; i < 9 * C; i += 256) wl[i] = w[(i % C) * 9 + i / C];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(i / CPR), ck = (int)(i % CPR), Wp = W + 2;
    const bool ok = r < R && grid_interior(r, R, H, W);
    float acc = 0.f;
    if (ok) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const uint4 v = *(const uint4*)(x + (long)(r + (t / 3 - 1) * Wp + (t % 3 - 1)) * ldx + ck * 8);
            float f[8];
            bf16x8_to_f32(v, f);
#pragma unroll
            for (int j = 0; j < 8; ++j) acc += f[j] * wl[t * C + ck * 8 + j];
        }
    }
#pragma unroll
    for (int o = 1; o < CPR; o <<= 1) acc += __shfl_xor(acc, o, 64);
    if (ck == 0 && r < R) out0[r] = ok ? acc + bias[0] : 0.f;
}
// dx[r][ci] = sum_t g[r - off_t] w[ci][t] on the pixels, 0 on the border (g fp32 per grid row, zero on the border rows)
template <int CPR>
__global__ __launch_bounds__(256) void head_conv_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ w, int R, int H, int W,
                                                              uint16_t* __restrict__ dx, long lddx) {
    constexpr int C = CPR * 8;
    __shared__ float wl[9 * C];
    for (int i = threadIdx.x; i < 9 * C; i += 256) wl[i] = w[(i % C) * 9 + i / C];
    __syncthreads();
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int r = (int)(i / CPR), ck = (int)(i % CPR), Wp = W + 2;
    if (r >= R) return;
    float a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = 0.f;
    if (grid_interior(r, R, H, W)) {
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float gv = g[r - ((t / 3 - 1) * Wp + (t % 3 - 1))];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] += gv * wl[t * C + ck * 8 + j];
        }
    }
    *(uint4*)(dx + (long)r * lddx + ck * 8) = make_uint4(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4], a[5]), pack_bf16x2(a[6], a[7]));
}
// partial[blk][t][ci] = sum over the block's rows r of x[r][ci] g[r - off_t]  (= the rows' share of dW[ci][t] = sum_r g[r] x[r + off_t][ci]);
// every x row is read once
template <int CPR>
__global__ __launch_bounds__(256) void head_conv_wgrad_kernel(const uint16_t* __restrict__ x, long ldx, const float* __restrict__ g, int R, int H, int W,
                                                              int rows_per_blk, float* __restrict__ partial) {
    constexpr int C = CPR * 8, RPP = 256 / CPR;
    __shared__ float red[4][9 * C];
    const int tid = threadIdx.x, ck = tid % CPR, slot = tid / CPR, Wp = W + 2;
    const int rbeg = blockIdx.x * rows_per_blk, rend = min(rbeg + rows_per_blk, R);
    float acc[9][8];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[t][j] = 0.f;
    // four rows per trip: four independent 16-byte loads in flight per thread (one at a time left the kernel latency-bound at 2 TB/s)
    for (int r = rbeg + slot; r < rend; r += 4 * RPP) {
        uint4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] = *(const uint4*)(x + (long)min(r + k * RPP, rend - 1) * ldx + ck * 8);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int rr = r + k * RPP;
            const bool live = rr < rend;
            float f[8];
            bf16x8_to_f32(v[k], f);
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const int q = rr - ((t / 3 - 1) * Wp + (t % 3 - 1));
                const float gv = (live && q >= 0 && q < R) ? g[q] : 0.f;
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[t][j] += gv * f[j];
            }
        }
    }
    // lanes with the same chunk (lane % CPR) add up, then the four waves, in a fixed order
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = acc[t][j];
#pragma unroll
            for (int o = CPR; o < 64; o <<= 1) v += __shfl_xor(v, o, 64);
            acc[t][j] = v;
        }
    if ((tid & 63) < CPR) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int j = 0; j < 8; ++j) red[tid >> 6][t * C + ck * 8 + j] = acc[t][j];
    }
    __syncthreads();
    for (int i = tid; i < 9 * C; i += 256) partial[(long)blockIdx.x * 9 * C + i] = ((red[0][i] + red[1][i]) + red[2][i]) + red[3][i];
}
// dw[ci][t] (+)= sum over the blocks: 32 slices of blocks per output, then the slices (fixed order); one workgroup per 8 outputs
__global__ __launch_bounds__(256) void head_conv_wgrad_final_kernel(const float* __restrict__ partial, int nblk, int C, float* __restrict__ dw, int accumulate) {
    __shared__ float red[32][8];
    const int o = threadIdx.x & 7, sl = threadIdx.x >> 3, idx = blockIdx.x * 8 + o;
    float a = 0.f;
    if (idx < 9 * C)
        for (int b = sl; b < nblk; b += 32) a += partial[(long)b * 9 * C + idx];
    red[sl][o] = a;
    __syncthreads();
    if (sl == 0 && idx < 9 * C) {
        float v = 0.f;
#pragma unroll
        for (int k = 0; k < 32; ++k) v += red[k][o];
        float* dst = dw + (idx % C) * 9 + idx / C;                  // partial is [tap][ci], the parameter (1, C, 3, 3)
        *dst = accumulate ? *dst + v : v;
    }
}
// d logit = upstream (softmax - onehot) / (N H W) as fp32 per grid row, zero on the border
__global__ void pixel_ce_bwd_rows_kernel(const float* __restrict__ logit0, const int* __restrict__ p0, const float* __restrict__ lse, const float* __restrict__ upstream,
                                         int N, int H, int W, float* __restrict__ g) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int Wp = W + 2, PP = (H + 2) * Wp;
    if (i >= (long)N * PP) return;
    const int r = (int)i, n = r / PP, rem = r - n * PP, yy = rem / Wp, xx = rem - yy * Wp;
    float d = 0.f;
    if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) {
        d = __expf(logit0[r] - lse[n]);
        if (yy - 1 == p0[2 * n] && xx - 1 == p0[2 * n + 1]) d -= 1.f;
        d *= upstream[0] / ((float)N * H * W);
    }
    g[r] = d;
}

// a dense NHWC bf16 map -> the padded grid (pixels copied, border rows zero): entry of the frozen trunk's stride-1 stages into hulc_gridconv3x3_fused
__global__ void grid_from_nhwc_kernel(const uint16_t* __restrict__ x, int R, int H, int W, int C, uint16_t* __restrict__ y, long ldy) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int cpr = C / 8, Wp = W + 2, PP = (H + 2) * Wp;
    if (i >= (long)R * cpr) return;
    const int r = (int)(i / cpr), ck = (int)(i % cpr);
    const int n = r / PP, rem = r - n * PP, yy = rem / Wp, xx = rem - yy * Wp;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (yy >= 1 && yy <= H && xx >= 1 && xx <= W) v = *(const uint4*)(x + (((long)n * H + (yy - 1)) * W + (xx - 1)) * C + ck * 8);
    *(uint4*)(y + (long)r * ldy + ck * 8) = v;
}

// ---- depth head tail: the two 256 -> 1 heads, sigma = exp(clamp(., -20, 2)) and nn.GaussianNLLLoss (depth_gaussian.py:67-69,94-102) ------------
// one workgroup (B rows of D features, a few KB): replaces ~35 framework launches of the forward + backward pass by two
__global__ __launch_bounds__(256) void depth_nll_fwd_kernel(const float* __restrict__ x, int B, int D, const float* __restrict__ wmu, const float* __restrict__ bmu,
                                                            const float* __restrict__ ws, const float* __restrict__ bs, const float* __restrict__ target,
                                                            float* __restrict__ mu, float* __restrict__ sigma, float* __restrict__ logsig, float* __restrict__ loss) {
    __shared__ float part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float acc = 0.f;
    for (int b = wave; b < B; b += 4) {
        float a = 0.f, c = 0.f;
        for (int d = lane; d < D; d += 64) { const float xv = x[(long)b * D + d]; a += xv * wmu[d]; c += xv * ws[d]; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
        const float m = a + bmu[0], ls = c + bs[0];
        const float sg = expf(fminf(fmaxf(ls, -20.f), 2.f));
        const float var = fmaxf(sg, 1e-6f), r = m - target[b];
        if (lane == 0) { mu[b] = m; sigma[b] = sg; logsig[b] = ls; }
        acc += 0.5f * (logf(var) + r * r / var);
    }
    if (lane == 0) part[wave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) loss[0] = (((part[0] + part[1]) + part[2]) + part[3]) / (float)B;
}

__global__ __launch_bounds__(256) void depth_nll_bwd_kernel(const float* __restrict__ x, int B, int D, const float* __restrict__ wmu, const float* __restrict__ ws,
                                                            const float* __restrict__ mu, const float* __restrict__ sigma, const float* __restrict__ logsig,
                                                            const float* __restrict__ target, const float* __restrict__ gout, float* __restrict__ dx,
                                                            float* __restrict__ dwmu, float* __restrict__ dbmu, float* __restrict__ dws, float* __restrict__ dbs, int accmask) {
    extern __shared__ float sm[];                                  // dmu[B], dls[B]
    float* dm = sm;
    float* dl = sm + B;
    const int tid = threadIdx.x;
    const float g = gout[0] / (float)B;
    for (int b = tid; b < B; b += 256) {
        const float sg = sigma[b], var = fmaxf(sg, 1e-6f), r = mu[b] - target[b], ls = logsig[b];
        const float dvar = g * 0.5f * (1.f / var - r * r / (var * var));
        dm[b] = g * r / var;
        dl[b] = (sg >= 1e-6f && ls >= -20.f && ls <= 2.f) ? dvar * sg : 0.f;       // clamp(min) and clamp(-20, 2) pass the gradient inside their ranges
    }
    __syncthreads();
    if (dx)
        for (long i = tid; i < (long)B * D; i += 256) { const int b = (int)(i / D), d = (int)(i % D); dx[i] = dm[b] * wmu[d] + dl[b] * ws[d]; }
    for (int d = tid; d < D; d += 256) {
        float a = 0.f, c = 0.f;
        for (int b = 0; b < B; ++b) { const float xv = x[(long)b * D + d]; a += dm[b] * xv; c += dl[b] * xv; }
        dwmu[d] = (accmask & 1) ? dwmu[d] + a : a;
        dws[d] = (accmask & 4) ? dws[d] + c : c;
    }
    if (tid == 0) {
        float a = 0.f, c = 0.f;
        for (int b = 0; b < B; ++b) { a += dm[b]; c += dl[b]; }
        dbmu[0] = (accmask & 2) ? dbmu[0] + a : a;
        dbs[0] = (accmask & 8) ? dbs[0] + c : c;
    }
}

}  // namespace

// ---- C ABI (include/hulc2_amd.h) ---------------------------------------------------------------------------------------------------------
extern "C" int hulc_grid_bn_finalize(const float* part, int nb, int C, long count, const float* gamma, const float* beta, float eps, float momentum,
                                     float* bn, float* run_mean, float* run_var, float* mid, unsigned* counters, void* stream) {
    if (!part || !gamma || !beta || !bn || C <= 0 || nb <= 0 || count < 2) return hulc_fail(-1, "hulc_grid_bn_finalize: bad argument");
    int G = 1;
    if (mid && counters) { while (G < 32 && nb / (G * 2) >= 64) G *= 2; }      // >= 64 partial rows per row group
    grid_bn_finalize_kernel<<<dim3((C + 63) / 64, G), 1024, 0, (hipStream_t)stream>>>(part, nb, C, (float)count, gamma, beta, eps, momentum, bn, run_mean, run_var,
                                                                                      mid, counters);
    return hulc_check_launch("hulc_grid_bn_finalize");
}

extern "C" int hulc_grid_bn_relu_fwd(const void* y, long ldy, const float* bn, int N, int H, int W, int C, void* out, long ldo, void* stream) {
    if (!y || !bn || !out || C % 8 || ldy % 8 || ldo % 8) return hulc_fail(-1, "hulc_grid_bn_relu_fwd: bad argument");
    const long R = (long)N * (H + 2) * (W + 2), n = R * (C / 8);
    grid_bn_relu_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const uint16_t*)y, ldy, bn, (int)R, H, W, C, (uint16_t*)out, ldo);
    return hulc_check_launch("hulc_grid_bn_relu_fwd");
}

extern "C" long hulc_grid_bn_bwd_workspace(int N, int H, int W, int C) {
    const long R = (long)N * (H + 2) * (W + 2);
    return (((R + 255) / 256) * 2 * C + 2 * C + 64 * C) * (long)sizeof(float);      // partial rows, the two sums, the row groups' intermediate sums
}

extern "C" int hulc_grid_bn_relu_bwd(const void* dout, long ldd, const void* out, long ldo, const void* y, long ldy, const float* bn, int N, int H, int W,
                                     int C, void* dz, long ldz, float* dgamma, float* dbeta, int accumulate_params, void* ws, unsigned* counters, void* stream) {
    if (!dout || !out || !y || !bn || !dz || !ws || C % 8) return hulc_fail(-1, "hulc_grid_bn_relu_bwd: bad argument (C must be a multiple of 8)");
    const long R = (long)N * (H + 2) * (W + 2);
    const int nb = (int)((R + 255) / 256);
    float* part = (float*)ws;
    float* sums = part + (long)nb * 2 * C;
    float* mid = sums + 2 * C;
    int G = 1;
    if (counters) { while (G < 32 && nb / (G * 2) >= 64) G *= 2; }
    hipStream_t s = (hipStream_t)stream;
    grid_bn_relu_bwd_reduce_kernel<<<dim3(nb, (C + 63) / 64), 256, 0, s>>>((const uint16_t*)dout, ldd, (const uint16_t*)out, ldo, (const uint16_t*)y, ldy, bn, (int)R, C, part);
    grid_bn_bwd_sums_kernel<<<dim3((C + 63) / 64, G), 1024, 0, s>>>(part, nb, C, sums, dgamma, dbeta, accumulate_params, mid, counters);
    const long n = R * (C / 8);
    grid_bn_relu_bwd_apply_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((const uint16_t*)dout, ldd, (const uint16_t*)out, ldo, (const uint16_t*)y, ldy, bn, sums,
                                                                             1.0f / ((float)N * H * W), (int)R, H, W, C, (uint16_t*)dz, ldz);
    return hulc_check_launch("hulc_grid_bn_relu_bwd");
}

extern "C" int hulc_grid_upcat_fwd(const void* x, long xsn, long xsy, long xsx, const float* g, const void* skip, long ssn, long ssy, long ssx, int N, int Ho,
                                   int Wo, int s, int Cx, int Cs, void* out, void* stream) {
    if (!x || !out || Cx % 8 || Cs % 8 || s < 1 || Ho % s || Wo % s || (Cs && !skip)) return hulc_fail(-1, "hulc_grid_upcat_fwd: bad argument");
    const long n = (long)N * (Ho + 2) * (Wo + 2) * ((Cx + Cs) / 8);
    grid_upcat_fwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const uint16_t*)x, xsn, xsy, xsx, g, (const uint16_t*)skip, ssn, ssy, ssx, N, Ho,
                                                                                        Wo, s, Cx, Cs, (uint16_t*)out);
    return hulc_check_launch("hulc_grid_upcat_fwd");
}

extern "C" int hulc_grid_upcat_bwd(const void* dX, long ldd, const void* x, long xsn, long xsy, long xsx, const float* g, int N, int Hi, int Wi, int s, int Cx,
                                   void* dsmall, float* dg, int accumulate_dg, void* stream) {
    if (!dX || Cx % 64 || (dg && !x) || (!dsmall && !dg)) return hulc_fail(-1, "hulc_grid_upcat_bwd: bad argument (Cx must be a multiple of 64)");
    int npt = 1;
    if (!dg) { while ((long)N * (Cx / 64) * npt < 512 && npt * 64 < Hi * Wi) npt *= 2; }
    grid_upcat_bwd_kernel<<<dim3(N, Cx / 64, npt), 256, 0, (hipStream_t)stream>>>((const uint16_t*)dX, ldd, (const uint16_t*)x, xsn, xsy, xsx, g, Hi, Wi, s, Cx,
                                                                                  (uint16_t*)dsmall, dg, accumulate_dg);
    return hulc_check_launch("hulc_grid_upcat_bwd");
}

extern "C" int hulc_pixel_ce_fwd(const float* logit0, const int* p0, int N, int H, int W, float* lse, float* picked, void* stream) {
    if (!logit0 || !p0 || !lse || !picked) return hulc_fail(-1, "hulc_pixel_ce_fwd: null pointer");
    pixel_ce_fwd_kernel<<<N, 1024, 0, (hipStream_t)stream>>>(logit0, p0, H, W, lse, picked);
    return hulc_check_launch("hulc_pixel_ce_fwd");
}

extern "C" int hulc_pixel_ce_bwd(const float* logit0, const int* p0, const float* lse, const float* upstream, int N, int H, int W, int C, void* dz, void* stream) {
    if (!logit0 || !p0 || !lse || !upstream || !dz || C % 8) return hulc_fail(-1, "hulc_pixel_ce_bwd: bad argument");
    const long n = (long)N * (H + 2) * (W + 2) * (C / 8);
    pixel_ce_bwd_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(logit0, p0, lse, upstream, N, H, W, C, (uint16_t*)dz);
    return hulc_check_launch("hulc_pixel_ce_bwd");
}

extern "C" int hulc_depth_nll_fwd(const float* x, int B, int D, const float* w_mu, const float* b_mu, const float* w_sigma, const float* b_sigma, const float* target,
                                  float* mu, float* sigma, float* log_sigma, float* loss, void* stream) {
    if (!x || !w_mu || !b_mu || !w_sigma || !b_sigma || !target || !mu || !sigma || !log_sigma || !loss) return hulc_fail(-1, "hulc_depth_nll_fwd: null pointer");
    if (B <= 0 || D <= 0) return hulc_fail(-2, "hulc_depth_nll_fwd: B and D must be positive");
    depth_nll_fwd_kernel<<<1, 256, 0, (hipStream_t)stream>>>(x, B, D, w_mu, b_mu, w_sigma, b_sigma, target, mu, sigma, log_sigma, loss);
    return hulc_check_launch("hulc_depth_nll_fwd");
}

extern "C" int hulc_depth_nll_bwd(const float* x, int B, int D, const float* w_mu, const float* w_sigma, const float* mu, const float* sigma, const float* log_sigma,
                                  const float* target, const float* gout, float* dx, float* dw_mu, float* db_mu, float* dw_sigma, float* db_sigma, int accumulate_mask,
                                  void* stream) {
    if (!x || !w_mu || !w_sigma || !mu || !sigma || !log_sigma || !target || !gout || !dw_mu || !db_mu || !dw_sigma || !db_sigma) return hulc_fail(-1, "hulc_depth_nll_bwd: null pointer");
    if (B <= 0 || D <= 0 || B > 4096) return hulc_fail(-2, "hulc_depth_nll_bwd: 1 <= B <= 4096");
    depth_nll_bwd_kernel<<<1, 256, 2 * B * sizeof(float), (hipStream_t)stream>>>(x, B, D, w_mu, w_sigma, mu, sigma, log_sigma, target, gout, dx, dw_mu, db_mu, dw_sigma,
                                                                                 db_sigma, accumulate_mask);
    return hulc_check_launch("hulc_depth_nll_bwd");
}

#define HEAD_DISPATCH(C_, CALL)                                                                                        \
    switch ((C_) / 8) {                                                                                                \
    case 1: { constexpr int CPR = 1; CALL; } break;                                                                    \
    case 2: { constexpr int CPR = 2; CALL; } break;                                                                    \
    case 4: { constexpr int CPR = 4; CALL; } break;                                                                    \
    case 8: { constexpr int CPR = 8; CALL; } break;                                                                    \
    default: return hulc_fail(-2, "hulc_head_conv: C must be 8, 16, 32 or 64");                                        \
    }

extern "C" int hulc_head_conv_fwd(const void* x, long ldx, const float* w, const float* bias, int N, int H, int W, int C, float* out0, void* stream) {
    if (!x || !w || !bias || !out0 || ldx % 8 || (uintptr_t)x % 16 || C % 8) return hulc_fail(-1, "hulc_head_conv_fwd: bad argument");
    const long R = (long)N * (H + 2) * (W + 2);
    const unsigned nb = (unsigned)((R * (C / 8) + 255) / 256);
    HEAD_DISPATCH(C, (head_conv_fwd_kernel<CPR><<<nb, 256, 0, (hipStream_t)stream>>>((const uint16_t*)x, ldx, w, bias, (int)R, H, W, out0)))
    return hulc_check_launch("hulc_head_conv_fwd");
}

extern "C" int hulc_head_conv_dgrad(const float* g, const float* w, int N, int H, int W, int C, void* dx, long lddx, void* stream) {
    if (!g || !w || !dx || lddx % 8 || (uintptr_t)dx % 16 || C % 8) return hulc_fail(-1, "hulc_head_conv_dgrad: bad argument");
    const long R = (long)N * (H + 2) * (W + 2);
    const unsigned nb = (unsigned)((R * (C / 8) + 255) / 256);
    HEAD_DISPATCH(C, (head_conv_dgrad_kernel<CPR><<<nb, 256, 0, (hipStream_t)stream>>>(g, w, (int)R, H, W, (uint16_t*)dx, lddx)))
    return hulc_check_launch("hulc_head_conv_dgrad");
}

static int head_wgrad_blocks(long R, int C) {
    const int rpp = 256 / (C / 8);                                  // rows per pass of a workgroup
    long nb = (R + 8L * rpp - 1) / (8L * rpp);                      // >= 8 passes per workgroup ...
    return (int)(nb > 512 ? 512 : (nb < 1 ? 1 : nb));              // ... and at most 512 of them
}
extern "C" long hulc_head_conv_wgrad_workspace(int N, int H, int W, int C) {
    return (long)head_wgrad_blocks((long)N * (H + 2) * (W + 2), C) * 9 * C * (long)sizeof(float);
}
extern "C" int hulc_head_conv_wgrad(const void* x, long ldx, const float* g, int N, int H, int W, int C, float* dw, int accumulate, void* ws, void* stream) {
    if (!x || !g || !dw || !ws || ldx % 8 || (uintptr_t)x % 16 || C % 8) return hulc_fail(-1, "hulc_head_conv_wgrad: bad argument");
    const long R = (long)N * (H + 2) * (W + 2);
    const int nb = head_wgrad_blocks(R, C);
    const int rows = (int)((R + nb - 1) / nb);
    HEAD_DISPATCH(C, (head_conv_wgrad_kernel<CPR><<<nb, 256, 0, (hipStream_t)stream>>>((const uint16_t*)x, ldx, g, (int)R, H, W, rows, (float*)ws)))
    head_conv_wgrad_final_kernel<<<(9 * C + 7) / 8, 256, 0, (hipStream_t)stream>>>((const float*)ws, nb, C, dw, accumulate);
    return hulc_check_launch("hulc_head_conv_wgrad");
}
#undef HEAD_DISPATCH

extern "C" int hulc_pixel_ce_bwd_rows(const float* logit0, const int* p0, const float* lse, const float* upstream, int N, int H, int W, float* g, void* stream) {
    if (!logit0 || !p0 || !lse || !upstream || !g) return hulc_fail(-1, "hulc_pixel_ce_bwd_rows: null pointer");
    const long n = (long)N * (H + 2) * (W + 2);
    pixel_ce_bwd_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(logit0, p0, lse, upstream, N, H, W, g);
    return hulc_check_launch("hulc_pixel_ce_bwd_rows");
}

extern "C" int hulc_grid_from_nhwc(const void* x, int N, int H, int W, int C, void* y, long ldy, void* stream) {
    if (!x || !y || C % 8 || ldy % 8 || ldy < C || ((uintptr_t)x | (uintptr_t)y) % 16) return hulc_fail(-1, "hulc_grid_from_nhwc: bad argument");
    const long R = (long)N * (H + 2) * (W + 2);
    const long n = R * (C / 8);
    grid_from_nhwc_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>((const uint16_t*)x, (int)R, H, W, C, (uint16_t*)y, ldy);
    return hulc_check_launch("hulc_grid_from_nhwc");
}
