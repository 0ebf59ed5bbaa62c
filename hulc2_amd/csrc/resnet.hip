// resnet.hip — the pieces of the frozen ResNet-18 trunk of VisionR3M that are not convolutions.
//
// reference behaviour: hulc2/models/perceptual_encoders/vision_r3m.py:8-32 runs `self.r3m(x)` under no_grad on frames in [0, 255]
// (conf/datamodule/transforms/real_world_r3m.yaml:2-13).  `r3m` is an un-vendored submodule (parity unpinned); its public
// forward is obs / 255 -> Normalize(ImageNet mean, std) -> torchvision resnet18 with fc = Identity.  The convolutions (BatchNorm folded
// into weights + bias, residual add and ReLU in the epilogue) are hulc_conv2d_padded_fwd (conv.hip); this file holds the input
// normalisation and the stem's max pool.  The global average pool is hulc_strided_seq_sum over the (H*W) axis.
#include "hulc_common.h"
#include "hulc_abi_internal.h"

namespace {

// x fp32 NCHW [N][3][H][W] in [0, 255] -> y NHWC [N][H][W][8]: channel c < 3 = (x / 255 - mean[c]) * inv_std[c], channels 3..7 zero
// (8 channels make a pixel one aligned 16-byte bf16 chunk — the gather unit of the conv kernel; the stem's weights are zero there)
__global__ __launch_bounds__(256) void r3m_normalize_kernel(const float* __restrict__ x, long HW, long total, float m0, float m1, float m2, float s0,
                                                            float s1, float s2, void* __restrict__ y, int y_dtype) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // pixel index over N*H*W
    if (i >= total) return;
    const long n = i / HW, p = i - n * HW;
    const float* px = x + n * 3 * HW + p;
    const float a = (px[0] / 255.f - m0) * s0, b = (px[HW] / 255.f - m1) * s1, c = (px[2 * HW] / 255.f - m2) * s2;
    if (y_dtype == HULC_BF16) {
        ((uint4*)y)[i] = make_uint4(pack_bf16x2(a, b), pack_bf16x2(c, 0.f), 0u, 0u);
    } else {
        float4* o = (float4*)y + 2 * i;
        o[0] = make_float4(a, b, c, 0.f); o[1] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// the same normalisation into the packed stem input: xp bf16 [N][H+6][Wp][4], pixel (y, x) at [y+3][x+3] = (R, G, B, 0), zero border
__global__ __launch_bounds__(256) void r3m_normalize_packed_kernel(const float* __restrict__ x, int H, int W, int Wp, long total, float m0, float m1,
                                                                   float m2, float s0, float s1, float s2, uint2* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // padded pixel index over N*(H+6)*Wp
    if (i >= total) return;
    const int xp = (int)(i % Wp); const long r = i / Wp;
    const int yp = (int)(r % (H + 6)); const long n = r / (H + 6);
    const int yy = yp - 3, xx = xp - 3;
    uint2 o = make_uint2(0u, 0u);
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
        const long HW = (long)H * W;
        const float* px = x + n * 3 * HW + (long)yy * W + xx;
        o = make_uint2(pack_bf16x2((px[0] / 255.f - m0) * s0, (px[HW] / 255.f - m1) * s1), pack_bf16x2((px[2 * HW] / 255.f - m2) * s2, 0.f));
    }
    y[i] = o;
}

// max pool k x k, stride s, padding p (padded positions never win: nn.MaxPool2d pads with -inf), NHWC, 8 channels per thread
__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const void* __restrict__ x, int dtype, int H, int W, int C8, int OH, int OW, int k, int s,
                                                           int pad, long total, void* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over N*OH*OW*C8
    if (i >= total) return;
    const int c8 = (int)(i % C8); long r = i / C8;
    const int ox = (int)(r % OW); r /= OW;
    const int oy = (int)(r % OH); const long n = r / OH;
    float best[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = -INFINITY;
    for (int ky = 0; ky < k; ++ky) {
        const int iy = oy * s - pad + ky;
        if (iy < 0 || iy >= H) continue;
        for (int kx = 0; kx < k; ++kx) {
            const int ix = ox * s - pad + kx;
            if (ix < 0 || ix >= W) continue;
            Chunk8 c;
            chunk_load_contig(c, x, dtype, ((n * H + iy) * W + ix) * (long)C8 * 8 + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) best[e] = fmaxf(best[e], c.v[e]);
        }
    }
    const long o = i * 8;
    if (dtype == HULC_BF16) {
        *(uint4*)((uint16_t*)y + o) = make_uint4(pack_bf16x2(best[0], best[1]), pack_bf16x2(best[2], best[3]), pack_bf16x2(best[4], best[5]),
                                                 pack_bf16x2(best[6], best[7]));
    } else {
        float4* q = (float4*)((float*)y + o);
        q[0] = make_float4(best[0], best[1], best[2], best[3]); q[1] = make_float4(best[4], best[5], best[6], best[7]);
    }
}

// ---- BatchNorm2d on BATCH statistics over NHWC rows (round 4: the affordance trunk "as the reference runs it") --------------------------------
// hulc2/affordance/models/visual_lang_encoders/r3m_rn18.py:27-43 freezes the PARAMETERS of layer1..layer4 and
// pixel_aff_lang_detector.py:51-53 never puts the trunk into eval mode: under Lightning's train() its nn.BatchNorm2d layers normalise with
// the statistics of the batch (biased variance) and keep updating running_mean / running_var (momentum 0.1, unbiased variance).  Three
// launches per layer behind the (unfolded, bias-free) convolution: partial sums, finalize, apply.
//
// partial[p][0][c] = sum z, partial[p][1][c] = sum z^2 over the rows of slice p; rows per slice fixed, lanes walk channels (coalesced), the row
// lanes of a workgroup are summed in a fixed order: bit-reproducible
__global__ __launch_bounds__(256) void nhwc_bn_stats_kernel(const void* __restrict__ z, int dtype, long M, int C, long rows_per, float* __restrict__ partial) {
    // (round 5: 8-channel chunks per thread — 16-byte / 2 x 16-byte loads — instead of one channel with scalar loads; same slices, the row
    //  lanes of a chunk summed through LDS in lane order: bit-reproducible)
    __shared__ float red[256][17];
    const int C8 = C / 8;
    const int RL = C8 >= 256 ? 1 : 256 / C8;
    const int rl = threadIdx.x / (C8 >= 256 ? 256 : C8), c80 = threadIdx.x % (C8 >= 256 ? 256 : C8);
    const long r0 = (long)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int c8 = c80; c8 < C8; c8 += 256) {
        float s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
        auto term = [&](long row) {
            Chunk8 v;
            chunk_load_contig(v, z, dtype, row * C + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) { s[e] += v.v[e]; q[e] += v.v[e] * v.v[e]; }
        };
        long r = r0 + rl;
        for (; r + RL < r1; r += 2 * RL) { term(r); term(r + RL); }
        if (r < r1) term(r);
        if (RL > 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][8 + e] = q[e]; }
            __syncthreads();
            if (rl == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
                for (int k = 0; k < RL; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[e] += red[k * C8 + c80][e]; q[e] += red[k * C8 + c80][8 + e]; }
            }
            __syncthreads();
        }
        if (rl == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                partial[((long)blockIdx.x * 2) * C + c8 * 8 + e] = s[e];
                partial[((long)blockIdx.x * 2 + 1) * C + c8 * 8 + e] = q[e];
            }
        }
    }
}

// per channel: the P partials summed in order (double), scale = gamma * rsqrt(var_biased + eps), shift = beta - mean * scale; running statistics
// updated like nn.BatchNorm2d in training mode (momentum, unbiased variance); ss[c] = scale, ss[C + c] = shift
__global__ __launch_bounds__(256) void nhwc_bn_finalize_kernel(const float* __restrict__ partial, int P, long M, int C, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum, float* __restrict__ run_mean,
                                                               float* __restrict__ run_var, float* __restrict__ ss, float* __restrict__ saved) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int p = 0; p < P; ++p) { s += (double)partial[((long)p * 2) * C + c]; q += (double)partial[((long)p * 2 + 1) * C + c]; }
    const double mean = s / (double)M;
    double var = q / (double)M - mean * mean;
    if (var < 0.0) var = 0.0;
    const float scale = gamma[c] * (float)(1.0 / sqrt(var + (double)eps));
    ss[c] = scale; ss[C + c] = beta[c] - (float)mean * scale;
    if (saved) { saved[c] = (float)mean; saved[C + c] = (float)(1.0 / sqrt(var + (double)eps)); }      // what the backward needs: batch mean, rstd
    if (run_mean) run_mean[c] = (1.f - momentum) * run_mean[c] + momentum * (float)mean;
    if (run_var) run_var[c] = (1.f - momentum) * run_var[c] + momentum * (float)(M > 1 ? var * (double)M / (double)(M - 1) : var);
}

// y = relu?(z * scale[c] + shift[c] + add): 8 channels per thread
__global__ __launch_bounds__(256) void nhwc_bn_apply_kernel(const void* __restrict__ z, int z_dtype, long total8, int C8, const float* __restrict__ ss,
                                                            const void* __restrict__ add, int add_dtype, int relu, void* __restrict__ y, int y_dtype) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int c = (int)(i % C8) * 8, C = C8 * 8;
    Chunk8 v, a;
    chunk_load_contig(v, z, z_dtype, i * 8);
    if (add) chunk_load_contig(a, add, add_dtype, i * 8);
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        o[e] = v.v[e] * ss[c + e] + ss[C + c + e] + (add ? a.v[e] : 0.f);
        if (relu) o[e] = fmaxf(o[e], 0.f);
    }
    if (y_dtype == HULC_BF16) {
        *(uint4*)((uint16_t*)y + i * 8) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
    } else {
        float4* q = (float4*)((float*)y + i * 8);
        q[0] = make_float4(o[0], o[1], o[2], o[3]); q[1] = make_float4(o[4], o[5], o[6], o[7]);
    }
}

// ---- backward of the same BatchNorm (+ ReLU) in training mode (round 5: the affordance trunk's trainable stem needs the data gradient through
// every frozen layer, r3m_rn18.py:34-38).  g = dy * (y > 0) (relu) or dy;  s1[c] = sum g, s2[c] = sum g * xhat, xhat = (z - mean) * rstd;
// dz = gamma * rstd * (g - s1 / M - xhat * s2 / M);  dgamma = s2, dbeta = s1.  Same slicing and summation order as the forward statistics.
__global__ __launch_bounds__(256) void nhwc_bn_bwd_stats_kernel(const void* __restrict__ dy, int dy_dtype, const void* __restrict__ y, int y_dtype,
                                                                const float* __restrict__ z, long M, int C, long rows_per,
                                                                const float* __restrict__ saved, float* __restrict__ partial) {
    // a thread owns an 8-channel chunk (16-byte loads of the bf16 maps, two of the fp32 z) and every RL-th row of the slice, two rows in flight;
    // the row lanes of a chunk are summed through LDS in lane order (fixed: bit-reproducible).  (The first version walked one channel per
    // thread with scalar loads: 250 us per layer at 32 images — 15 x the time its bytes take.)
    __shared__ float red[256][17];
    const int C8 = C / 8;
    const int RL = C8 >= 256 ? 1 : 256 / C8;               // row lanes per chunk column
    const int rl = threadIdx.x / (C8 >= 256 ? 256 : C8), c80 = threadIdx.x % (C8 >= 256 ? 256 : C8);
    const long r0 = (long)blockIdx.x * rows_per, r1 = r0 + rows_per < M ? r0 + rows_per : M;
    for (int c8 = c80; c8 < C8; c8 += 256) {
        float mean[8], rstd[8], s[8], q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mean[e] = saved[c8 * 8 + e]; rstd[e] = saved[C + c8 * 8 + e]; s[e] = 0.f; q[e] = 0.f; }
        auto term = [&](long row) {
            const long off = row * C + c8 * 8;
            Chunk8 g, yy;
            chunk_load_contig(g, dy, dy_dtype, off);
            if (y) chunk_load_contig(yy, y, y_dtype, off);
            const float4 za = *(const float4*)(z + off), zb = *(const float4*)(z + off + 4);
            const float zv[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float gg = (y && !(yy.v[e] > 0.f)) ? 0.f : g.v[e];
                s[e] += gg; q[e] += gg * ((zv[e] - mean[e]) * rstd[e]);
            }
        };
        long r = r0 + rl;
        for (; r + RL < r1; r += 2 * RL) { term(r); term(r + RL); }
        if (r < r1) term(r);
        if (RL > 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { red[threadIdx.x][e] = s[e]; red[threadIdx.x][8 + e] = q[e]; }
            __syncthreads();
            if (rl == 0) {
#pragma unroll
                for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; }
                for (int k = 0; k < RL; ++k)
#pragma unroll
                    for (int e = 0; e < 8; ++e) { s[e] += red[k * C8 + c80][e]; q[e] += red[k * C8 + c80][8 + e]; }
            }
            __syncthreads();
        }
        if (rl == 0) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                partial[((long)blockIdx.x * 2) * C + c8 * 8 + e] = s[e];
                partial[((long)blockIdx.x * 2 + 1) * C + c8 * 8 + e] = q[e];
            }
        }
    }
}

__global__ __launch_bounds__(256) void nhwc_bn_bwd_finalize_kernel(const float* __restrict__ partial, int P, long M, int C, float* __restrict__ sums,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int accumulate) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int p = 0; p < P; ++p) { s += (double)partial[((long)p * 2) * C + c]; q += (double)partial[((long)p * 2 + 1) * C + c]; }
    sums[c] = (float)(s / (double)M); sums[C + c] = (float)(q / (double)M);
    if (dbeta) dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
    if (dgamma) dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)q;
}

// dz = gamma * rstd * (g - mean(g) - xhat * mean(g xhat)), and (optionally) g itself — the gradient the block's shortcut branch receives
__global__ __launch_bounds__(256) void nhwc_bn_bwd_apply_kernel(const void* __restrict__ dy, int dy_dtype, const void* __restrict__ y, int y_dtype,
                                                                const float* __restrict__ z, long total8, int C8, const float* __restrict__ gamma,
                                                                const float* __restrict__ saved, const float* __restrict__ sums,
                                                                void* __restrict__ dz, void* __restrict__ gout, int o_dtype) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total8) return;
    const int c = (int)(i % C8) * 8, C = C8 * 8;
    Chunk8 g, yy;
    chunk_load_contig(g, dy, dy_dtype, i * 8);
    if (y) chunk_load_contig(yy, y, y_dtype, i * 8);
    const float4 za = *(const float4*)(z + i * 8), zb = *(const float4*)(z + i * 8 + 4);
    const float zv[8] = {za.x, za.y, za.z, za.w, zb.x, zb.y, zb.z, zb.w};
    float o[8], gg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        gg[e] = (y && !(yy.v[e] > 0.f)) ? 0.f : g.v[e];
        const float rstd = saved[C + c + e], xhat = (zv[e] - saved[c + e]) * rstd;
        o[e] = gamma[c + e] * rstd * (gg[e] - sums[c + e] - xhat * sums[C + c + e]);
    }
    if (o_dtype == HULC_BF16) {
        *(uint4*)((uint16_t*)dz + i * 8) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]), pack_bf16x2(o[4], o[5]), pack_bf16x2(o[6], o[7]));
        if (gout) *(uint4*)((uint16_t*)gout + i * 8) = make_uint4(pack_bf16x2(gg[0], gg[1]), pack_bf16x2(gg[2], gg[3]), pack_bf16x2(gg[4], gg[5]), pack_bf16x2(gg[6], gg[7]));
    } else {
        float4* q = (float4*)((float*)dz + i * 8);
        q[0] = make_float4(o[0], o[1], o[2], o[3]); q[1] = make_float4(o[4], o[5], o[6], o[7]);
        if (gout) { float4* h = (float4*)((float*)gout + i * 8); h[0] = make_float4(gg[0], gg[1], gg[2], gg[3]); h[1] = make_float4(gg[4], gg[5], gg[6], gg[7]); }
    }
}

// max pool backward, NHWC, 8 channels per thread: dx[pixel] = sum of dy over the windows whose FIRST maximum (scan order kh, kw; strict >, as
// nn.MaxPool2d records its indices) is this pixel.  A gather over the <= ceil(k/s)^2 windows that contain the pixel: no atomics, deterministic.
__global__ __launch_bounds__(256) void maxpool_nhwc_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy, int dtype, int H, int W, int C8,
                                                               int OH, int OW, int k, int s, int pad, long total, void* __restrict__ dx) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over N*H*W*C8
    if (i >= total) return;
    const int c8 = (int)(i % C8); long r = i / C8;
    const int ix = (int)(r % W); r /= W;
    const int iy = (int)(r % H); const long n = r / H;
    Chunk8 me;
    chunk_load_contig(me, x, dtype, i * 8);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // windows (oy, ox) with oy*s - pad <= iy < oy*s - pad + k
    int oy_lo = iy + pad - k + 1; oy_lo = oy_lo <= 0 ? 0 : (oy_lo + s - 1) / s;
    int oy_hi = (iy + pad) / s; if (oy_hi > OH - 1) oy_hi = OH - 1;
    int ox_lo = ix + pad - k + 1; ox_lo = ox_lo <= 0 ? 0 : (ox_lo + s - 1) / s;
    int ox_hi = (ix + pad) / s; if (ox_hi > OW - 1) ox_hi = OW - 1;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
        for (int ox = ox_lo; ox <= ox_hi; ++ox) {
            // is (iy, ix) the FIRST maximum of window (oy, ox)?  per channel: every element before it in scan order is strictly smaller (an equal
            // earlier one would have kept the index), every element after it is not larger
            bool win[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) win[e] = true;
            for (int ky = 0; ky < k; ++ky) {
                const int yy = oy * s - pad + ky;
                if (yy < 0 || yy >= H) continue;
                for (int kx = 0; kx < k; ++kx) {
                    const int xx = ox * s - pad + kx;
                    if (xx < 0 || xx >= W || (yy == iy && xx == ix)) continue;
                    Chunk8 o;
                    chunk_load_contig(o, x, dtype, ((n * H + yy) * W + xx) * (long)C8 * 8 + c8 * 8);
                    const bool before = yy < iy || (yy == iy && xx < ix);
#pragma unroll
                    for (int e = 0; e < 8; ++e) win[e] = win[e] && (before ? o.v[e] < me.v[e] : o.v[e] <= me.v[e]);
                }
            }
            Chunk8 d;
            chunk_load_contig(d, dy, dtype, ((n * OH + oy) * OW + ox) * (long)C8 * 8 + c8 * 8);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += win[e] ? d.v[e] : 0.f;
        }
    if (dtype == HULC_BF16) {
        *(uint4*)((uint16_t*)dx + i * 8) = make_uint4(pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3]), pack_bf16x2(acc[4], acc[5]), pack_bf16x2(acc[6], acc[7]));
    } else {
        float4* q = (float4*)((float*)dx + i * 8);
        q[0] = make_float4(acc[0], acc[1], acc[2], acc[3]); q[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
    }
}

// y [N][Hy][Wy][C] = x [N][H][W][C] placed at (oy + a * step, ox + a * step), zero elsewhere: step 2 / offset 0 = the zero-inserted gradient of a
// stride-2 convolution (its data gradient is then a stride-1 convolution with the flipped taps); step 1 / offset p = zero padding
__global__ __launch_bounds__(256) void nhwc_scatter_kernel(const void* __restrict__ x, int dtype, int H, int W, int C8, int Hy, int Wy, int step, int off,
                                                           long total, void* __restrict__ y) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;      // over N*Hy*Wy*C8
    if (i >= total) return;
    const int c8 = (int)(i % C8); long r = i / C8;
    const int xo = (int)(r % Wy); r /= Wy;
    const int yo = (int)(r % Hy); const long n = r / Hy;
    const int ys = yo - off, xs = xo - off;
    const bool live = ys >= 0 && xs >= 0 && ys % step == 0 && xs % step == 0 && ys / step < H && xs / step < W;
    if (dtype == HULC_BF16) {
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (live) v = *(const uint4*)((const uint16_t*)x + (((n * H + ys / step) * W + xs / step) * (long)C8 + c8) * 8);
        *(uint4*)((uint16_t*)y + i * 8) = v;
    } else {
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
        if (live) { const float4* q = (const float4*)((const float*)x + (((n * H + ys / step) * W + xs / step) * (long)C8 + c8) * 8); a = q[0]; b = q[1]; }
        float4* o = (float4*)((float*)y + i * 8); o[0] = a; o[1] = b;
    }
}

}  // namespace

// see include/hulc2_amd.h
extern "C" long hulc_nhwc_bn_train_workspace(long M, int C) {
    long P = (M + 255) / 256; if (P > 1024) P = 1024; if (P < 1) P = 1;
    return (P * 2 * C + 2 * C) * (long)sizeof(float);
}

extern "C" int hulc_nhwc_bn_train_fwd(const void* z, int z_dtype, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                                      float* run_mean, float* run_var, const void* add, int add_dtype, int relu, void* y, int y_dtype, void* ws,
                                      void* stream) {
    return hulc_nhwc_bn_train_fwd_saved(z, z_dtype, M, C, gamma, beta, eps, momentum, run_mean, run_var, add, add_dtype, relu, y, y_dtype, nullptr, ws, stream);
}

extern "C" int hulc_nhwc_bn_train_fwd_saved(const void* z, int z_dtype, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                                            float* run_mean, float* run_var, const void* add, int add_dtype, int relu, void* y, int y_dtype,
                                            float* saved, void* ws, void* stream) {
    if (!z || !gamma || !beta || !y || !ws) return hulc_fail(-1, "hulc_nhwc_bn_train_fwd: null pointer");
    if (M < 1 || C < 8 || C % 8 || C > 2048 || (C < 256 && 256 % C)) return hulc_fail(-2, "hulc_nhwc_bn_train_fwd: C must be a multiple of 8 (a divisor of 256 below 256)");
    if (((uintptr_t)z | (uintptr_t)y | (uintptr_t)add) % 16) return hulc_fail(-4, "hulc_nhwc_bn_train_fwd: 16-byte aligned rows");
    long P = (M + 255) / 256; if (P > 1024) P = 1024; if (P < 1) P = 1;
    const long rows_per = (M + P - 1) / P;
    P = (M + rows_per - 1) / rows_per;
    float* partial = (float*)ws;
    float* ss = partial + P * 2 * C;
    hipStream_t s = (hipStream_t)stream;
    nhwc_bn_stats_kernel<<<(unsigned)P, 256, 0, s>>>(z, z_dtype, M, C, rows_per, partial);
    nhwc_bn_finalize_kernel<<<(unsigned)((C + 255) / 256), 256, 0, s>>>(partial, (int)P, M, C, gamma, beta, eps, momentum, run_mean, run_var, ss, saved);
    const long total8 = M * (C / 8);
    nhwc_bn_apply_kernel<<<(unsigned)((total8 + 255) / 256), 256, 0, s>>>(z, z_dtype, total8, C / 8, ss, add, add_dtype, relu, y, y_dtype);
    return hulc_check_launch("hulc_nhwc_bn_train_fwd");
}

extern "C" int hulc_r3m_normalize(const float* x, int N, int H, int W, const float* mean3, const float* std3, void* y, int y_dtype, void* stream) {
    if (!x || !mean3 || !std3 || !y) return hulc_fail(-1, "hulc_r3m_normalize: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return hulc_fail(-2, "hulc_r3m_normalize: bad shape");
    const long HW = (long)H * W, total = HW * N;
    r3m_normalize_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, HW, total, mean3[0], mean3[1], mean3[2], 1.f / std3[0],
                                                                                          1.f / std3[1], 1.f / std3[2], y, y_dtype);
    return hulc_check_launch("hulc_r3m_normalize");
}

extern "C" int hulc_r3m_normalize_packed(const float* x, int N, int H, int W, const float* mean3, const float* std3, void* xp, void* stream) {
    if (!x || !mean3 || !std3 || !xp) return hulc_fail(-1, "hulc_r3m_normalize_packed: null pointer");
    if (N <= 0 || H <= 0 || W <= 0) return hulc_fail(-2, "hulc_r3m_normalize_packed: bad shape");
    const int Wp = hulc_r3m_packed_width(W);
    const long total = (long)N * (H + 6) * Wp;
    r3m_normalize_packed_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, H, W, Wp, total, mean3[0], mean3[1], mean3[2],
                                                                                                  1.f / std3[0], 1.f / std3[1], 1.f / std3[2], (uint2*)xp);
    return hulc_check_launch("hulc_r3m_normalize_packed");
}

extern "C" int hulc_maxpool_nhwc(const void* x, int dtype, int N, int H, int W, int C, int k, int stride, int pad, void* y, void* stream) {
    if (!x || !y) return hulc_fail(-1, "hulc_maxpool_nhwc: null pointer");
    if (N <= 0 || C <= 0 || C % 8 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k || H + 2 * pad < k || W + 2 * pad < k)
        return hulc_fail(-2, "hulc_maxpool_nhwc: bad geometry (C must be a multiple of 8, 2 * pad <= k)");
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    const long total = (long)N * OH * OW * (C / 8);
    maxpool_nhwc_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, dtype, H, W, C / 8, OH, OW, k, stride, pad, total, y);
    return hulc_check_launch("hulc_maxpool_nhwc");
}

extern "C" int hulc_nhwc_bn_train_bwd(const void* dy, int dy_dtype, const void* y, int y_dtype, const float* z, long M, int C, const float* gamma,
                                      const float* saved, void* dz, void* g_out, int o_dtype, float* dgamma, float* dbeta, int accumulate_params,
                                      void* ws, void* stream) {
    if (!dy || !z || !gamma || !saved || !dz || !ws) return hulc_fail(-1, "hulc_nhwc_bn_train_bwd: null pointer");
    if (M < 1 || C < 8 || C % 8 || C > 2048 || (C < 256 && 256 % C)) return hulc_fail(-2, "hulc_nhwc_bn_train_bwd: C must be a multiple of 8 (a divisor of 256 below 256)");
    if (((uintptr_t)dy | (uintptr_t)y | (uintptr_t)z | (uintptr_t)dz | (uintptr_t)g_out) % 16) return hulc_fail(-4, "hulc_nhwc_bn_train_bwd: 16-byte aligned rows");
    long P = (M + 255) / 256; if (P > 1024) P = 1024; if (P < 1) P = 1;
    const long rows_per = (M + P - 1) / P;
    P = (M + rows_per - 1) / rows_per;
    float* partial = (float*)ws;
    float* sums = partial + P * 2 * C;
    hipStream_t s = (hipStream_t)stream;
    nhwc_bn_bwd_stats_kernel<<<(unsigned)P, 256, 0, s>>>(dy, dy_dtype, y, y_dtype, z, M, C, rows_per, saved, partial);
    nhwc_bn_bwd_finalize_kernel<<<(unsigned)((C + 255) / 256), 256, 0, s>>>(partial, (int)P, M, C, sums, dgamma, dbeta, accumulate_params);
    const long total8 = M * (C / 8);
    nhwc_bn_bwd_apply_kernel<<<(unsigned)((total8 + 255) / 256), 256, 0, s>>>(dy, dy_dtype, y, y_dtype, z, total8, C / 8, gamma, saved, sums, dz, g_out, o_dtype);
    return hulc_check_launch("hulc_nhwc_bn_train_bwd");
}

extern "C" int hulc_maxpool_nhwc_bwd(const void* x, const void* dy, int dtype, int N, int H, int W, int C, int k, int stride, int pad, void* dx, void* stream) {
    if (!x || !dy || !dx) return hulc_fail(-1, "hulc_maxpool_nhwc_bwd: null pointer");
    if (N <= 0 || C <= 0 || C % 8 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k || H + 2 * pad < k || W + 2 * pad < k)
        return hulc_fail(-2, "hulc_maxpool_nhwc_bwd: bad geometry (C must be a multiple of 8, 2 * pad <= k)");
    const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
    const long total = (long)N * H * W * (C / 8);
    maxpool_nhwc_bwd_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, dy, dtype, H, W, C / 8, OH, OW, k, stride, pad, total, dx);
    return hulc_check_launch("hulc_maxpool_nhwc_bwd");
}

extern "C" int hulc_nhwc_scatter(const void* x, int dtype, int N, int H, int W, int C, int Hy, int Wy, int step, int off, void* y, void* stream) {
    if (!x || !y) return hulc_fail(-1, "hulc_nhwc_scatter: null pointer");
    if (N <= 0 || C <= 0 || C % 8 || step < 1 || off < 0 || (long)(H - 1) * step + off >= Hy || (long)(W - 1) * step + off >= Wy)
        return hulc_fail(-2, "hulc_nhwc_scatter: the placed map must fit the destination (C a multiple of 8)");
    const long total = (long)N * Hy * Wy * (C / 8);
    nhwc_scatter_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, dtype, H, W, C / 8, Hy, Wy, step, off, total, y);
    return hulc_check_launch("hulc_nhwc_scatter");
}
