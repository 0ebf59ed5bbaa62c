// optim.hip — optimizer step and dtype shadows over the flat parameter arena.
//
//   Adam: torch.optim.Adam(lr=2e-4, betas=(0.9,0.999), eps=1e-8, weight_decay=0) as configured by
//   conf/model/optimizer/adam.yaml + hulc2/models/hulc2.py:185-198, applied to one contiguous fp32
//   arena (params / grads / exp_avg / exp_avg_sq share offsets).  HBM-bound: 4 reads + 3 writes of
//   4 bytes (+2 for the bf16 weight shadow the MFMA kernels consume) per parameter, float4-vectorised.
#include <cmath>
#include <cstdlib>
#include "hulc_common.h"
#include "hulc_abi_internal.h"

namespace {

// lo ranges (round 4): the rounding remainders w - bf16(w) of the weights a split-operand forward reads are written by THIS pass (the new
// value is in registers) into a second shadow arena at the same offsets — they were a separate residual launch behind Adam (66-73 us per
// step: a read of p and of the shadow for 6 M elements through 2-byte stores).  Up to 8 element ranges, multiples of 4, kernel arguments.
struct LoRanges { int n; long b[8], e[8]; };

typedef float v4f_t __attribute__((ext_vector_type(4)));
typedef unsigned v2u_t __attribute__((ext_vector_type(2)));
template <bool NT> HULC_DEVICE float4 ld4(const float* a) {
    if (NT) { const v4f_t t = __builtin_nontemporal_load((const v4f_t*)a); return make_float4(t.x, t.y, t.z, t.w); }
    return *(const float4*)a;
}
template <bool NT> HULC_DEVICE void st4(float* a, float x, float y, float z, float w) {
    if (NT) { v4f_t t = {x, y, z, w}; __builtin_nontemporal_store(t, (v4f_t*)a); }
    else *(float4*)a = make_float4(x, y, z, w);
}

template <int NT>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, uint16_t* __restrict__ shadow, long n, float lr, float b1,
                                                   float b2, float eps, float wd, float bc1, float bc2_sqrt, float gscale,
                                                   const unsigned long long* __restrict__ step_state, const int* __restrict__ skip_flag,
                                                   uint16_t* __restrict__ lo, LoRanges lr_, float omb1, float omb2, double b1d, double b2d,
                                                   const float* __restrict__ loss_scale, const float* __restrict__ found_inf) {
    // omb1 / omb2 = 1 - beta as torch forms them: in DOUBLE from the decimal the caller meant (0.999), then rounded to fp32 — 1.f - 0.999f is
    // 4.7e-5 (relative) away from that, and exp_avg_sq with it; the bias corrections likewise come from double powers (torch: Python floats)
    if (skip_flag && *skip_flag) return;    // an upstream kernel reported a fault (barrier timeout): keep the weights, the host raises
    // torch.amp.GradScaler's device scalars (ABI 5, hulc_adam_step_amp): a step whose gradients hold an inf / NaN is skipped (found_inf != 0),
    // the gradients are still multiplied by the loss scale S: unscaled here by 1 / S formed as torch's unscale_ forms it (double reciprocal)
    if (found_inf && *found_inf != 0.f) return;
    if (loss_scale) gscale *= (float)(1.0 / (double)*loss_scale);
    if (step_state) {                       // bias corrections from the device-resident step count (graph replay)
        const double t = (double)step_state[1];
        bc1 = (float)(1.0 - pow(b1d, t));
        bc2_sqrt = (float)sqrt(1.0 - pow(b2d, t));
    }
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 pv = ld4<(NT & 2) != 0>(p + i), gv = ld4<(NT & 1) != 0>(g + i), mv = ld4<(NT & 2) != 0>(m + i), vv = ld4<(NT & 2) != 0>(v + i);
            float pa[4] = {pv.x, pv.y, pv.z, pv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
            float ma[4] = {mv.x, mv.y, mv.z, mv.w}, va[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float gg = ga[k] * gscale + wd * pa[k];
                ma[k] = b1 * ma[k] + omb1 * gg;
                va[k] = b2 * va[k] + omb2 * gg * gg;
                const float denom = sqrtf(va[k]) / bc2_sqrt + eps;
                pa[k] -= (lr / bc1) * (ma[k] / denom);
            }
            st4<(NT & 4) != 0>(p + i, pa[0], pa[1], pa[2], pa[3]);
            st4<(NT & 4) != 0>(m + i, ma[0], ma[1], ma[2], ma[3]);
            st4<(NT & 4) != 0>(v + i, va[0], va[1], va[2], va[3]);
            if (shadow) {
                uint2 s; s.x = pack_bf16x2(pa[0], pa[1]); s.y = pack_bf16x2(pa[2], pa[3]);
                if (NT & 8) { v2u_t t = {s.x, s.y}; __builtin_nontemporal_store(t, (v2u_t*)(shadow + i)); }
                else *(uint2*)(shadow + i) = s;
                if (lo) {
                    bool in = false;
#pragma unroll
                    for (int q = 0; q < 8; ++q) in = in || (q < lr_.n && i >= lr_.b[q] && i < lr_.e[q]);
                    if (in) {
                        uint2 l;
                        l.x = pack_bf16x2(pa[0] - __uint_as_float(s.x << 16), pa[1] - __uint_as_float(s.x & 0xFFFF0000u));
                        l.y = pack_bf16x2(pa[2] - __uint_as_float(s.y << 16), pa[3] - __uint_as_float(s.y & 0xFFFF0000u));
                        *(uint2*)(lo + i) = l;
                    }
                }
            }
        } else {
            for (long k = i; k < n; ++k) {
                float gg = g[k] * gscale + wd * p[k];
                float mm = b1 * m[k] + omb1 * gg, vv = b2 * v[k] + omb2 * gg * gg;
                m[k] = mm; v[k] = vv;
                float pn = p[k] - (lr / bc1) * (mm / (sqrtf(vv) / bc2_sqrt + eps));
                p[k] = pn;
                if (shadow) {
                    const uint16_t hb = f32_to_bf16_bits(pn);
                    shadow[k] = hb;
                    if (lo) {
                        bool in = false;
                        for (int q = 0; q < lr_.n; ++q) in = in || (k >= lr_.b[q] && k < lr_.e[q]);
                        if (in) lo[k] = f32_to_bf16_bits(pn - bf16_bits_to_f32(hb));
                    }
                }
            }
        }
    }
}

__global__ void step_count_advance_if_kernel(unsigned long long* state, const float* found_inf) {
    if (!found_inf || *found_inf == 0.f) state[1] += 1ull;      // (torch's fused Adam takes a skipped step's increment back the same way)
}

__global__ void step_state_advance_kernel(unsigned long long* state, int rng, int step) {
    if (rng) state[0] = state[0] * 6364136223846793005ull + 1442695040888963407ull;   // 64-bit LCG walk of the RNG word
    if (step) state[1] += 1ull;                                                         // optimizer step count
}

// bf16 -> fp32 (gradient payloads come back from a bf16 all-reduce), 8 elements per lane and trip
__global__ __launch_bounds__(256) void cast_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 8;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
        if (i + 7 < n) {
            const uint4 a = *(const uint4*)(src + i);
            *(float4*)(dst + i) = make_float4(__uint_as_float(a.x << 16), __uint_as_float(a.x & 0xFFFF0000u),
                                              __uint_as_float(a.y << 16), __uint_as_float(a.y & 0xFFFF0000u));
            *(float4*)(dst + i + 4) = make_float4(__uint_as_float(a.z << 16), __uint_as_float(a.z & 0xFFFF0000u),
                                                  __uint_as_float(a.w << 16), __uint_as_float(a.w & 0xFFFF0000u));
        } else for (long k = i; k < n; ++k) dst[k] = __uint_as_float((uint32_t)src[k] << 16);
    }
}

// dst[i] = sum_r src[r * chunk + i] in rank order r = 0..W-1, fp32 accumulation: the local reduction of the direct (all-to-all) gradient
// all-reduce.  Every rank owns one chunk and sums the W contributions in the same order, so replicas stay bit-identical.
template <bool BF16>
__global__ __launch_bounds__(256) void sum_chunks_kernel(const void* __restrict__ src_, int W, long chunk, void* __restrict__ dst_) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < chunk; i += stride) {
        float a[4] = {0.f, 0.f, 0.f, 0.f};
        if (BF16) {
            const uint16_t* src = (const uint16_t*)src_;
            for (int r = 0; r < W; ++r) {
                const uint2 v = *(const uint2*)(src + r * chunk + i);
                a[0] += __uint_as_float(v.x << 16); a[1] += __uint_as_float(v.x & 0xFFFF0000u);
                a[2] += __uint_as_float(v.y << 16); a[3] += __uint_as_float(v.y & 0xFFFF0000u);
            }
            uint2 o; o.x = pack_bf16x2(a[0], a[1]); o.y = pack_bf16x2(a[2], a[3]);
            *(uint2*)((uint16_t*)dst_ + i) = o;
        } else {
            const float* src = (const float*)src_;
            for (int r = 0; r < W; ++r) {
                const float4 v = *(const float4*)(src + r * chunk + i);
                a[0] += v.x; a[1] += v.y; a[2] += v.z; a[3] += v.w;
            }
            *(float4*)((float*)dst_ + i) = make_float4(a[0], a[1], a[2], a[3]);
        }
    }
}

__global__ __launch_bounds__(256) void cast_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, long n) {
    const long stride = (long)gridDim.x * blockDim.x * 4;
    for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
        if (i + 3 < n) {
            float4 a = *(const float4*)(src + i);
            uint2 s; s.x = pack_bf16x2(a.x, a.y); s.y = pack_bf16x2(a.z, a.w);
            *(uint2*)(dst + i) = s;
        } else for (long k = i; k < n; ++k) dst[k] = f32_to_bf16_bits(src[k]);
    }
}

// transposed bf16 copies of 2-D weights, all in one launch: tile q = {src offset, rows, cols, tile row, tile col} (64 x 64 tiles);
// dst holds W^T ([cols][rows]) at the same offset.  Lets the data-gradient GEMMs dX = dY W stream W k-major.
HULC_DEVICE void transpose_tiles_job(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, const long* __restrict__ tiles, int ntiles,
                                     int first, int stride) {
    __shared__ __attribute__((aligned(16))) uint16_t t[64][72];
    for (int tile = first; tile < ntiles; tile += stride) {
    const long* q = tiles + (long)tile * 5;
    const long off = q[0]; const int rows = (int)q[1], cols = (int)q[2], r0 = (int)q[3] * 64, c0 = (int)q[4] * 64;
    __syncthreads();                                                            // the previous tile's reads of t are done
    if (((rows | cols) & 7) == 0 && (off & 7) == 0) {
        // 16-byte accesses both ways: a lane moves 8 consecutive elements, a wave 8 tile rows (1 KB per instruction); the 8 elements of an
        // output piece are 8 consecutive ROWS of one source column, picked out of LDS one by one.  (No faster than the 8-byte version: 42 us
        // for the step's 34 M weight elements either way, 2.9 TB/s — the destination is written as 128-byte segments rows x 2 bytes apart,
        // one DRAM page each; the access width was not the limit)
        const int tx = threadIdx.x & 7, ty = threadIdx.x >> 3;                  // 8 lanes x 8 columns, 32 row slots
#pragma unroll
        for (int i = ty; i < 64; i += 32)
            if (r0 + i < rows && c0 + 8 * tx < cols) *(uint4*)&t[i][8 * tx] = *(const uint4*)(src + off + (long)(r0 + i) * cols + c0 + 8 * tx);
        __syncthreads();
#pragma unroll
        for (int i = ty; i < 64; i += 32)
            if (c0 + i < cols && r0 + 8 * tx < rows) {
                uint32_t w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (uint32_t)t[8 * tx + 2 * e][i] | ((uint32_t)t[8 * tx + 2 * e + 1][i] << 16);
                *(uint4*)(dst + off + (long)(c0 + i) * rows + r0 + 8 * tx) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        continue;
    }
    if (((rows | cols) & 3) == 0 && (off & 3) == 0) {
        // 8-byte accesses: a lane moves 4 consecutive elements, a wave 4 tile rows (512 B per instruction instead of 128 B)
        const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;                 // 16 lanes x 4 columns, 16 row slots
        for (int i = ty; i < 64; i += 16)
            if (r0 + i < rows && c0 + 4 * tx < cols) *(uint2*)&t[i][4 * tx] = *(const uint2*)(src + off + (long)(r0 + i) * cols + c0 + 4 * tx);
        __syncthreads();
        for (int i = ty; i < 64; i += 16)
            if (c0 + i < cols && r0 + 4 * tx < rows) {
                const uint32_t lo = (uint32_t)t[4 * tx][i] | ((uint32_t)t[4 * tx + 1][i] << 16), hi = (uint32_t)t[4 * tx + 2][i] | ((uint32_t)t[4 * tx + 3][i] << 16);
                *(uint2*)(dst + off + (long)(c0 + i) * rows + r0 + 4 * tx) = make_uint2(lo, hi);
            }
        continue;
    }
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (int i = ty; i < 64; i += 4)
        if (r0 + i < rows && c0 + tx < cols) t[i][tx] = src[off + (long)(r0 + i) * cols + c0 + tx];
    __syncthreads();
    for (int i = ty; i < 64; i += 4)
        if (c0 + i < cols && r0 + tx < rows) dst[off + (long)(c0 + i) * rows + r0 + tx] = t[tx][i];
    }
}

__global__ __launch_bounds__(256) void transpose_tiles_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst,
                                                              const long* __restrict__ tiles, int ntiles) {
    transpose_tiles_job(src, dst, tiles, ntiles, blockIdx.x, gridDim.x);
}

// conv weight repacks (fp32 OIHW parameter -> bf16 kernel layouts), one workgroup column per table entry
HULC_DEVICE void repack_conv_job(const float* __restrict__ src, uint16_t* __restrict__ dst, const long* __restrict__ table, int entry, int sub, int nsub) {
    const long* q = table + (long)entry * 7;
    const long so = q[0], d0 = q[1];
    const int Cout = (int)q[2], Cin = (int)q[3], KH = (int)q[4], KW = (int)q[5], mode = (int)q[6] & 7;
    const bool lo = ((int)q[6] & 8) != 0;                  // (round 5) + 8: the ROUNDING REMAINDER bf16(w - float(bf16(w))) in the same layout
    const int n = Cout * Cin * KH * KW, taps = KH * KW;
    for (int i = sub * blockDim.x + threadIdx.x; i < n; i += nsub * blockDim.x) {
        int o, c, t;                                      // destination index i -> (o, c, tap)
        if (mode == 0) { o = i / (Cin * taps); c = (i / taps) % Cin; t = i % taps; }
        else if (mode == 1) { o = i / (taps * Cin); t = (i / Cin) % taps; c = i % Cin; }
        else if (mode == 2) { c = i / (taps * Cout); t = (i / Cout) % taps; o = i % Cout; }
        else { t = i / (Cin * Cout); c = (i / Cout) % Cin; o = i % Cout; }     // mode 3: [tap][c][o], the transposed flatten-linear operand
        const float w = src[so + ((long)o * Cin + c) * taps + t];
        const uint16_t hi = f32_to_bf16_bits(w);
        dst[d0 + i] = lo ? f32_to_bf16_bits(w - bf16_bits_to_f32(hi)) : hi;
    }
}

__global__ __launch_bounds__(256) void repack_conv_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst, const long* __restrict__ table) {
    repack_conv_job(src, dst, table, blockIdx.y, blockIdx.x, gridDim.x);
}

// (round 4) the transposed tiles and the conv repacks of a step as ONE launch: independent jobs on disjoint block ranges — the first nbt
// workgroups walk the tiles, the rest are 128 workgroups per conv-table entry.  They were two launches (38 + 12 us) behind Adam.
__global__ __launch_bounds__(256) void derive_copies_kernel(const uint16_t* __restrict__ bf16, uint16_t* __restrict__ bf16_t, const long* __restrict__ tiles,
                                                            int ntiles, int nbt, const float* __restrict__ p32, uint16_t* __restrict__ conv_dst,
                                                            const long* __restrict__ conv_table) {
    const int b = blockIdx.x;
    if (b < nbt) transpose_tiles_job(bf16, bf16_t, tiles, ntiles, b, nbt);
    else repack_conv_job(p32, conv_dst, conv_table, (b - nbt) / 128, (b - nbt) % 128, 128);
}

}  // namespace

extern "C" int hulc_derive_copies(const void* bf16, void* bf16_t, const long* tiles, int ntiles, const float* p32, void* conv_dst,
                                  const long* conv_table, int nconv, void* stream) {
    if (ntiles > 0 && (!bf16 || !bf16_t || !tiles)) return hulc_fail(-1, "hulc_derive_copies: null pointer (tiles)");
    if (nconv > 0 && (!p32 || !conv_dst || !conv_table)) return hulc_fail(-1, "hulc_derive_copies: null pointer (conv table)");
    if (ntiles < 0 || nconv < 0) return hulc_fail(-2, "hulc_derive_copies: negative count");
    const int nbt = ntiles < 4096 ? ntiles : 4096;
    const long grid = (long)nbt + (long)nconv * 128;
    if (grid == 0) return 0;
    derive_copies_kernel<<<(unsigned)grid, 256, 0, (hipStream_t)stream>>>((const uint16_t*)bf16, (uint16_t*)bf16_t, tiles, ntiles, nbt, p32,
                                                                         (uint16_t*)conv_dst, conv_table);
    return hulc_check_launch("hulc_derive_copies");
}

extern "C" int hulc_step_state_advance_words(unsigned long long* state, int rng, int step, void* stream) {
    if (!state) return hulc_fail(-1, "hulc_step_state_advance: null pointer");
    step_state_advance_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state, rng, step);
    return hulc_check_launch("hulc_step_state_advance");
}

extern "C" int hulc_step_state_advance(unsigned long long* state, void* stream) { return hulc_step_state_advance_words(state, 1, 1, stream); }

extern "C" int hulc_cast_bf16_to_f32(const void* src, float* dst, long n, void* stream) {
    if (!src || !dst) return hulc_fail(-1, "hulc_cast_bf16_to_f32: null pointer");
    if (((uintptr_t)src % 16) || ((uintptr_t)dst % 16)) return hulc_fail(-4, "hulc_cast_bf16_to_f32: misaligned");
    long blocks = (n / 8 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    cast_f32_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint16_t*)src, dst, n);
    return hulc_check_launch("hulc_cast_bf16_to_f32");
}

extern "C" int hulc_sum_chunks(const void* src, int dtype, int W, long chunk, void* dst, void* stream) {
    if (!src || !dst) return hulc_fail(-1, "hulc_sum_chunks: null pointer");
    if (W < 1 || chunk < 0 || (chunk & 7)) return hulc_fail(-2, "hulc_sum_chunks: chunk must be a multiple of 8 elements, W >= 1");
    if (((uintptr_t)src | (uintptr_t)dst) % 16) return hulc_fail(-4, "hulc_sum_chunks: misaligned");
    if (chunk == 0) return 0;
    long blocks = (chunk / 4 + 255) / 256; if (blocks > 4096) blocks = 4096;
    if (dtype == HULC_BF16) sum_chunks_kernel<true><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(src, W, chunk, dst);
    else sum_chunks_kernel<false><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(src, W, chunk, dst);
    return hulc_check_launch("hulc_sum_chunks");
}

extern "C" int hulc_adam_step_amp(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                                  void* lo_shadow, const long* lo_ranges, int n_ranges, const float* loss_scale, const float* found_inf, void* stream);
extern "C" int hulc_adam_step_lo(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                                 void* lo_shadow, const long* lo_ranges, int n_ranges, void* stream);

extern "C" int hulc_adam_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                              float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag, void* stream) {
    return hulc_adam_step_lo(p, g, m, v, bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, step, step_state, grad_scale, skip_flag, nullptr, nullptr, 0, stream);
}

extern "C" int hulc_adam_step_lo(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                                 float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                                 void* lo_shadow, const long* lo_ranges, int n_ranges, void* stream) {
    return hulc_adam_step_amp(p, g, m, v, bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, step, step_state, grad_scale, skip_flag, lo_shadow, lo_ranges,
                              n_ranges, nullptr, nullptr, stream);
}

extern "C" int hulc_step_count_advance_if(unsigned long long* state, const float* found_inf, void* stream) {
    if (!state) return hulc_fail(-1, "hulc_step_count_advance_if: null state");
    step_count_advance_if_kernel<<<1, 1, 0, (hipStream_t)stream>>>(state, found_inf);
    return hulc_check_launch("hulc_step_count_advance_if");
}

// see include/hulc2_amd.h
extern "C" int hulc_adam_step_amp(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                                  float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                                  void* lo_shadow, const long* lo_ranges, int n_ranges, const float* loss_scale, const float* found_inf, void* stream) {
    if (!p || !g || !m || !v) return hulc_fail(-1, "hulc_adam_step: null pointer");
    LoRanges lr_;
    lr_.n = 0;
    for (int q = 0; q < 8; ++q) lr_.b[q] = lr_.e[q] = 0;
    if (lo_shadow) {
        if (!bf16_shadow || !lo_ranges || n_ranges < 1 || n_ranges > 8) return hulc_fail(-2, "hulc_adam_step_lo: lo_shadow needs bf16_shadow and 1..8 ranges");
        if ((uintptr_t)lo_shadow % 8) return hulc_fail(-4, "hulc_adam_step_lo: lo_shadow must be 8-byte aligned");
        for (int q = 0; q < n_ranges; ++q) {
            if ((lo_ranges[2 * q] & 3) || lo_ranges[2 * q] < 0 || lo_ranges[2 * q + 1] < lo_ranges[2 * q] || lo_ranges[2 * q + 1] > n)
                return hulc_fail(-2, "hulc_adam_step_lo: ranges must start at multiples of 4 inside the arena");
            lr_.b[q] = lo_ranges[2 * q]; lr_.e[q] = lo_ranges[2 * q + 1];
        }
        lr_.n = n_ranges;
    }
    if (((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) % 16) return hulc_fail(-4, "hulc_adam_step: arenas must be 16-byte aligned");
    if (!step_state && step < 1) return hulc_fail(-2, "hulc_adam_step: step counts from 1");
    if (step < 1) step = 1;
    // the betas arrive as floats (ABI); the decimal the caller wrote is recovered when the float is within rounding of a 7-digit decimal
    auto meant = [](float b) -> double { const double d = (double)b, r = std::round(d * 1e7) / 1e7; return std::fabs(r - d) <= 6e-8 * std::fabs(d) ? r : d; };
    const double b1d = meant(beta1), b2d = meant(beta2);
    const float bc1 = (float)(1.0 - std::pow(b1d, (double)step)), bc2s = (float)std::sqrt(1.0 - std::pow(b2d, (double)step));
    // (round 5, tools/adam_sweep.py on two boxes) every operand of the pass is touched once per step and the arenas (753 MB) are three times the
    // MALL: non-temporal loads of p / g / m / v and stores of p / m / v take the 47 M-element pass from 256-288 us to 241-251 us (5.5 -> 5.8 TB/s of
    // its 30 B per element); the bf16 shadow keeps the default policy (derive_copies reads it next).  8192 workgroups instead of 4096: -3 %.
    // HULC_ADAM_NT (bit 0: g loads, 1: p / m / v loads, 2: p / m / v stores, 3: shadow stores) / HULC_ADAM_BLOCKS: the sweep's knobs.
    static const long cap = getenv("HULC_ADAM_BLOCKS") ? atol(getenv("HULC_ADAM_BLOCKS")) : 8192;
    static const int nt = getenv("HULC_ADAM_NT") ? atoi(getenv("HULC_ADAM_NT")) : 7;
    long blocks = (n / 4 + 255) / 256; if (blocks > cap) blocks = cap; if (blocks < 1) blocks = 1;
#define ADAM_GO(NTV) adam_kernel<NTV><<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(p, g, m, v, (uint16_t*)bf16_shadow, n, lr, beta1, beta2, eps, weight_decay, \
                                                                   bc1, bc2s, grad_scale, step_state, skip_flag, (uint16_t*)lo_shadow, lr_, \
                                                                   (float)(1.0 - b1d), (float)(1.0 - b2d), b1d, b2d, loss_scale, found_inf)
    switch (nt) {
        case 1: ADAM_GO(1); break; case 3: ADAM_GO(3); break; case 4: ADAM_GO(4); break; case 5: ADAM_GO(5); break;
        case 0: ADAM_GO(0); break; case 15: ADAM_GO(15); break; case 12: ADAM_GO(12); break; case 13: ADAM_GO(13); break;
        default: ADAM_GO(7);
    }
#undef ADAM_GO
    return hulc_check_launch("hulc_adam_step");
}

extern "C" int hulc_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream) {
    if (!src || !dst) return hulc_fail(-1, "hulc_cast_f32_to_bf16: null pointer");
    if (((uintptr_t)src % 16) || ((uintptr_t)dst % 8)) return hulc_fail(-4, "hulc_cast_f32_to_bf16: misaligned");
    long blocks = (n / 4 + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    cast_bf16_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(src, (uint16_t*)dst, n);
    return hulc_check_launch("hulc_cast_f32_to_bf16");
}

extern "C" int hulc_transpose_bf16_tiles(const void* src, void* dst, const long* tiles, int ntiles, void* stream) {
    if (!src || !dst || !tiles) return hulc_fail(-1, "hulc_transpose_bf16_tiles: null pointer");
    if (ntiles <= 0) return 0;
    transpose_tiles_kernel<<<(unsigned)(ntiles < 4096 ? ntiles : 4096), 256, 0, (hipStream_t)stream>>>((const uint16_t*)src, (uint16_t*)dst, tiles, ntiles);
    return hulc_check_launch("hulc_transpose_bf16_tiles");
}

// ---- fragment-packed copies of the transformer feed-forward weights (csrc/txl_block.hip) ---------------------------------------------
// element i = ((hb * 8 + f) * 64 + lane) * 8 + j of a packed array is the j-th k-slot of MFMA fragment f that lane `lane` (r = lane & 31,
// hf = lane >> 5) needs for the 32 hidden units 32 hb ..: one 16-byte load per lane, 1 KB contiguous per instruction.
//   layout 0: A of z^T = W1 y^T           source W1  [FF][128]:  row 32 hb + r,  col 16 f + 8 hf + j
//   layout 1: A of f^T += W2 h^T          source W2  [128][FF]:  row 32 (f >> 1) + r,  col 32 hb + 16 (f & 1) + 4 hf + split(j)
//   layout 2: A of dh^T = W2^T df^T       source W2T [FF][128]:  row 32 hb + r,  col 32 (f >> 1) + 16 (f & 1) + 4 hf + split(j)
//   layout 3: A of dy^T += W1^T dh^T      source W1T [128][FF]:  row 32 (f >> 1) + r,  col 32 hb + 16 (f & 1) + 4 hf + split(j)
// split(j) = j for j < 4, j + 4 otherwise: the k-slot order of an MFMA accumulator tile reused as an operand (txl_fused.hip).
extern "C" int hulc_ffn_frag_perm(int layout, int FF, int* out) {
    if (!out || layout < 0 || layout > 3 || FF < 32 || FF % 32) return hulc_fail(-2, "hulc_ffn_frag_perm: layout 0..3, FF a multiple of 32");
    long i = 0;
    for (int hb = 0; hb < FF / 32; ++hb)
        for (int f = 0; f < 8; ++f)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j, ++i) {
                    const int r = lane & 31, hf = lane >> 5, sp = j < 4 ? j : j + 4;
                    long src;
                    if (layout == 0) src = (long)(32 * hb + r) * 128 + 16 * f + 8 * hf + j;
                    else if (layout == 1) src = (long)(32 * (f >> 1) + r) * FF + 32 * hb + 16 * (f & 1) + 4 * hf + sp;
                    else if (layout == 2) src = (long)(32 * hb + r) * 128 + 32 * (f >> 1) + 16 * (f & 1) + 4 * hf + sp;
                    else src = (long)(32 * (f >> 1) + r) * FF + 32 * hb + 16 * (f & 1) + 4 * hf + sp;
                    out[i] = (int)src;
                }
    return 0;
}

namespace {
// dst chunk c (4 bf16 = 8 bytes) = chunk idx[c] & 0x7fffffff of src1 (bit 31 set) or src0
__global__ __launch_bounds__(256) void gather_chunks_kernel(const uint2* __restrict__ src0, const uint2* __restrict__ src1, uint2* __restrict__ dst,
                                                           const unsigned* __restrict__ idx, long n) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n) return;
    const unsigned k = idx[c];
    dst[c] = (k >> 31) ? src1[k & 0x7fffffffu] : src0[k];
}
}  // namespace

namespace {
__global__ __launch_bounds__(256) void gather_chunks2_kernel(const uint2* __restrict__ a0, const uint2* __restrict__ a1, uint2* __restrict__ ad,
                                                            const unsigned* __restrict__ ai, long an, const uint2* __restrict__ b0,
                                                            uint2* __restrict__ bd, const unsigned* __restrict__ bi, long bn) {
    long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long ablocks = (an + 255) / 256 * 256;
    if (c < ablocks) {
        if (c >= an) return;
        const unsigned k = ai[c];
        ad[c] = (k >> 31) ? a1[k & 0x7fffffffu] : a0[k];
    } else {
        c -= ablocks;
        if (c >= bn) return;
        bd[c] = b0[bi[c] & 0x7fffffffu];
    }
}
}  // namespace

// see include/hulc2_amd.h
extern "C" int hulc_gather_chunks2(const void* a0, const void* a1, void* ad, const unsigned* ai, long an, const void* b0, void* bd,
                                   const unsigned* bi, long bn, void* stream) {
    if (an < 0 || bn < 0) return hulc_fail(-2, "hulc_gather_chunks2: negative count");
    if (an > 0 && (!a0 || !ad || !ai)) return hulc_fail(-1, "hulc_gather_chunks2: null pointer (first gather)");
    if (bn > 0 && (!b0 || !bd || !bi)) return hulc_fail(-1, "hulc_gather_chunks2: null pointer (second gather)");
    if (((uintptr_t)a0 | (uintptr_t)a1 | (uintptr_t)ad | (uintptr_t)b0 | (uintptr_t)bd) % 8) return hulc_fail(-4, "hulc_gather_chunks2: arrays must be 8-byte aligned");
    const long blocks = (an + 255) / 256 + (bn + 255) / 256;
    if (blocks == 0) return 0;
    gather_chunks2_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>((const uint2*)a0, (const uint2*)a1, (uint2*)ad, ai, an, (const uint2*)b0,
                                                                            (uint2*)bd, bi, bn);
    return hulc_check_launch("hulc_gather_chunks2");
}

// see include/hulc2_amd.h
extern "C" int hulc_gather_chunks(const void* src0, const void* src1, void* dst, const unsigned* idx, long nchunks, void* stream) {
    if (!src0 || !dst || !idx) return hulc_fail(-1, "hulc_gather_chunks: null pointer");
    if (((uintptr_t)src0 | (uintptr_t)src1 | (uintptr_t)dst) % 8) return hulc_fail(-4, "hulc_gather_chunks: arrays must be 8-byte aligned");
    if (nchunks <= 0) return 0;
    gather_chunks_kernel<<<(unsigned)((nchunks + 255) / 256), 256, 0, (hipStream_t)stream>>>((const uint2*)src0, (const uint2*)src1, (uint2*)dst, idx, nchunks);
    return hulc_check_launch("hulc_gather_chunks");
}

namespace {
// lo[dst + i] = bf16(p[src + i] - float(hi[src + i])) for the segments of the table {src offset, count, dst offset}
__global__ __launch_bounds__(256) void residual_bf16_kernel(const float* __restrict__ p32, const uint16_t* __restrict__ hi, uint16_t* __restrict__ lo,
                                                           const long* __restrict__ seg) {
    const long* q = seg + (long)blockIdx.y * 3;
    const long src = q[0], n = q[1], dst = q[2];
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        lo[dst + i] = f32_to_bf16_bits(p32[src + i] - bf16_bits_to_f32(hi[src + i]));
}
}  // namespace

// see include/hulc2_amd.h
extern "C" int hulc_residual_bf16(const float* p32, const void* hi, void* lo, const long* segments, int nseg, void* stream) {
    if (!p32 || !hi || !lo || !segments) return hulc_fail(-1, "hulc_residual_bf16: null pointer");
    if (nseg <= 0) return 0;
    residual_bf16_kernel<<<dim3(64, (unsigned)nseg), 256, 0, (hipStream_t)stream>>>(p32, (const uint16_t*)hi, (uint16_t*)lo, segments);
    return hulc_check_launch("hulc_residual_bf16");
}

extern "C" int hulc_repack_conv_weights(const float* src, void* dst, const long* table, int n, void* stream) {
    if (!src || !dst || !table) return hulc_fail(-1, "hulc_repack_conv_weights: null pointer");
    if (n <= 0) return 0;
    repack_conv_kernel<<<dim3(128, (unsigned)n), 256, 0, (hipStream_t)stream>>>(src, (uint16_t*)dst, table);
    return hulc_check_launch("hulc_repack_conv_weights");
}
