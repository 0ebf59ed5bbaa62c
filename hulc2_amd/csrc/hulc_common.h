// hulc_common.h — device-side helpers shared by every gfx950 kernel of the HULC++ hot path.
//
// CDNA4 only: 64-lane wavefronts, MFMA 32x32 tiles, LDS-staged operands.  No CUDA/compat paths.
//
// Compute types (CT):
//   bf16  : v_mfma_f32_32x32x16_bf16, fp32 accumulate (throughput mode, the benchmarked one)
//   float : v_mfma_f32_32x32x2_f32, exact fp32 (k-ordered fmaf chain) — parity/debug mode
// Storage types (runtime dtype codes, HULC_F32 / HULC_BF16) are decoupled from the compute type:
// tiles are converted while they are staged into LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define HULC_F32 0
#define HULC_BF16 1
#define HULC_F16 2      // (ABI 7) the finer twin of a bf16 map: conv forward output next to y_bf16, spatial softmax input

typedef __bf16 bf16_t;
typedef bf16_t bf16x8_t __attribute__((ext_vector_type(8)));
typedef float f32x16_t __attribute__((ext_vector_type(16)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));

#define HULC_DEVICE __device__ __forceinline__

// ---------------------------------------------------------------------------------------------
// scalar conversions
// ---------------------------------------------------------------------------------------------
HULC_DEVICE float bf16_bits_to_f32(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

typedef bf16_t bf16x2_t __attribute__((ext_vector_type(2)));

// fp32 -> bf16 goes through the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, the rounding
// torch uses for .to(bfloat16)); one instruction per pair instead of ~6 VALU ops per element in software.
HULC_DEVICE uint32_t pack_bf16x2(float lo, float hi) {
    f32x2_t f = {lo, hi};
    union { bf16x2_t b; uint32_t u; } x;
    x.b = __builtin_convertvector(f, bf16x2_t);
    return x.u;
}

// ---- packed bf16 pairs as 16-bit integers (epilogues of the band kernels: their tile loops are bound by instruction issue, every VALU counts)
typedef short s16x2_t __attribute__((ext_vector_type(2)));
HULC_DEVICE uint32_t max_s16x2(uint32_t w, uint32_t floor2) {      // floor2 = 0: ReLU of two bf16 (their bit patterns order like sign-magnitude
    union { uint32_t u; s16x2_t s; } x, f; x.u = w; f.u = floor2;   //  integers: negative halves, -0 included, become +0); 0x80008000: identity
    x.s = __builtin_elementwise_max(x.s, f.s);
    return x.u;
}
HULC_DEVICE uint32_t nonzero_u16x2(uint32_t w) {                    // 1 per non-zero 16-bit half, at bits 0 and 16 (as min(half, 1); written as an
    uint32_t r;                                                     //  elementwise min the compiler expands it to compare + select + permute per half)
    asm("v_pk_min_u16 %0, %1, %2" : "=v"(r) : "v"(w), "s"(0x00010001u));
    return r;
}
HULC_DEVICE uint32_t keep_u16x2(uint32_t w, uint32_t two) {         // bit 0 / bit 1 of `two`: keep the low / high half of w, else zero it
    const uint32_t t = (two | (two << 15)) & 0x00010001u;
    uint32_t r;
    asm("v_pk_mul_lo_u16 %0, %1, %2" : "=v"(r) : "v"(w), "v"(t));
    return r;
}

HULC_DEVICE uint16_t f32_to_bf16_bits(float f) { return (uint16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }

// generic typed element load/store by runtime dtype code
HULC_DEVICE float load_elem(const void* p, int dtype, long idx) {
    if (dtype == HULC_F32) return ((const float*)p)[idx];
    return bf16_bits_to_f32(((const uint16_t*)p)[idx]);
}
HULC_DEVICE void store_elem(void* p, int dtype, long idx, float v) {
    if (dtype == HULC_F32) ((float*)p)[idx] = v;
    else ((uint16_t*)p)[idx] = f32_to_bf16_bits(v);
}

// ---------------------------------------------------------------------------------------------
// 8-element k-chunk: the unit every MFMA operand loader works in.  Held as 8 floats in registers
// between the global load and the LDS write so one code path serves f32 and bf16 sources.
// ---------------------------------------------------------------------------------------------
struct Chunk8 {
    float v[8];
};

HULC_DEVICE void chunk_zero(Chunk8& c) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c.v[j] = 0.f;
}

// keep the chunk when `keep`, else zeros — as selects, never as a branch around the load that produced it: hipcc
// wraps a conditionally executed load in a branch and waits vmcnt(0) behind it, which serialises every load of a
// tile stage into its own memory round trip (cdna_hip_programming.md §5 "Three .s-level traps", c).  All operand
// loaders therefore load unconditionally from a clamped (always valid) address and zero the result afterwards.
HULC_DEVICE void chunk_keep_if(Chunk8& c, bool keep) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c.v[j] = keep ? c.v[j] : 0.f;
}

// 8 contiguous elements starting at element offset `off` (off % 8 == 0, base 16B aligned)
HULC_DEVICE void chunk_load_contig(Chunk8& c, const void* base, int dtype, long off) {
    if (dtype == HULC_F32) {
        const float4* p = (const float4*)((const float*)base + off);
        float4 a = p[0], b = p[1];
        c.v[0] = a.x; c.v[1] = a.y; c.v[2] = a.z; c.v[3] = a.w;
        c.v[4] = b.x; c.v[5] = b.y; c.v[6] = b.z; c.v[7] = b.w;
    } else if (dtype == HULC_F16) {
        union { uint4 u; _Float16 h[8]; } r; r.u = *(const uint4*)((const uint16_t*)base + off);
#pragma unroll
        for (int j = 0; j < 8; ++j) c.v[j] = (float)r.h[j];
    } else {
        uint4 r = *(const uint4*)((const uint16_t*)base + off);
        c.v[0] = __uint_as_float(r.x << 16); c.v[1] = __uint_as_float(r.x & 0xffff0000u);
        c.v[2] = __uint_as_float(r.y << 16); c.v[3] = __uint_as_float(r.y & 0xffff0000u);
        c.v[4] = __uint_as_float(r.z << 16); c.v[5] = __uint_as_float(r.z & 0xffff0000u);
        c.v[6] = __uint_as_float(r.w << 16); c.v[7] = __uint_as_float(r.w & 0xffff0000u);
    }
}

// 8 elements at stride `ld` (element j at off + j*ld); elements with j >= nvalid are zero.
HULC_DEVICE void chunk_load_strided(Chunk8& c, const void* base, int dtype, long off, long ld, int nvalid) {
#pragma unroll
    for (int j = 0; j < 8; ++j) c.v[j] = (j < nvalid) ? load_elem(base, dtype, off + (long)j * ld) : 0.f;
}

// write a chunk into an LDS tile row as compute-type elements
template <typename CT>
HULC_DEVICE void chunk_store_lds(char* dst, const Chunk8& c);

template <>
HULC_DEVICE void chunk_store_lds<bf16_t>(char* dst, const Chunk8& c) {
    uint4 r;
    r.x = pack_bf16x2(c.v[0], c.v[1]);
    r.y = pack_bf16x2(c.v[2], c.v[3]);
    r.z = pack_bf16x2(c.v[4], c.v[5]);
    r.w = pack_bf16x2(c.v[6], c.v[7]);
    *(uint4*)dst = r;
}
template <>
HULC_DEVICE void chunk_store_lds<float>(char* dst, const Chunk8& c) {
    ((float4*)dst)[0] = make_float4(c.v[0], c.v[1], c.v[2], c.v[3]);
    ((float4*)dst)[1] = make_float4(c.v[4], c.v[5], c.v[6], c.v[7]);
}

// ---------------------------------------------------------------------------------------------
// LDS operand tile geometry.  A tile row holds 64 bytes of k-data (32 bf16 or 16 f32) plus a 16-byte
// pad: row stride 80 B = 20 dwords, so the 16 lanes of every ds_read_b128 lane group land on 16
// distinct 16-byte slots of the 256-byte bank row (20*i mod 64 are distinct multiples of 4 for any
// 16 rows that differ mod 16) — conflict-free fragment reads without an XOR swizzle.
// ---------------------------------------------------------------------------------------------
#define HULC_ROWB 80

template <typename CT> struct MmaTraits;
template <> struct MmaTraits<bf16_t> {
    static constexpr int KT = 32;    // k elements per LDS tile
    static constexpr int NCH = 4;    // 8-element chunks per tile row
    static constexpr int CHB = 16;   // bytes per chunk in LDS
};
template <> struct MmaTraits<float> {
    static constexpr int KT = 16;
    static constexpr int NCH = 2;
    static constexpr int CHB = 32;
};

// One LDS tile (KT k-elements) worth of MFMA work for a wave owning TM x TN 32x32 accumulators.
//   a_rows: LDS address of the wave's first A row (rows = output rows m), b_rows: same for B
//   (rows = output columns n).  MFMA semantics: D[i][j] += sum_k A[i][k] * B[k][j], A-fragment lane
//   l supplies row i = l & 31, B-fragment lane l supplies column j = l & 31; lanes < 32 carry the
//   low half of the instruction's k-range and lanes >= 32 the high half, identically for A and B.
template <typename CT, int TM, int TN>
HULC_DEVICE void mma_tile(const char* a_rows, const char* b_rows, f32x16_t (&acc)[TM][TN], int lane);

template <int TM, int TN>
HULC_DEVICE void mma_tile_bf16(const char* a_rows, const char* b_rows, f32x16_t (&acc)[TM][TN], int lane) {
    const int r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        bf16x8_t a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) a[i] = *(const bf16x8_t*)(a_rows + (i * 32 + r) * HULC_ROWB + (ks * 2 + h) * 16);
#pragma unroll
        for (int j = 0; j < TN; ++j) b[j] = *(const bf16x8_t*)(b_rows + (j * 32 + r) * HULC_ROWB + (ks * 2 + h) * 16);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[j], acc[i][j], 0, 0, 0);
    }
}

template <int TM, int TN>
HULC_DEVICE void mma_tile_f32(const char* a_rows, const char* b_rows, f32x16_t (&acc)[TM][TN], int lane) {
    const int r = lane & 31, h = lane >> 5;
    float a[TM][8], b[TN][8];
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        const float4* p = (const float4*)(a_rows + (i * 32 + r) * HULC_ROWB + h * 32);
        float4 x = p[0], y = p[1];
        a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w;
        a[i][4] = y.x; a[i][5] = y.y; a[i][6] = y.z; a[i][7] = y.w;
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const float4* p = (const float4*)(b_rows + (j * 32 + r) * HULC_ROWB + h * 32);
        float4 x = p[0], y = p[1];
        b[j][0] = x.x; b[j][1] = x.y; b[j][2] = x.z; b[j][3] = x.w;
        b[j][4] = y.x; b[j][5] = y.y; b[j][6] = y.z; b[j][7] = y.w;
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
}

template <typename CT, int TM, int TN> struct MmaTile;
template <int TM, int TN> struct MmaTile<bf16_t, TM, TN> {
    static HULC_DEVICE void run(const char* a, const char* b, f32x16_t (&acc)[TM][TN], int lane) {
        mma_tile_bf16<TM, TN>(a, b, acc, lane);
    }
};
template <int TM, int TN> struct MmaTile<float, TM, TN> {
    static HULC_DEVICE void run(const char* a, const char* b, f32x16_t (&acc)[TM][TN], int lane) {
        mma_tile_f32<TM, TN>(a, b, acc, lane);
    }
};

// accumulator element (reg) of a 32x32 tile -> row inside the tile; the column is lane & 31.
// x / d for 0 <= x < 2^20 with inv = v_rcp_f32(d) (1 ulp): exact — (x + 0.5) / d is at least 0.5 / d away from an integer, the error of the
// product is below x / d x 2^-22, i.e. below that margin for every x < 2^20 — three VALU ops instead of the ~25 of an integer division.  The conv
// band staging plans divide by a band / map width for every 16-byte chunk of every unit (500 VALU instructions per wave and unit before the
// first MFMA of conv3).
HULC_DEVICE int fast_div(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

HULC_DEVICE float fast_rcp(int d) { return __builtin_amdgcn_rcpf((float)d); }

HULC_DEVICE int acc_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

// ---------------------------------------------------------------------------------------------
// wavefront reductions (64 lanes)
// ---------------------------------------------------------------------------------------------
HULC_DEVICE float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
HULC_DEVICE float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// ---------------------------------------------------------------------------------------------
// counter-based RNG for dropout / sampling: one 32-bit draw per (seed, index).  Philox-like mixing
// (two rounds of a 64-bit multiply-xorshift); the same (seed, idx) gives the same bit pattern in the
// forward and the backward kernels, so no mask tensor is stored.
// ---------------------------------------------------------------------------------------------
HULC_DEVICE uint64_t hulc_rand64(uint64_t seed, uint64_t idx) {
    uint64_t z = idx * 0x9E3779B97F4A7C15ull + seed;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
HULC_DEVICE uint32_t hulc_rand32(uint64_t seed, uint64_t idx) { return (uint32_t)(hulc_rand64(seed, idx) >> 32); }
// Dropout keeps element idx when ITS 16 bits of the 64-bit draw of (seed, idx / 4) reach p * 2^16: four consecutive elements share one
// draw (three 64-bit multiplies are ~50 instruction slots on this ALU; a kernel that owns 4 consecutive elements pays them once:
// dropout_scale4), every kernel sees the same mask for the same (seed, idx), and no mask tensor is stored.  The keep probability is
// 1 - floor(p * 65536) / 65536 (p = 0.1: 0.900009), the scale 1 / (1 - p).
HULC_DEVICE float dropout_scale(uint64_t seed, uint64_t idx, float p) {
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    const uint32_t u = (uint32_t)(hulc_rand64(seed, idx >> 2) >> (16 * (uint32_t)(idx & 3))) & 0xffffu;
    return u >= thr ? 1.0f / (1.0f - p) : 0.0f;
}
// the four elements idx0 .. idx0 + 3 (idx0 a multiple of 4): dropout_scale of each, one draw
HULC_DEVICE void dropout_scale4(uint64_t seed, uint64_t idx0, float p, float (&s)[4]) {
    const uint32_t thr = (uint32_t)(p * 65536.0f);
    const float keep = 1.0f / (1.0f - p);
    const uint64_t z = hulc_rand64(seed, idx0 >> 2);
    const uint32_t lo = (uint32_t)z, hi = (uint32_t)(z >> 32);
    s[0] = (lo & 0xffffu) >= thr ? keep : 0.f; s[1] = (lo >> 16) >= thr ? keep : 0.f;
    s[2] = (hi & 0xffffu) >= thr ? keep : 0.f; s[3] = (hi >> 16) >= thr ? keep : 0.f;
}
// the 16 elements an MFMA accumulator lane holds along a row of consecutive indices: base + (e & 3) + 8 (e >> 2) + 4 hf — four runs of 4
HULC_DEVICE void dropout_scale_acc16(uint64_t seed, uint64_t base, int hf, float p, float (&s)[16]) {
    if ((base & 3) == 0) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float t[4];
            dropout_scale4(seed, base + 8 * g + 4 * hf, p, t);
            s[4 * g] = t[0]; s[4 * g + 1] = t[1]; s[4 * g + 2] = t[2]; s[4 * g + 3] = t[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) s[e] = dropout_scale(seed, base + (e & 3) + 8 * (e >> 2) + 4 * hf, p);
    }
}
HULC_DEVICE float hulc_uniform01(uint64_t seed, uint64_t idx) {
    return (hulc_rand32(seed, idx) >> 8) * (1.0f / 16777216.0f);
}
