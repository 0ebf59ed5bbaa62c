// mlp_chain.hip — a whole stack of Linear(+ReLU) layers on M <= 64 rows as ONE persistent launch (bf16 MFMA, fp32 accumulate).
//
// reference arithmetic: the per-sequence MLPs of the policy — PlanProposalNetwork.fc_model + fc_state (plan_proposal_net.py:26-47),
// Visual/LanguageGoalEncoder.mlp (goal_encoders.py:21-34,53-71), ProjVisLang (proj_vis_lang.py:10-21), the posterior's fc -> fc_state
// (plan_recognition_net.py:122-123,144-148) — and, with W^T and the stored activations as masks, the data-gradient chain of their backward.
//
// At 64 rows a 2048 x 2048 layer is 0.5 GFLOP and 8 MB of weights: as its own launch it costs ~10 us + a ~5 us split-K epilogue, all of it
// latency.  Here the launch is paid once per chain:
//   * 256 workgroups, one 16-column output tile of the current layer each (N / 16 tiles; idle workgroups still join the barriers);
//     4 waves split K, 16x16x32 MFMA tiles, a fixed-order LDS reduction, bias / ReLU / mask epilogue;
//   * the tile's weight fragments are requested BEFORE the wait for the previous layer (weights depend on nothing), so the barrier wait
//     hides their HBM latency;
//   * layer outputs are exchanged through L2 as bf16 in the blocked layout [k / 8][64 rows][8] (the 16 lanes of an MFMA k-block read 256
//     contiguous bytes), every layer into its OWN region, written with device-scope write-through stores and read with plain loads only
//     after the grid barrier (one arrival counter per XCD, 4 KB apart) — the exchange protocol of rnn_wavefront.hip;
//   * every layer's fp32 output also goes to its caller-visible buffer (what autograd saves / returns).
// Like the recurrent kernel it needs all its workgroups resident (one per CU): a barrier timeout sets the sticky fault word.
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

constexpr int CH_MAXL = 8;
constexpr int CH_CTR_STRIDE = 1024;                 // unsigned words between the 8 arrival counters (4 KB)
constexpr long CH_HEADER = 9L * CH_CTR_STRIDE * 4;   // 8 counters + the error word

struct ChLayer {
    const uint16_t* W; long ldw;        // bf16 [N][K], k contiguous
    const uint16_t* Wlo;                // split operands (X3): bf16 of the remainders w - bf16(w), same layout, or null
    const float* bias;                  // [N] or null
    const float* mask; long ld_mask; float mask_scale;   // out = mask > 0 ? v * mask_scale : 0   (data-gradient chain) or null
    float* out; long ld_out;            // fp32 [M][N]
    int N, K, relu;
    long xb_off;                        // element offset of this layer's OUTPUT exchange region inside xb
};
struct ChainP {
    ChLayer L[CH_MAXL];
    ChLayer L2[CH_MAXL];                // dual: the SECOND chain's layers (same depth and widths, its own weights / inputs / outputs)
    int dual, M2, nl2;                  // nl2 <= nl: the second chain may be shorter (it sits out the trailing layers); dual: rows 0..31 of every 64-row block belong to chain 1 (M <= 32), rows 32.. to chain 2 (M2 <= 32)
    const float* x0_2; long ld_x0_2; int K0_2;
    int nl, M;
    const float* x0; long ld_x0; int K0;    // layer 0 input, fp32 [M][K0] ...
    uint16_t* x0b;                      // ... and its bf16 blocked copy (chain_stage_input)
    uint16_t* xb;
    // split operands (X3): the "exact" chain — the second one of a pair, else the only one (then M <= 32) — forms its products from hi / lo
    // splits of both operands (three MFMAs): lo copies of its input and of its exchanged activations live in x0b_lo / xb_lo (same offsets)
    int x3; uint16_t* x0b_lo; uint16_t* xb_lo;
    unsigned* bar; int* err; int* err_sticky;
};

union F8 { uint4 u; bf16x8_t b; };
HULC_DEVICE constexpr int s_of(int bi, int q, int ab) { return bi * ab + q; }

// stage 0 of the chain (all workgroups): lay the fp32 input out the way every later layer finds its input — bf16, blocked
// [k / 8][64 rows][8] (rows >= M and columns >= K0 zero), written through to memory like the layers' exchange copies.  Each of the chain's
// 256 workgroups then reads it as 16-byte pieces; gathering the fp32 rows directly cost 256 x M x K0 x 4 bytes of strided L2 reads
// (74 us for a 4096-wide input).
HULC_DEVICE void chain_stage_input(const ChainP& p, int tid, int nwg) {
    const int kmax = (p.dual && p.K0_2 > p.K0) ? p.K0_2 : p.K0;
    const long nchunk = ((long)(kmax + 7) / 8) * 64;
    for (long i = (long)blockIdx.x * 256 + tid; i < nchunk; i += (long)nwg * 256) {
        const int m = (int)(i & 63), k = (int)(i >> 6) * 8;
        uint4 o = make_uint4(0u, 0u, 0u, 0u), ol = make_uint4(0u, 0u, 0u, 0u);
        const bool second = p.dual && m >= 32;
        const int mm = second ? m - 32 : m;
        if (mm < (second ? p.M2 : p.M) && k < (second ? p.K0_2 : p.K0)) {      // K0 is a multiple of 8
            const float* src = (second ? p.x0_2 + (long)mm * p.ld_x0_2 : p.x0 + (long)mm * p.ld_x0) + k;
            const float4 a = *(const float4*)src, b = *(const float4*)(src + 4);
            o = make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(b.x, b.y), pack_bf16x2(b.z, b.w));
            ol = make_uint4(pack_bf16x2(a.x - __uint_as_float(o.x << 16), a.y - __uint_as_float(o.x & 0xffff0000u)),
                            pack_bf16x2(a.z - __uint_as_float(o.y << 16), a.w - __uint_as_float(o.y & 0xffff0000u)),
                            pack_bf16x2(b.x - __uint_as_float(o.z << 16), b.y - __uint_as_float(o.z & 0xffff0000u)),
                            pack_bf16x2(b.z - __uint_as_float(o.w << 16), b.w - __uint_as_float(o.w & 0xffff0000u)));
        }
        unsigned long long* dst = (unsigned long long*)(p.x0b + i * 8);
        __hip_atomic_store(dst, (unsigned long long)o.x | ((unsigned long long)o.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(dst + 1, (unsigned long long)o.z | ((unsigned long long)o.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.x3 && (second || !p.dual)) {                           // the exact chain's remainders
            unsigned long long* dl = (unsigned long long*)(p.x0b_lo + i * 8);
            __hip_atomic_store(dl, (unsigned long long)ol.x | ((unsigned long long)ol.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(dl + 1, (unsigned long long)ol.z | ((unsigned long long)ol.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_fetch_add(p.bar + (blockIdx.x & 7) * CH_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// one layer for this workgroup's column tile; KSW = k-steps (of 32) per wave, K padded up to 128 * KSW
template <int KSW, bool X3>
HULC_DEVICE void chain_layer(const ChainP& p, int l, int tile, bool last, const uint16_t* __restrict__ xin, const uint16_t* __restrict__ xin_lo,
                             float (*red)[64][16], int tid, bool& timed_out, int nwg) {
    const ChLayer& c = p.L[l];
    const bool dual = p.dual && l < p.nl2;
    const ChLayer& c2 = dual ? p.L2[l] : p.L[l];                      // dual: the second chain's layer (row tiles 2, 3)
    const bool x3 = X3 && (p.dual ? dual : true);                     // (a pair's second chain may be shorter: no exact tiles in the trailing layers)
    const int lane = tid & 63, wave = tid >> 6, r = lane & 15, g = lane >> 4;
    const int ntile = c.N / 16;
    const bool active = tile < ntile;
    const int n0 = (active ? tile : 0) * 16;
    // ---- weight fragments of this wave's k range: requested before the wait on the previous layer
    F8 wf[KSW], wf2[KSW], wfl[X3 ? KSW : 1];
    {
        const uint16_t* wrow = c.W + (long)(n0 + r) * c.ldw;
        const uint16_t* wrow2 = c2.W + (long)(n0 + r) * c2.ldw;
        const uint16_t* wrowl = x3 ? c2.Wlo + (long)(n0 + r) * c2.ldw : nullptr;      // (c2 == c for a single chain)
#pragma unroll
        for (int s = 0; s < KSW; ++s) {
            const int k = (wave * KSW + s) * 32 + g * 8;
            wf[s].u = *(const uint4*)(wrow + (k < c.K ? k : 0));            // clamped (always valid) address; the matching A fragment is zero
            if (dual) wf2[s].u = *(const uint4*)(wrow2 + (k < c2.K ? k : 0));
            if (X3 && x3) wfl[s].u = *(const uint4*)(wrowl + (k < c2.K ? k : 0));
        }
    }
    // ---- wait for the previous stage's outputs (all workgroups): stage 0 = the input copy, stage l = layer l - 1
    {
        if (tid < 8) {
            const unsigned want = (unsigned)(nwg / 8) * (unsigned)(l + 1);
            const unsigned* bar = p.bar + tid * CH_CTR_STRIDE;
            long spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 22)) {
                    __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.err_sticky) __hip_atomic_fetch_or(p.err_sticky, 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // bit 1 = mlp_chain
                    break;
                }
            }
        }
        __syncthreads();
        asm volatile("" ::: "memory");
        // behind the LAST wait of the launch nobody looks at the arrival counters again: the workgroup that gets here last puts them
        // (and the count of workgroups that got here) back to zero — the header is clean for the next launch without a memset / prep launch
        if (last && tid == 0) {
            unsigned* done = p.bar + 8 * CH_CTR_STRIDE + 16;
            if (__hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)(nwg - 1)) {
#pragma unroll
                for (int x = 0; x < 8; ++x) __hip_atomic_store(p.bar + x * CH_CTR_STRIDE, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    // ---- products: 4 row tiles of 16, this wave's k-steps.  The A fragments are fetched a batch of k-steps ahead of the MFMAs that consume them
    // (two register sets): left to itself the compiler loads one k-step, waits a full L2 round trip, multiplies, and repeats — 16 round trips
    // per 2048-deep layer (16 us per layer measured)
    f32x4_t acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    const int MT = dual ? 2 + (p.M2 + 15) / 16 : (p.M + 15) / 16;     // (dual: tiles 0, 1 = chain 1, tiles 2, 3 = chain 2)
    const int MT1 = (p.M + 15) / 16;
    constexpr int AB = KSW >= 32 ? 2 : (KSW >= 4 ? 4 : KSW);            // k-steps per batch
    constexpr int NB = KSW / AB;
    static_assert(KSW % AB == 0, "batches tile the k range");
    F8 af[2][AB][4], afl[2][AB][X3 ? 2 : 1];                         // afl: lo fragments of the exact chain's two row tiles
    const int xt0 = dual ? 2 : 0;                                      // first row tile of the exact chain
    auto load_batch = [&](F8 (&dst)[AB][4], F8 (&dstl)[AB][X3 ? 2 : 1], int bi) {
#pragma unroll
        for (int q = 0; q < AB; ++q) {
            const int k = (wave * KSW + bi * AB + q) * 32 + g * 8;
            const bool kin = k < (c.K > c2.K ? c.K : c2.K);
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int m = mt * 16 + r;
                dst[q][mt].u = *(const uint4*)(xin + ((long)((kin ? k : 0) / 8) * 64 + m) * 8);
            }
            if (X3 && x3) {
#pragma unroll
                for (int t = 0; t < 2; ++t) dstl[q][t].u = *(const uint4*)(xin_lo + ((long)((kin ? k : 0) / 8) * 64 + (xt0 + t) * 16 + r) * 8);
            }
        }
    };
    if (active) {
        load_batch(af[0], afl[0], 0);
#pragma unroll
        for (int bi = 0; bi < NB; ++bi) {
            if (bi + 1 < NB) load_batch(af[(bi + 1) & 1], afl[(bi + 1) & 1], bi + 1);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < AB; ++q) {
                const int k = (wave * KSW + bi * AB + q) * 32 + g * 8;
                const bool kin1 = k < c.K, kin2 = k < c2.K;
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    F8 a = af[bi & 1][q][mt];
                    const bool sec = dual && mt >= 2;
                    if (!(sec ? kin2 : kin1)) a.u = make_uint4(0u, 0u, 0u, 0u);
                    const bool on = dual ? (mt < 2 ? mt < MT1 : mt < MT) : mt < MT;
                    if (on) acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, (sec ? wf2 : wf)[s_of(bi, q, AB)].b, acc[mt], 0, 0, 0);
                    if (X3 && x3 && on && (dual ? mt >= 2 : mt < 2)) {       // + a_lo w_hi + a_hi w_lo   (exact tiles 2, 3 of a pair / 0, 1 alone: slot mt & 1)
                        F8 al = afl[bi & 1][q][mt & 1];
                        if (!(sec ? kin2 : kin1)) al.u = make_uint4(0u, 0u, 0u, 0u);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al.b, (sec ? wf2 : wf)[s_of(bi, q, AB)].b, acc[mt], 0, 0, 0);
                        acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a.b, wfl[s_of(bi, q, AB)].b, acc[mt], 0, 0, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // ---- fixed-order reduction over the 4 waves + epilogue: 64 x 16 outputs, 4 per thread
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) red[wave][mt * 16 + g * 4 + e][r] = acc[mt][e];
    __syncthreads();
    if (active) {
        const int m = tid >> 2, n4 = (tid & 3) * 4;
        float v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] = (red[0][m][n4 + j] + red[1][m][n4 + j]) + (red[2][m][n4 + j] + red[3][m][n4 + j]);
        const bool sec = dual && m >= 32;                   // this row belongs to the second chain
        const ChLayer& e = sec ? c2 : c;
        const int rows = sec ? p.M2 : p.M, mm = sec ? m - 32 : m;
        const bool live = mm < rows;
        const int mc = live ? mm : rows - 1;
        if (e.bias) {
            const float4 b = *(const float4*)(e.bias + n0 + n4);
            v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
        }
        if (e.mask) {
            const float4 mk = *(const float4*)(e.mask + (long)mc * e.ld_mask + n0 + n4);
            v[0] = mk.x > 0.f ? v[0] * e.mask_scale : 0.f; v[1] = mk.y > 0.f ? v[1] * e.mask_scale : 0.f;
            v[2] = mk.z > 0.f ? v[2] * e.mask_scale : 0.f; v[3] = mk.w > 0.f ? v[3] * e.mask_scale : 0.f;
        } else if (e.relu) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fmaxf(v[j], 0.f);
        }
        if (live) *(float4*)(e.out + (long)mm * e.ld_out + n0 + n4) = make_float4(v[0], v[1], v[2], v[3]);
        if (l + 1 < (sec ? p.nl2 : p.nl) && live) {              // exchange copy for the next layer: [n / 8][64][8] bf16, device-scope write-through
            uint16_t* dst = p.xb + c.xb_off + ((long)((n0 + n4) / 8) * 64 + m) * 8 + (n4 & 7);
            const uint32_t h0 = pack_bf16x2(v[0], v[1]), h1 = pack_bf16x2(v[2], v[3]);
            const unsigned long long bits = (unsigned long long)h0 | ((unsigned long long)h1 << 32);
            __hip_atomic_store((unsigned long long*)dst, bits, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (X3 && x3 && (sec || !p.dual)) {                  // the exact chain also hands on the remainders of its activations
                const uint32_t l0 = pack_bf16x2(v[0] - __uint_as_float(h0 << 16), v[1] - __uint_as_float(h0 & 0xffff0000u));
                const uint32_t l1 = pack_bf16x2(v[2] - __uint_as_float(h1 << 16), v[3] - __uint_as_float(h1 & 0xffff0000u));
                __hip_atomic_store((unsigned long long*)(p.xb_lo + (dst - p.xb)), (unsigned long long)l0 | ((unsigned long long)l1 << 32), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    if (l + 1 < p.nl) {                          // arrival: everybody's outputs acknowledged before anybody may read them
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(p.bar + (blockIdx.x & 7) * CH_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        __syncthreads();
    }
    (void)timed_out;
}

// X3 = the split-operand instance (p.x3): its own kernel, so that the plain one keeps its register allocation (no scratch)
template <bool X3>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void mlp_chain_kernel(ChainP p) {
    __shared__ float red[4][64][16];
    const int tid = threadIdx.x;
    bool timed_out = false;
    const int nwg = gridDim.x, tile = blockIdx.x;                         // N <= 16 * gridDim: at most one tile per workgroup and layer
    chain_stage_input(p, tid, nwg);
    for (int l = 0; l < p.nl; ++l) {
        const int ksw = ((p.dual && l < p.nl2 && p.L2[l].K > p.L[l].K ? p.L2[l].K : p.L[l].K) + 127) / 128;
        const uint16_t* xin = l ? p.xb + p.L[l - 1].xb_off : p.x0b;
        const uint16_t* xin_lo = l ? p.xb_lo + p.L[l - 1].xb_off : p.x0b_lo;
#define CH_CASE(N) case N: chain_layer<N, X3>(p, l, tile, l + 1 == p.nl, xin, xin_lo, red, tid, timed_out, nwg); break;
        switch (ksw) {
            CH_CASE(1) CH_CASE(2) CH_CASE(3) CH_CASE(4) CH_CASE(8) CH_CASE(16)
            default: if constexpr (!X3) chain_layer<32, false>(p, l, tile, l + 1 == p.nl, xin, xin_lo, red, tid, timed_out, nwg);   // (X3: K <= 2048, host-checked)
                     break;
        }
#undef CH_CASE
    }
}

}  // namespace

extern "C" long hulc_mlp_chain_workspace(const hulc_mlp_chain_desc* d) {
    if (!d || d->nl < 1 || d->nl > CH_MAXL) return 0;
    long elems = 0;
    for (int l = 0; l + 1 < d->nl; ++l) elems += (long)d->layers[l].N * 64;
    // header, exchange regions, the blocked input — twice: the split-operand mode keeps remainder copies of both
    return CH_HEADER + 2 * (elems * 2 + 64 + ((long)(d->K0 + 7) / 8) * 64 * 16);
}

// validates one chain description and fills its device-side layer table; returns 0 or an error code
static int chain_fill(const hulc_mlp_chain_desc* d, ChLayer* L, int max_rows) {
    if (d->nl < 1 || d->nl > CH_MAXL) return hulc_fail(-2, "hulc_mlp_chain: 1..8 layers");
    if (d->M < 1 || d->M > max_rows) return hulc_fail(-2, "hulc_mlp_chain: 1 <= M <= 64 rows (32 per chain of a pair)");
    if (!d->x0 || (uintptr_t)d->x0 % 16 || d->ld_x0 % 4) return hulc_fail(-4, "hulc_mlp_chain: input must be 16-byte aligned");
    long off = 0;
    int kin = d->K0;
    for (int l = 0; l < d->nl; ++l) {
        const hulc_mlp_chain_layer& s = d->layers[l];
        if (!s.W || !s.out) return hulc_fail(-1, "hulc_mlp_chain: null layer operand");
        if (s.N < 16 || s.N % 16 || s.N > 4096 || kin < 8 || kin % 8 || kin > 4096 || s.ldw % 8 || (uintptr_t)s.W % 16 || s.ld_out % 4 || (uintptr_t)s.out % 16)
            return hulc_fail(-3, "hulc_mlp_chain: N must be a multiple of 16, K a multiple of 8 and <= 4096, operands 16-byte aligned");
        const int ksw = (kin + 127) / 128;
        if (!(ksw == 1 || ksw == 2 || ksw == 3 || ksw == 4 || ksw == 8 || ksw == 16 || ksw == 32))
            return hulc_fail(-3, "hulc_mlp_chain: K must round up to 128 x {1, 2, 3, 4, 8, 16, 32}");
        if (s.bias && (uintptr_t)s.bias % 16) return hulc_fail(-4, "hulc_mlp_chain: bias must be 16-byte aligned");
        if (s.mask && ((uintptr_t)s.mask % 16 || s.ld_mask % 4)) return hulc_fail(-4, "hulc_mlp_chain: mask must be 16-byte aligned");
        ChLayer& c = L[l];
        if (s.W_lo && (uintptr_t)s.W_lo % 16) return hulc_fail(-4, "hulc_mlp_chain: W_lo must be 16-byte aligned");
        c.W = (const uint16_t*)s.W; c.Wlo = (const uint16_t*)s.W_lo; c.ldw = s.ldw; c.bias = s.bias; c.mask = s.mask; c.ld_mask = s.ld_mask; c.mask_scale = s.mask_scale;
        c.out = s.out; c.ld_out = s.ld_out; c.N = s.N; c.K = kin; c.relu = s.relu; c.xb_off = off;
        off += (long)s.N * 64;
        kin = s.N;
    }
    return 0;
}

static int chain_launch(const hulc_mlp_chain_desc* d, const hulc_mlp_chain_desc* d2, void* ws, int* err_sticky, void* stream) {
    if (!d || !ws) return hulc_fail(-1, "hulc_mlp_chain: null pointer");
    if ((uintptr_t)ws % 16) return hulc_fail(-4, "hulc_mlp_chain: workspace must be 16-byte aligned");
    ChainP p = {};
    int rc = chain_fill(d, p.L, d2 ? 32 : 64);
    if (rc) return rc;
    p.nl = d->nl; p.M = d->M; p.x0 = d->x0; p.ld_x0 = d->ld_x0; p.K0 = d->K0;
    if (d2) {
        rc = chain_fill(d2, p.L2, 32);
        if (rc) return rc;
        if (d2->nl > d->nl) return hulc_fail(-2, "hulc_mlp_chain2: the first chain is the deeper one");
        for (int l = 0; l < d2->nl; ++l)
            if (d2->layers[l].N != d->layers[l].N) return hulc_fail(-2, "hulc_mlp_chain2: the chains have the same layer widths where both run");
        p.dual = 1; p.nl2 = d2->nl; p.M2 = d2->M; p.x0_2 = d2->x0; p.ld_x0_2 = d2->ld_x0; p.K0_2 = d2->K0;
        const int k1 = (d->K0 + 127) / 128, k2 = (d2->K0 + 127) / 128, km = k1 > k2 ? k1 : k2;   // layer 0 runs at the larger input width
        if (!(km == 1 || km == 2 || km == 3 || km == 4 || km == 8 || km == 16 || km == 32)) return hulc_fail(-3, "hulc_mlp_chain2: K0 must round up to 128 x {1, 2, 3, 4, 8, 16, 32}");
    }
    p.bar = (unsigned*)ws; p.err = (int*)((char*)ws + 8L * CH_CTR_STRIDE * 4); p.err_sticky = err_sticky;
    p.xb = (uint16_t*)((char*)ws + CH_HEADER);
    long off = 0;
    for (int l = 0; l + 1 < d->nl; ++l) off += (long)d->layers[l].N * 64;
    p.x0b = p.xb + ((off + 7) / 8) * 8;                      // behind the exchange regions of layers 0 .. nl-2
    {   // split operands: the exact chain (the second of a pair, else the only one) carries W_lo on every layer
        const hulc_mlp_chain_desc* ex = d2 ? d2 : d;
        int nlo = 0;
        for (int l = 0; l < ex->nl; ++l) nlo += ex->layers[l].W_lo != nullptr;
        if (nlo != 0 && nlo != ex->nl) return hulc_fail(-3, "hulc_mlp_chain: W_lo on every layer of the chain or on none");
        if (nlo && !d2 && d->M > 32) return hulc_fail(-2, "hulc_mlp_chain: split operands take <= 32 rows");
        if (d2) for (int l = 0; l < d->nl; ++l) if (d->layers[l].W_lo) return hulc_fail(-3, "hulc_mlp_chain2: split operands are the second chain's");
        p.x3 = nlo != 0;
        const int k1 = d->K0, k2 = d2 ? d2->K0 : 0, km = k1 > k2 ? k1 : k2;
        const long data = ((off + 7) / 8) * 8 + ((long)(km + 7) / 8) * 64 * 8 + 64;    // elements of one copy (exchange regions + blocked input)
        p.xb_lo = p.xb + data;
        p.x0b_lo = p.xb_lo + ((off + 7) / 8) * 8;
    }
    // grid: a layer of N columns keeps N / 16 workgroups busy — a chain no wider than 2048 runs on 128 workgroups (the others of a 256-grid only
    // take part in the barriers), which leaves the other half of the device to a second cooperative launch (round 6: g_coop_share)
    int widest = 0;
    for (int l = 0; l < d->nl; ++l) widest = d->layers[l].N > widest ? d->layers[l].N : widest;
    int grid = 256;
    {
        static const char* e = getenv("HULC_CHAIN_GRID");
        const int want = e ? atoi(e) : (hulc_coop_share() > 1 ? 256 / hulc_coop_share() : 256);
        if (want >= 8 && want < 256 && want % 8 == 0 && widest <= 16 * want) grid = want;
        else if (hulc_coop_share() > 1) return hulc_fail(-9, "hulc_mlp_chain: the chain is wider than its share of the device (hulc_set_coop_share)");
    }
    if (p.x3) {
        for (int l = 0; l < d->nl; ++l) {
            const int kk = d2 && l < d2->nl && p.L2[l].K > p.L[l].K ? p.L2[l].K : p.L[l].K;
            if (kk > 2048) return hulc_fail(-3, "hulc_mlp_chain: split operands take K <= 2048");
        }
        mlp_chain_kernel<true><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    } else {
        mlp_chain_kernel<false><<<grid, 256, 0, (hipStream_t)stream>>>(p);
    }
    return hulc_check_launch("hulc_mlp_chain");
}

// see include/hulc2_amd.h
extern "C" int hulc_mlp_chain(const hulc_mlp_chain_desc* d, void* ws, int* err_sticky, void* stream) {
    return chain_launch(d, nullptr, ws, err_sticky, stream);
}

// Two INDEPENDENT chains (their own weights, inputs of possibly different width, outputs, <= 32 rows each; b no deeper than a, and the same
// layer widths on the layers both run) as one launch: the visual and the language goal encoder (goal_encoders.py:21-34 / :53-71) and their data-gradient chains.  A workgroup's
// column tile multiplies row tiles 0-1 by the first chain's weight fragments and row tiles 2-3 by the second's; a launch (and a device-wide
// barrier per layer) is paid once for both.  ws: hulc_mlp_chain_workspace(a) + hulc_mlp_chain_workspace(b) bytes.
extern "C" int hulc_mlp_chain2(const hulc_mlp_chain_desc* a, const hulc_mlp_chain_desc* b, void* ws, int* err_sticky, void* stream) {
    if (!a || !b) return hulc_fail(-1, "hulc_mlp_chain2: null pointer");
    return chain_launch(a, b, ws, err_sticky, stream);
}
