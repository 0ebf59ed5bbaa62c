// rnn_wavefront.hip — the 2-layer ReLU RNN of the action decoder as ONE persistent kernel per direction (bf16 compute).
//
// reference arithmetic: nn.RNN(num_layers=2, nonlinearity="relu", batch_first=True) of
// hulc2/models/decoders/logistic_decoder_rnn.py:70-79 (forward) and its autograd backward.
//
// The recurrence is latency-bound: per time step two skinny GEMMs (64 rows, K = 2048 / 4096) whose 25 MB of weights were
// re-streamed by ~130 launches per direction (12 us kernel + 4 us split-K epilogue each).  With
//     z_t = [h0_t | h1_{t-1}]  (one row block of the time-major state buffer)
// both layers advance together as a wavefront:     z_{t+1} = f( z_t W^T + add_t ),   W = [[W_hh0, 0], [W_ih1, W_hh1]]
// and the backward pass has the same shape on d_t = [delta1_t | delta0_{t+1}] with W = [[W_hh1^T, 0], [W_ih1^T, W_hh0^T]].
// So one kernel serves both directions:
//   * 256 workgroups (one per CU) = 2 row halves x H/16 column groups; a workgroup owns 16 "first" + 16 "second" output
//     columns for 32 batch rows and keeps its 16+16 weight columns (x K = 2H) in REGISTERS for all S+1 wave steps
//     (96 VGPRs per wave) — weights are read from HBM once per pass instead of once per step;
//   * per wave step a workgroup reads its 32 rows of the bf16 state copy (256 KB, from L2), runs v_mfma_f32_16x16x32_bf16
//     with the 8 waves splitting K, sums the 8 partial tiles through LDS in a fixed order (deterministic), applies the
//     epilogue (bias / external add / ReLU or stored-activation mask) and writes fp32 (for the weight-gradient GEMMs) and
//     bf16 (its own next operand, ping-pong) state;
//   * wave steps are separated by a device-wide barrier (agent-scope counters, one per row half and XCD).  All 256 workgroups
//     are co-resident (1 per CU by register footprint; the stream runs nothing else), the spin is bounded, and a timeout
//     poisons the output with NaN instead of hanging the GPU.
#include <cstdio>
#include <vector>
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

#define RNN_CTR_STRIDE 1024                                   // unsigned words between barrier counters (4 KB)
#define RNN_WS_HEADER (17 * RNN_CTR_STRIDE * 4)               // 16 barrier counters (8 in the unpipelined kernel) + the error word
#define RNN_ERR_WORD (16 * RNN_CTR_STRIDE)

namespace {

struct WaveP {
    float* z; long z_step;                       // fp32 state rows: wave step tau reads z + tau*z_step, writes z + (tau+1)*z_step
    uint16_t* zb;                                // bf16 mirror of the state rows, [S+2][B][2H] row-major (operand of the weight-gradient GEMMs)
    uint16_t* zt; long ld_t;                     // optional transposed mirror [2H][(S+2)*B] (token = region*B + row): k-major operand of the weight-gradient GEMMs
    uint16_t* xb;                                // the same values in the exchange layout the kernel itself reads, [S+2][2][H/8][64][8] (see below)
    int zb_row0, zb_dir;                         // region read at wave step tau = zb_row0 + tau * zb_dir (0, +1 forward; S+1, -1 reversed)
    const uint16_t *wA, *wB1, *wB2;              // first (H x H), second k < H (H x H), second k >= H (H x H)
    long ldA, ldB1, ldB2; int tA, tB1, tB2;      // t: element (n, k) is w[k*ld + n] instead of w[n*ld + k]
    const float* add1; long add1_step, ld_add1;  // per-step external term of the first half (nullable)
    const float *bias1a, *bias1b, *bias2a, *bias2b;
    const float* mask1; long mask1_step, ld_mask1;   // stored activations: output kept where mask > 0 (nullable)
    const float* mask2; long mask2_step, ld_mask2;
    int relu, S, B, H;
    unsigned* bar; int* err; int* err_sticky;
    const float* add1c; long ld_add1c;           // per-row constant of the first half, the same at every step (nullable): folded into the bias term
    unsigned long long* ts;                      // HULC_RNN_DBG & 8: s_memrealtime stamps [workgroup 0 / 100][wave 1, 0, 7][sub-step][7 phases] (printed by the next launch)
    int dbg;                                     // experiments / tests only (HULC_RNN_DBG): 1 = skip state loads + MFMAs, 2 = skip the barrier, 4 = inject a barrier timeout
};

// The bf16 state copy is the only data exchanged between workgroups inside the kernel.  Measured alternatives:
//   * agent-scope release/acquire fences around the barrier: buffer_wbl2 / buffer_inv from 2048 waves per step, ~90 us/step;
//   * device-coherent (sc1) loads and stores on a ping-pong buffer: no cache maintenance, but every read bypasses the
//     per-XCD L2 -> 64 MB per step from memory, 14 us/step;
//   * (this) every wave step writes its OWN region of the copy with write-through sc1 stores and readers use ordinary
//     loads: a region's addresses are never cached before they are complete (first touch after the barrier; the caches
//     start clean at kernel launch), so the 128 workgroups sharing a row half hit the XCD's L2 instead of memory.
// Exchange layout: [region][state half][k / 8][row 0..63][8 values].  An MFMA A fragment is (row = lane % 16, 8 consecutive k
// at k-block lane / 16): in a row-major copy the 16 lanes of a k-block sit 8 KB apart, every lane is its own tag lookup and the
// 256 KB a workgroup reads per step took ~6 us (18 B/clk per CU).  Blocked, the 16 lanes read 256 contiguous bytes, an
// instruction touches 8 full lines instead of 64 partial ones, and every 128-byte line has exactly one writer workgroup.
HULC_DEVICE bf16x8_t load_state8(const uint16_t* p) { return *(const bf16x8_t*)p; }

// T: element (n, k) is w[k*ld + n] instead of w[n*ld + k] — compile-time, so that the 24 fragment loads of a wave are issued back to
// back (a run-time branch around a load makes hipcc wait for it at the join: 24 serial round trips at the start of every pass)
template <bool T>
HULC_DEVICE bf16x8_t load_w(const uint16_t* w, long ld, int n, int k) {
    union { uint4 u; bf16x8_t b; uint16_t h[8]; } x;
    if (!T) x.u = *(const uint4*)(w + (long)n * ld + k);
    else {
#pragma unroll
        for (int j = 0; j < 8; ++j) x.h[j] = w[(long)(k + j) * ld + n];
    }
    return x.b;
}

template <int H, bool WT>
__global__ __launch_bounds__(512) void rnn_wavefront_kernel(WaveP p) {
    constexpr int KS = H / 32;                   // k-steps (of 32) per half of the state row
    constexpr int KPW = KS / 8;                  // per wave
    static_assert(KS % 8 == 0, "H must be a multiple of 256");
    __shared__ float red[8][4][256];             // [wave][mt*2 + ct][reg*64 + lane]
    __shared__ uint4 wlds[8][KPW][64];           // second-half weights of the "second" columns (the registers hold the rest)
    __shared__ __attribute__((aligned(16))) uint16_t otile[32][2][16 + 8];    // bf16 outputs of the step, [row][half][col] (+pad), for 16-byte coherent stores
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kb = lane >> 4;
    // workgroups are dispatched round-robin over the 8 XCDs: xcd = blockIdx % 8.  64 "line groups" (row half x 64 columns)
    // of 4 workgroups each; line group L lives on XCD L % 8.
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lgrp = xcd + 8 * (slot >> 2);
    const int rowhalf = lgrp & 1, n0 = ((lgrp >> 1) * 4 + (slot & 3)) * 16;
    const int nblk = gridDim.x;
    const long ldz = 2 * H;

    // ---- resident weights: B-operand fragments (lane = column n0 + i, 8 consecutive k)
    bf16x8_t wfa[KPW], wfb[KPW];
#pragma unroll
    for (int q = 0; q < KPW; ++q) {
        const int k = (wave * KPW + q) * 32 + kb * 8;
        wfa[q] = load_w<WT>(p.wA, p.ldA, n0 + i, k);
        wfb[q] = load_w<WT>(p.wB1, p.ldB1, n0 + i, k);
        union { bf16x8_t b; uint4 u; } wc; wc.b = load_w<WT>(p.wB2, p.ldB2, n0 + i, k);
        wlds[wave][q][lane] = wc.u;                                          // read back only by this wave: no barrier needed
    }
    bool timed_out = false;
    // ---- this thread's two outputs per wave step (fixed): tile t4 = mt*2 + ct, accumulator element e
    int om[2], on[2]; float obias[2];
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
        const int o = tid + rep * 512, t4 = o >> 8, e = o & 255, ln = e & 63;
        om[rep] = rowhalf * 32 + (t4 >> 1) * 16 + 4 * (ln >> 4) + (e >> 6);
        on[rep] = n0 + (ln & 15);
        const float* ba = (t4 & 1) ? p.bias2a : p.bias1a;
        const float* bb = (t4 & 1) ? p.bias2b : p.bias1b;
        obias[rep] = (ba ? ba[on[rep]] : 0.f) + (bb ? bb[on[rep]] : 0.f);
        if (!(t4 & 1) && p.add1c) obias[rep] += p.add1c[(long)(om[rep] < p.B ? om[rep] : p.B - 1) * p.ld_add1c + on[rep]];
    }

    for (int tau = 0; tau <= p.S; ++tau) {
        const bool first_on = tau < p.S, second_on = tau >= 1;
        f32x4_t acc[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) acc[mt][ct] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        // epilogue operands first: their latency hides behind the state loads and the MFMAs
        float oadd[2], omask[2];
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int ct = ((tid + rep * 512) >> 8) & 1;
            const int m = om[rep] < p.B ? om[rep] : p.B - 1, n = on[rep];
            oadd[rep] = 0.f; omask[rep] = 1.f;
            if (ct == 0) {
                if (p.add1 && first_on) oadd[rep] = p.add1[(long)tau * p.add1_step + (long)m * p.ld_add1 + n];
                if (p.mask1 && first_on) omask[rep] = p.mask1[(long)tau * p.mask1_step + (long)m * p.ld_mask1 + n];
            } else if (p.mask2 && second_on) omask[rep] = p.mask2[(long)tau * p.mask2_step + (long)m * p.ld_mask2 + n];
        }
        if (tau > 0 && !(p.dbg & 1)) {                       // z_0 = 0: nothing to multiply
            const uint16_t* a = p.xb + (long)(p.zb_row0 + tau * p.zb_dir) * 64 * ldz + (long)(rowhalf * 32 + i) * 8;
            // the step's 32 fragment loads are independent of the MFMAs: the scheduler keeps as many in flight as registers allow
            bf16x8_t af[2][KPW][2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int q = 0; q < KPW; ++q)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        af[hf][q][mt] = load_state8(a + ((long)(hf * (H / 8) + (wave * KPW + q) * 4 + kb) * 64 + mt * 16) * 8);
#pragma unroll
            for (int q = 0; q < KPW; ++q)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    acc[mt][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][q][mt], wfa[q], acc[mt][0], 0, 0, 0);
                    acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][q][mt], wfb[q], acc[mt][1], 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < KPW; ++q) {
                union { bf16x8_t b; uint4 u; } wc; wc.u = wlds[wave][q][lane];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][q][mt], wc.b, acc[mt][1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wave][mt * 2 + ct][e * 64 + lane] = acc[mt][ct][e];
        __syncthreads();

        // ---- fixed-order sum over the 8 K slices + epilogue; 1024 outputs, 2 per thread
        float* zn = p.z + (long)(tau + 1) * p.z_step;
        uint16_t* zbn = p.zb + (long)(p.zb_row0 + (tau + 1) * p.zb_dir) * p.B * ldz;
        uint16_t* xbn = p.xb + (long)(p.zb_row0 + (tau + 1) * p.zb_dir) * 64 * ldz;
        float vout[2]; bool vst[2];
#pragma unroll
        for (int rep = 0; rep < 2; ++rep) {
            const int o = tid + rep * 512, t4 = o >> 8, e = o & 255, ct = t4 & 1;
            const int m = om[rep], n = on[rep];
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w][t4][e];
            const bool on_ = ct == 0 ? first_on : second_on;
            vst[rep] = m < p.B && !(ct == 0 && !on_);                    // last wave step: the first half is not produced
            v += oadd[rep] + obias[rep];
            if ((ct == 0 ? p.mask1 : p.mask2) != nullptr) v = omask[rep] > 0.f ? v : 0.f;
            else if (p.relu) v = fmaxf(v, 0.f);
            if (!on_) v = 0.f;                                           // wave step 0: the second half (h1_{-1}) is zero
            vout[rep] = v;
            if (vst[rep]) otile[m - rowhalf * 32][ct][n - n0] = f32_to_bf16_bits(v);
        }
        __syncthreads();
        // the exchange copy first — it is all the other workgroups wait for; the fp32 rows and the row-major mirror are only read
        // after the kernel and are stored behind the barrier arrival, off the critical path
        uint4 v16 = make_uint4(0, 0, 0, 0); bool st16 = false; long off16 = 0;
        if (tid < 128) {                                     // 32 rows x 2 halves x 2 chunks of 8 columns = 128 16-byte pieces
            const int row = tid & 31, ct = tid >> 6, ch = (tid >> 5) & 1;
            const int m = rowhalf * 32 + row;
            st16 = m < p.B && (ct == 1 || first_on);
            if (st16) {
                v16 = *(const uint4*)&otile[row][ct][ch * 8];
                off16 = (long)m * ldz + ct * H + n0 + ch * 8;
                unsigned long long* dst = (unsigned long long*)(xbn + ((long)(ct * (H / 8) + n0 / 8 + ch) * 64 + m) * 8);
                __hip_atomic_store(dst, (unsigned long long)v16.x | ((unsigned long long)v16.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, (unsigned long long)v16.z | ((unsigned long long)v16.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        const bool last = tau == p.S;
        if (!last) {
            // ---- device-wide barrier, arrival: everybody's z_{tau+1} is visible before anybody reads it
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // this wave's coherent stores have been acknowledged
            __syncthreads();
            if (tid == 0 && !(p.dbg & 2)) __hip_atomic_fetch_add(p.bar + xcd * RNN_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int rep = 0; rep < 2; ++rep)
            if (vst[rep]) zn[(long)om[rep] * ldz + (((tid + rep * 512) >> 8) & 1) * H + on[rep]] = vout[rep];
        if (st16) *(uint4*)(zbn + off16) = v16;                               // row-major mirror
        if (p.zt && tid < 128) {                                              // transposed mirror: (half, column, 8 rows) per thread
            const int ct = tid >> 6, col = (tid >> 2) & 15, ch = tid & 3;
            const int m = rowhalf * 32 + ch * 8;
            if (m < p.B && (ct == 1 || first_on)) {                           // B % 8 == 0 (checked by the launcher)
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (unsigned)otile[ch * 8 + 2 * e][ct][col] | ((unsigned)otile[ch * 8 + 2 * e + 1][ct][col] << 16);
                *(uint4*)(p.zt + (long)(ct * H + n0 + col) * p.ld_t + (long)(p.zb_row0 + (tau + 1) * p.zb_dir) * p.B + m) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        if (last) break;

        if (tid < 4 && !(p.dbg & 2)) {
            // The two row halves never exchange data, and a half lives on four XCDs (xcd parity = row half): one arrival counter per
            // (half, XCD), 4 KB apart (separate memory channels); a workgroup adds to its own and lanes 0..3 each poll one of its
            // half's four counters.  One counter for all 256 workgroups cost 13.4 us per wave step, one per half 11.1, this 10.5:
            // same-address atomics and 256 pollers on one line serialise at the memory side.
            const unsigned per = (unsigned)(nblk / 8) * (unsigned)(tau + 1);
            const unsigned* bar = p.bar + ((xcd & 1) + 2 * tid) * RNN_CTR_STRIDE;
            long spins = 0;
            while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 22)) {
                    __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if (p.err_sticky) __hip_atomic_fetch_or(p.err_sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // bit 0 = rnn_wavefront
                    break;
                }
            }
        }
        __syncthreads();
        asm volatile("" ::: "memory");
    }
    // a barrier timeout anywhere poisons the state so the failure is loud (NaN loss) instead of silent
    __syncthreads();
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0) {        // fault injection (tests): behave as if the barrier had timed out
        __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.err_sticky) __hip_atomic_fetch_or(p.err_sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // bit 0 = rnn_wavefront
    }
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) timed_out = true;
    if (timed_out) {
        float* zn = p.z + (long)(p.S + 1) * p.z_step;
        for (int o = tid; o < 32 * 16; o += 512) {
            const int m = rowhalf * 32 + o / 16, n = n0 + o % 16;
            if (m < p.B) { zn[(long)m * ldz + n] = __builtin_nanf(""); zn[(long)m * ldz + H + n] = __builtin_nanf(""); }
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------------------
// The same sweep, software-pipelined over two independent row groups (round 3).
//
// A wave step of the kernel above is a dependent chain: state loads + MFMA (2.8 us) -> 8-wave sum + epilogue -> write-through exchange stores ->
// their acknowledgement -> barrier arrival -> barrier propagation (1.3 us) -> next state loads: 6.5-7 us, the matrix pipes busy for 0.3 of them.
// Sequences (batch rows) are independent, so the 32 rows a workgroup owns are split into two groups of 16 (the two MFMA row tiles) with their OWN
// barrier counters, and the workgroup alternates between them: while group A's stores travel and its barrier collects the other workgroups,
// group B's state is fetched, multiplied and reduced.  Per sub-step roles (no wave waits for something it does not need):
//   wave 0      the exchange stores of the group just finished (64 lanes x 16 bytes = the whole 16 x 32 tile) and — one sub-step later, behind
//               the NEXT group's state loads — `s_waitcnt vmcnt(16)`: vector memory returns in order, so with 16 younger loads outstanding the
//               older stores have been acknowledged; then lane 0 bumps the group's barrier counter.  Nobody else waits for the acknowledgement.
//   waves 1-2   fp32 rows (read only after the kernel), wave 3 the row-major bf16 mirror, wave 4 the transposed mirror — from LDS tiles
//   wave 7      lanes 0-3 poll the four counters of (row half, group); it owns no stores, so its polling loads wait for nothing but themselves
//               (a polling load behind an un-acknowledged store would wait for that store first)
// Round 4, phase stamps (HULC_RNN_DBG=8, the TS instance): a sub-step of 2.75 us = poll + barrier 0.3, state loads + MFMAs 1.35 (first wave) ... 1.85 (last
// wave: 128 KB per CU at 29 B/clk), 8-wave sum + epilogue 0.22, stores 0.13-0.23.  Reading the NEXT sub-step's counters at the end of this one (to save the
// round trip at the top) made the pass 11 % slower: the counters are not complete yet at that point — a group's chain (stores -> acknowledgement ->
// counter -> propagation -> loads -> MFMAs -> sum) is as long as the two sub-steps it has; the kernel is bound by that chain, not by throughput.
template <int H, bool WT, bool TS = false>      // TS: the probe's instance with phase time stamps (HULC_RNN_DBG & 8) — the stamps' branches cost the plain kernel 5 %
__global__ __launch_bounds__(512) void rnn_wavefront2_kernel(WaveP p) {
    constexpr int KS = H / 32;
    constexpr int KPW = KS / 8;
    static_assert(KS % 8 == 0, "H must be a multiple of 256");
    __shared__ float red[8][2][256];             // [wave][ct][reg*64 + lane] partial tiles of the current group
    __shared__ uint4 wlds[8][KPW][64];
    __shared__ __attribute__((aligned(16))) uint16_t otile[16][2][16 + 8];   // bf16 outputs of the sub-step, [row][half][col] (+pad)
    __shared__ __attribute__((aligned(16))) float ftile[16][2][16 + 4];      // the same values in fp32
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = lane & 15, kb = lane >> 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int lgrp = xcd + 8 * (slot >> 2);
    const int rowhalf = lgrp & 1, n0 = ((lgrp >> 1) * 4 + (slot & 3)) * 16;
    const int nblk = gridDim.x;
    const long ldz = 2 * H;

    bf16x8_t wfa[KPW], wfb[KPW];
#pragma unroll
    for (int q = 0; q < KPW; ++q) {
        const int k = (wave * KPW + q) * 32 + kb * 8;
        wfa[q] = load_w<WT>(p.wA, p.ldA, n0 + i, k);
        wfb[q] = load_w<WT>(p.wB1, p.ldB1, n0 + i, k);
        union { bf16x8_t b; uint4 u; } wc; wc.b = load_w<WT>(p.wB2, p.ldB2, n0 + i, k);
        wlds[wave][q][lane] = wc.u;
    }
    // this thread's output of a sub-step: tile ct = tid >> 8, accumulator element e
    const int oct = tid >> 8, oe = tid & 255, oln = oe & 63;
    const int orow = 4 * (oln >> 4) + (oe >> 6);                  // row inside the 16-row group
    const int on = n0 + (oln & 15);
    float obias[2];                                               // per group (the per-row constant differs)
    {
        const float* ba = oct ? p.bias2a : p.bias1a;
        const float* bb = oct ? p.bias2b : p.bias1b;
        const float b0 = (ba ? ba[on] : 0.f) + (bb ? bb[on] : 0.f);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            const int m = rowhalf * 32 + g * 16 + orow;
            obias[g] = b0;
            if (!oct && p.add1c) obias[g] += p.add1c[(long)(m < p.B ? m : p.B - 1) * p.ld_add1c + on];
        }
    }
    int pending = -1;                                             // wave 0: group whose exchange stores await their acknowledgement
    const int U = 2 * (p.S + 1);
    // (probe) phase stamps of workgroups 0 and 100, waves 1 (an ordinary wave), 0 (exchange stores) and 7 (polls)
    const int ts_wg = blockIdx.x == 0 ? 0 : (blockIdx.x == 100 ? 1 : -1), ts_wv = wave == 1 ? 0 : (wave == 0 ? 1 : (wave == 7 ? 2 : -1));
    unsigned long long* ts = (TS && p.ts && ts_wg >= 0 && ts_wv >= 0 && lane == 0) ? p.ts + (long)((ts_wg * 3 + ts_wv) * 80) * 8 : nullptr;
#define RNN_TS(u_, ph_) if constexpr (TS) { if (ts && (u_) < 80) ts[(u_) * 8 + (ph_)] = __builtin_amdgcn_s_memrealtime(); }
    for (int u = 0; u < U; ++u) {
        const int g = u & 1, tau = u >> 1;
        RNN_TS(u, 0)
        const bool first_on = tau < p.S, second_on = tau >= 1;
        const int om = rowhalf * 32 + g * 16 + orow;
        if (tau > 0) {
            // ---- inputs of (g, tau): every workgroup of this row half has published the group's z_tau
            if (wave == 7 && lane < 4 && !(p.dbg & 2)) {
                const unsigned per = (unsigned)(nblk / 8) * (unsigned)tau;
                const unsigned* bar = p.bar + (g * 8 + (xcd & 1) + 2 * lane) * RNN_CTR_STRIDE;
                long spins = 0;
                while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < per) {
                    __builtin_amdgcn_s_sleep(1);
                    if (++spins > (1L << 22)) {
                        __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        if (p.err_sticky) __hip_atomic_fetch_or(p.err_sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __syncthreads();
            asm volatile("" ::: "memory");
        }
        RNN_TS(u, 1)
        f32x4_t acc[2] = {f32x4_t{0.f, 0.f, 0.f, 0.f}, f32x4_t{0.f, 0.f, 0.f, 0.f}};
        const bool mul = tau > 0 && !(p.dbg & 1);
        bf16x8_t af[2][KPW];
        if (mul) {
            const uint16_t* a = p.xb + (long)(p.zb_row0 + tau * p.zb_dir) * 64 * ldz + (long)(rowhalf * 32 + g * 16 + i) * 8;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int q = 0; q < KPW; ++q) af[hf][q] = load_state8(a + ((long)(hf * (H / 8) + (wave * KPW + q) * 4 + kb) * 64) * 8);
        }
        if (wave == 0 && pending >= 0) {
            // the previous sub-step's exchange stores are OLDER than everything this wave has issued since (compiler barrier after them): once at
            // most the 2 * KPW state loads above are outstanding they have been acknowledged — the other workgroups may read them
            asm volatile("" ::: "memory");
            if (mul) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * KPW) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0 && !(p.dbg & 2)) __hip_atomic_fetch_add(p.bar + (pending * 8 + xcd) * RNN_CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            pending = -1;
        }
        // epilogue operands: requested behind the state loads (their latency hides under the MFMAs) and consumed as opaque values in the
        // epilogue — hipcc otherwise hoists the `mask > 0` compare to the load and waits for it (and for everything older) right here
        float oadd = 0.f, omask = 1.f;
        {
            const int m = om < p.B ? om : p.B - 1;
            if (oct == 0) {
                if (p.add1 && first_on) oadd = p.add1[(long)tau * p.add1_step + (long)m * p.ld_add1 + on];
                if (p.mask1 && first_on) omask = p.mask1[(long)tau * p.mask1_step + (long)m * p.ld_mask1 + on];
            } else if (p.mask2 && second_on) omask = p.mask2[(long)tau * p.mask2_step + (long)m * p.ld_mask2 + on];
        }
        if (mul) {
#pragma unroll
            for (int q = 0; q < KPW; ++q) {
                acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][q], wfa[q], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[0][q], wfb[q], acc[1], 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < KPW; ++q) {
                union { bf16x8_t b; uint4 u; } wc; wc.u = wlds[wave][q][lane];
                acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[1][q], wc.b, acc[1], 0, 0, 0);
            }
        }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int e = 0; e < 4; ++e) red[wave][ct][e * 64 + lane] = acc[ct][e];
        if constexpr (TS) { if (ts) __builtin_amdgcn_s_waitcnt(0xC07F); }   // (lgkmcnt(0): the partial tile is in LDS = the MFMAs are done)
        RNN_TS(u, 2)
        __syncthreads();
        RNN_TS(u, 3)
        // ---- fixed-order sum over the 8 K slices + epilogue: 512 outputs, one per thread
        {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) v += red[w][oct][oe];
            const bool on_ = oct == 0 ? first_on : second_on;
            asm volatile("" : "+v"(oadd), "+v"(omask));
            v += oadd + obias[g];
            if ((oct == 0 ? p.mask1 : p.mask2) != nullptr) v = omask > 0.f ? v : 0.f;
            else if (p.relu) v = fmaxf(v, 0.f);
            if (!on_) v = 0.f;
            otile[orow][oct][on - n0] = f32_to_bf16_bits(v);
            ftile[orow][oct][on - n0] = v;
        }
        RNN_TS(u, 4)
        __syncthreads();
        RNN_TS(u, 5)
        const int region = p.zb_row0 + (tau + 1) * p.zb_dir;
        if (wave == 0) {
            // exchange copy: lane = (row 0..15, half, 8-column chunk); skipped once nobody reads it any more (last wave step)
            const int row = lane & 15, ct = (lane >> 4) & 1, ch = lane >> 5;
            const int m = rowhalf * 32 + g * 16 + row;
            if (tau < p.S) {
                if (m < p.B && (ct == 1 || first_on)) {
                    const uint4 v16 = *(const uint4*)&otile[row][ct][ch * 8];
                    unsigned long long* dst = (unsigned long long*)(p.xb + (long)region * 64 * ldz + ((long)(ct * (H / 8) + n0 / 8 + ch) * 64 + m) * 8);
                    __hip_atomic_store(dst, (unsigned long long)v16.x | ((unsigned long long)v16.y << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + 1, (unsigned long long)v16.z | ((unsigned long long)v16.w << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                pending = g;
            }
            asm volatile("" ::: "memory");                         // nothing this wave loads later may be hoisted above the stores
        } else if (wave <= 2) {
            // fp32 rows: 16 rows x 2 halves x 4 chunks of 4 columns = 128 float4 pieces
            const int t = tid - 64, row = t & 15, ct = (t >> 4) & 1, ch = t >> 5;
            const int m = rowhalf * 32 + g * 16 + row;
            if (m < p.B && (ct == 1 || first_on)) {
                float* zn = p.z + (long)(tau + 1) * p.z_step;
                *(float4*)(zn + (long)m * ldz + ct * H + n0 + ch * 4) = *(const float4*)&ftile[row][ct][ch * 4];
            }
        } else if (wave == 3) {
            const int row = lane & 15, ct = (lane >> 4) & 1, ch = lane >> 5;
            const int m = rowhalf * 32 + g * 16 + row;
            if (m < p.B && (ct == 1 || first_on))
                *(uint4*)(p.zb + (long)region * p.B * ldz + (long)m * ldz + ct * H + n0 + ch * 8) = *(const uint4*)&otile[row][ct][ch * 8];
        } else if (wave == 4 && p.zt) {
            // transposed mirror: lane = (half, column, 8-row chunk)
            const int ct = lane >> 5, col = (lane >> 1) & 15, ch = lane & 1;
            const int m = rowhalf * 32 + g * 16 + ch * 8;
            if (m < p.B && (ct == 1 || first_on)) {               // B % 8 == 0 (checked by the launcher)
                unsigned w[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) w[e] = (unsigned)otile[ch * 8 + 2 * e][ct][col] | ((unsigned)otile[ch * 8 + 2 * e + 1][ct][col] << 16);
                *(uint4*)(p.zt + (long)(ct * H + n0 + col) * p.ld_t + (long)region * p.B + m) = make_uint4(w[0], w[1], w[2], w[3]);
            }
        }
        // (the LDS tiles are rewritten two barriers later: no extra barrier needed here)
        RNN_TS(u, 6)
    }
#undef RNN_TS
    __syncthreads();
    if ((p.dbg & 4) && blockIdx.x == 0 && tid == 0) {
        __hip_atomic_store(p.err, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (p.err_sticky) __hip_atomic_fetch_or(p.err_sticky, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (__hip_atomic_load(p.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
        float* zn = p.z + (long)(p.S + 1) * p.z_step;
        for (int o = tid; o < 32 * 16; o += 512) {
            const int m = rowhalf * 32 + o / 16, n = n0 + o % 16;
            if (m < p.B) { zn[(long)m * ldz + n] = __builtin_nanf(""); zn[(long)m * ldz + H + n] = __builtin_nanf(""); }
        }
    }
}

// zero the barrier header and the bf16 mirror of the initial state row: a kernel, not hipMemsetAsync, so that a captured hipGraph holds
// nothing but kernel nodes with plain pointer arguments (a captured hipMemsetAsync node
// was found to leave the barrier header un-zeroed on later replays once other allocations ran in between: NaN from replay 2 on, round 2;
// HULC_RNN_MEMSET=1 restores the memset calls to reproduce it)
__global__ __launch_bounds__(256) void rnn_prep_kernel(uint4* __restrict__ a, long na, uint4* __restrict__ b, long nb, float4* __restrict__ z0, long nz0,
                                                       float* __restrict__ zl, int B, int H) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    if (i < na) a[i] = z;
    if (i < nb) b[i] = z;
    // the fp32 state rows the sweep reads but never writes: the initial row (B x 2H) and the FIRST half of the last row (its second half is
    // the sweep's final output) — were four torch fill launches per step
    if (z0 && i < nz0) z0[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (zl && i < (long)B * H / 4) { const long e = i * 4; *(float4*)(zl + (e / H) * 2 * H + e % H) = make_float4(0.f, 0.f, 0.f, 0.f); }
}

// rows x width bf16 elements at row pitch `pitch` <- 0 (initial-state column block of the transposed mirror)
__global__ __launch_bounds__(256) void rnn_zero2d_kernel(uint16_t* __restrict__ dst, long pitch, int width, int rows) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < (long)rows * width) dst[(i / width) * pitch + i % width] = 0;
}

}  // namespace

extern "C" long hulc_rnn_wavefront_mirror_offset(void) { return RNN_WS_HEADER; }
// header | row-major mirror | exchange copy | transposed mirror (2H, (S+2)*B) when B % 8 == 0
extern "C" long hulc_rnn_wavefront_mirror_t_offset(int S, int B, int H) { return B % 8 ? 0 : RNN_WS_HEADER + (long)(S + 2) * (B + 64) * 2 * H * 2; }
extern "C" long hulc_rnn_wavefront_workspace(int S, int B, int H) { return RNN_WS_HEADER + (long)(S + 2) * (B + 64 + (B % 8 ? 0 : B)) * 2 * H * 2; }

// see include/hulc2_amd.h
extern "C" int hulc_rnn_wavefront(const hulc_rnn_wave_desc* d, void* ws, void* stream) {
    if (!d || !ws || !d->z || !d->wA || !d->wB1 || !d->wB2) return hulc_fail(-1, "hulc_rnn_wavefront: null pointer");
    if (d->H != 2048) return hulc_fail(-2, "hulc_rnn_wavefront: built for hidden size 2048 (use the per-step hulc_gemm path otherwise)");
    if (d->B < 1 || d->B > 64 || d->S < 1) return hulc_fail(-2, "hulc_rnn_wavefront: needs 1 <= B <= 64 rows and S >= 1");
    if ((!d->tA && (d->ldA % 8 || (uintptr_t)d->wA % 16)) || (!d->tB1 && (d->ldB1 % 8 || (uintptr_t)d->wB1 % 16)) ||
        (!d->tB2 && (d->ldB2 % 8 || (uintptr_t)d->wB2 % 16)))
        return hulc_fail(-4, "hulc_rnn_wavefront: k-major weights must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    WaveP p;
    p.z = d->z; p.z_step = d->z_step;
    p.bar = (unsigned*)ws; p.err = (int*)((unsigned*)ws + RNN_ERR_WORD); p.zb = (uint16_t*)((char*)ws + RNN_WS_HEADER);
    p.xb = p.zb + (long)(d->S + 2) * d->B * 2 * d->H;
    p.zt = (d->B % 8 || !d->mirror_t) ? nullptr : (uint16_t*)((char*)ws + hulc_rnn_wavefront_mirror_t_offset(d->S, d->B, d->H));
    p.ld_t = (long)(d->S + 2) * d->B;
    p.wA = (const uint16_t*)d->wA; p.wB1 = (const uint16_t*)d->wB1; p.wB2 = (const uint16_t*)d->wB2;
    p.ldA = d->ldA; p.ldB1 = d->ldB1; p.ldB2 = d->ldB2; p.tA = d->tA; p.tB1 = d->tB1; p.tB2 = d->tB2;
    p.add1 = d->add1; p.add1_step = d->add1_step; p.ld_add1 = d->ld_add1;
    p.bias1a = d->bias1a; p.bias1b = d->bias1b; p.bias2a = d->bias2a; p.bias2b = d->bias2b;
    p.mask1 = d->mask1; p.mask1_step = d->mask1_step; p.ld_mask1 = d->ld_mask1;
    p.mask2 = d->mask2; p.mask2_step = d->mask2_step; p.ld_mask2 = d->ld_mask2;
    p.relu = d->relu; p.S = d->S; p.B = d->B; p.H = d->H; p.err_sticky = d->err_sticky;
    p.add1c = d->add1c; p.ld_add1c = d->ld_add1c;
    p.dbg = getenv("HULC_RNN_DBG") ? atoi(getenv("HULC_RNN_DBG")) : 0;
    p.ts = nullptr;
    if (p.dbg & 8) {                                         // (probe, eager launches only: the next call prints the previous launch's phase stamps)
        static unsigned long long* buf = nullptr; static int calls = 0;
        const int NW = 2 * 3 * 80 * 8;
        if (!buf) { hipMalloc(&buf, NW * 8); hipMemset(buf, 0, NW * 8); }
        if (calls >= 1 && calls <= 4) {
            std::vector<unsigned long long> h(NW);
            hipMemcpy(h.data(), buf, NW * 8, hipMemcpyDeviceToHost);
            const char* wn[3] = {"wave 1", "wave 0 (exchange)", "wave 7 (poll)"};
            for (int wg = 0; wg < 2; ++wg) for (int wv = 0; wv < 3; ++wv) {
                double ph[7] = {0, 0, 0, 0, 0, 0, 0}; int cnt = 0;
                for (int u = 8; u < 60; ++u) {
                    const unsigned long long* t = &h[((wg * 3 + wv) * 80 + u) * 8];
                    const unsigned long long* tn = t + 8;
                    if (!t[0] || !tn[0]) continue;
                    for (int k = 0; k < 6; ++k) ph[k] += (double)(t[k + 1] - t[k]);
                    ph[6] += (double)(tn[0] - t[6]); ++cnt;
                }
                if (cnt) fprintf(stderr, "[rnn stamps call %d wg %d %s, 10 ns ticks, mean of %d sub-steps] poll+sync %.1f | loads+mfma %.1f | sync2 %.1f | sum+epilogue %.1f | sync3 %.1f | stores %.1f | loop %.1f\n",
                                 calls - 1, wg ? 100 : 0, wn[wv], cnt, ph[0] / cnt, ph[1] / cnt, ph[2] / cnt, ph[3] / cnt, ph[4] / cnt, ph[5] / cnt, ph[6] / cnt);
            }
        }
        if (calls < 4) p.ts = buf;
        ++calls;
    }
    p.zb_row0 = d->z_step > 0 ? 0 : d->S + 1; p.zb_dir = d->z_step > 0 ? 1 : -1;
    // barrier words, and the bf16 copy of the (zero) initial state row: the copy is a full mirror of the fp32 rows for the weight-gradient GEMMs
    static const bool use_memset = getenv("HULC_RNN_MEMSET") != nullptr;
    if (use_memset) {
        if (hipMemsetAsync(ws, 0, RNN_WS_HEADER, s) != hipSuccess ||
            hipMemsetAsync(p.zb + (long)p.zb_row0 * d->B * 2 * d->H, 0, (size_t)d->B * 2 * d->H * 2, s) != hipSuccess)   // (exchange region 0 is never read: z_0 = 0 is skipped)
            return hulc_fail(-9, "hulc_rnn_wavefront: could not reset the barrier words");
    } else {
        if ((uintptr_t)ws % 16 || RNN_WS_HEADER % 16) return hulc_fail(-4, "hulc_rnn_wavefront: workspace must be 16-byte aligned");
        const long na = RNN_WS_HEADER / 16, nb = (long)d->B * 2 * d->H * 2 / 16;
        const long nz0 = d->zero_edges ? (long)d->B * 2 * d->H / 4 : 0;
        long n = na > nb ? na : nb;
        if (nz0 > n) n = nz0;
        float* zlast = d->zero_edges ? d->z + (long)(d->S + 1) * d->z_step : nullptr;       // row S+1 of the sweep
        rnn_prep_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>((uint4*)ws, na, (uint4*)(p.zb + (long)p.zb_row0 * d->B * 2 * d->H), nb,
                                                                  d->zero_edges ? (float4*)d->z : nullptr, nz0, zlast, d->B, d->H);
    }
    if (p.zt) {
        const long n = 2L * d->H * d->B;
        rnn_zero2d_kernel<<<(unsigned)((n + 255) / 256), 256, 0, s>>>(p.zt + (long)p.zb_row0 * d->B, p.ld_t, d->B, 2 * d->H);
    }
    if (d->tA != d->tB1 || d->tA != d->tB2) return hulc_fail(-3, "hulc_rnn_wavefront: the three weight matrices share one layout (tA == tB1 == tB2)");
    static const bool pipelined = !(getenv("HULC_RNN_PIPE") && atoi(getenv("HULC_RNN_PIPE")) == 0);     // HULC_RNN_PIPE=0: the unpipelined kernel
    if (pipelined && p.ts) {
        if (d->tA) rnn_wavefront2_kernel<2048, true, true><<<2 * (2048 / 16), 512, 0, s>>>(p);
        else rnn_wavefront2_kernel<2048, false, true><<<2 * (2048 / 16), 512, 0, s>>>(p);
    } else if (pipelined) {
        if (d->tA) rnn_wavefront2_kernel<2048, true><<<2 * (2048 / 16), 512, 0, s>>>(p);
        else rnn_wavefront2_kernel<2048, false><<<2 * (2048 / 16), 512, 0, s>>>(p);
    } else if (d->tA) rnn_wavefront_kernel<2048, true><<<2 * (2048 / 16), 512, 0, s>>>(p);
    else rnn_wavefront_kernel<2048, false><<<2 * (2048 / 16), 512, 0, s>>>(p);
    return hulc_check_launch("hulc_rnn_wavefront");
}
