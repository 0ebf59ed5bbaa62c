// xcd_exchange_probe — what does one round of "publish my slice -> arrive at a counter -> wait for the group -> read everybody's slices" cost
// when the group lives on ONE XCD (its own L2) against a group spread over four XCDs (the recurrent decoder kernel's row halves today)?
//
// Context (DESIGN: recurrent kernel): rnn_wavefront2's sub-step chain is stores -> write-through acknowledgement -> agent-scope
// counter -> propagation -> 128 KB of state loads.  If the 32 workgroups that exchange a state lived on one XCD, the stores could stop at that
// XCD's L2 (plain stores), the counter could be an L2 atomic (workgroup-scope encoding: no sc bits -> executed in the local L2) and the poll a
// sc0 load (L1 bypass only).  This probe measures both protocols with the real sizes: 128 KB per round, every workgroup reads all of it.
//
//   variant 0  group = the 128 workgroups of an XCD parity (4 XCDs), 1 KB slice each, write-through (agent-scope) stores, one agent-scope
//              counter per XCD, pollers wait for the group's four counters            [the shipped protocol]
//   variant 1  group = the 32 workgroups of one XCD, 4 KB slice each, plain stores, one workgroup-scope (L2) counter, polled with a
//              workgroup-scope fetch_add(0) (atomics execute in the XCD's L2; a workgroup-scope LOAD may be served by the CU's L1 for ever:
//              measured — the first version of this probe polled with sc0 loads and timed out)
//   variant 4  as 1, polled with buffer_inv sc0 (L1 invalidate) + an ordinary load
//   variant 2  as 1 but with the agent-scope stores / counter / poll of variant 0     [placement alone]
//   variant 3  as 0 but plain stores + L2 atomics (expected to FAIL its checksum or hang-guard: not coherent across XCDs) — skipped unless argv[1]=="3"
// Every round writes a fresh region (addresses never cached before they are complete).  Output: us per round, checksum status.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

constexpr int REGION = 128 * 1024;        // bytes exchanged per group and round
constexpr int CTR_STRIDE = 1024;          // words between counters (4 KB)

__device__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 15; }

template <int V>
__global__ __launch_bounds__(512) void probe(unsigned char* buf, unsigned* ctr, int rounds, unsigned long long* sums, int* fail) {
    const int tid = threadIdx.x, wg = blockIdx.x, xcd = wg & 7, slot = wg >> 3;
    // group id, member index, members
    int grp, mem, nmem;
    if (V == 0 || V == 3) { grp = xcd & 1; mem = (xcd >> 1) * 32 + slot; nmem = 128; }
    else { grp = xcd; mem = slot; nmem = 32; }
    if (tid == 0 && (int)xcc_id() != xcd) *fail = 100 + V;     // the placement assumption itself (workgroup -> XCD round-robin)
    const int slice = REGION / nmem;                           // bytes this workgroup publishes per round
    unsigned long long acc = 0;
    for (int r = 0; r < rounds; ++r) {
        unsigned char* region = buf + ((size_t)r * 8 + grp) * REGION;
        // ---- publish: slice bytes, 16 per thread (threads beyond the slice idle)
        if (tid * 16 < slice) {
            const unsigned v = (unsigned)(r * 2654435761u) ^ (unsigned)(mem * 40503u + tid);
            unsigned long long* dst = (unsigned long long*)(region + (size_t)mem * slice + tid * 16);
            const unsigned long long lo = v | ((unsigned long long)(v + 1) << 32), hi = (v + 2) | ((unsigned long long)(v + 3) << 32);
            if (V == 0 || V == 2) {
                __hip_atomic_store(dst, lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(dst + 1, hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else { dst[0] = lo; dst[1] = hi; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        // ---- arrive
        if (tid == 0) {
            if (V == 0 || V == 2) __hip_atomic_fetch_add(ctr + (V == 0 ? xcd : 8 + xcd) * CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else __hip_atomic_fetch_add(ctr + (V == 3 ? 16 + (xcd & 1) : 24 + xcd) * CTR_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        // ---- wait for the group
        const int npoll = V == 0 ? 4 : 1;
        if (tid < npoll) {
            const unsigned* c;
            unsigned want;
            if (V == 0) { c = ctr + ((xcd & 1) + 2 * tid) * CTR_STRIDE; want = 32u * (unsigned)(r + 1); }
            else if (V == 2) { c = ctr + (8 + xcd) * CTR_STRIDE; want = 32u * (unsigned)(r + 1); }
            else if (V == 3) { c = ctr + (16 + (xcd & 1)) * CTR_STRIDE; want = 128u * (unsigned)(r + 1); }
            else { c = ctr + (24 + xcd) * CTR_STRIDE; want = 32u * (unsigned)(r + 1); }
            long spins = 0;
            while (true) {
                unsigned got;
                if (V == 0 || V == 2) got = __hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                else if (V == 4) { asm volatile("buffer_inv sc0" ::: "memory"); got = *(volatile const unsigned*)c; }
                else got = __hip_atomic_fetch_add((unsigned*)c, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (got >= want) break;
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 16)) { *fail = 1 + V; break; }
            }
        }
        __syncthreads();
        asm volatile("" ::: "memory");
        // ---- read the whole region (ordinary loads: 512 threads x 16 bytes x 16 trips)
        const uint4* src = (const uint4*)region;
        uint4 q[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) q[j] = src[j * 512 + tid];
#pragma unroll
        for (int j = 0; j < 16; ++j) acc += (unsigned long long)q[j].x + q[j].y + q[j].z + q[j].w;
    }
    // per-thread sums are reduced on the host side of the check (thread 0 of each workgroup stores a block sum)
    __shared__ unsigned long long red[512];
    red[tid] = acc;
    __syncthreads();
    if (tid == 0) { unsigned long long s = 0; for (int i = 0; i < 512; ++i) s += red[i]; sums[wg] = s; }
}

static unsigned long long expect(int V, int rounds) {
    // every workgroup reads its group's whole region every round: sum over members, threads, 4 words
    const int nmem = (V == 0 || V == 3) ? 128 : 32, slice = REGION / nmem, nthr = slice / 16;
    unsigned long long s = 0;
    for (int r = 0; r < rounds; ++r)
        for (int m = 0; m < nmem; ++m)
            for (int t = 0; t < nthr; ++t) {
                const unsigned v = (unsigned)(r * 2654435761u) ^ (unsigned)(m * 40503u + t);
                s += (unsigned long long)v + (unsigned)(v + 1) + (unsigned)(v + 2) + (unsigned)(v + 3);
            }
    return s;
}

template <int V>
static void run(const char* name, int rounds) {
    unsigned char* buf; unsigned* ctr; unsigned long long* sums; int* fail;
    CK(hipMalloc(&buf, (size_t)rounds * 8 * REGION));
    CK(hipMalloc(&ctr, 32 * CTR_STRIDE * 4));
    CK(hipMalloc(&sums, 256 * 8));
    CK(hipMalloc(&fail, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f; bool ok = true; int failed = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CK(hipMemset(buf, 0xff, (size_t)rounds * 8 * REGION));
        CK(hipMemset(ctr, 0, 32 * CTR_STRIDE * 4));
        CK(hipMemset(fail, 0, 4));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        probe<V><<<256, 512>>>(buf, ctr, rounds, sums, fail);
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
        std::vector<unsigned long long> h(256);
        CK(hipMemcpy(h.data(), sums, 256 * 8, hipMemcpyDeviceToHost));
        CK(hipMemcpy(&failed, fail, 4, hipMemcpyDeviceToHost));
        const unsigned long long want = expect(V, rounds);
        for (int i = 0; i < 256; ++i) ok = ok && h[i] == want;
        if (failed) break;
    }
    printf("variant %d  %-62s %7.2f us per round   checksum %s%s\n", V, name, best * 1e3f / rounds, ok ? "ok" : "WRONG", failed >= 100 ? "  (workgroup -> XCD is NOT id % 8 here)" : (failed ? "  (poll timed out)" : ""));
    CK(hipFree(buf)); CK(hipFree(ctr)); CK(hipFree(sums)); CK(hipFree(fail));
}

int main(int argc, char** argv) {
    const int rounds = 64;
    run<0>("4 XCDs x 32 wgs, write-through stores, agent counters (shipped)", rounds);
    run<1>("1 XCD x 32 wgs, plain stores, L2 counter, fetch_add(0) poll", rounds);
    run<4>("1 XCD x 32 wgs, plain stores, L2 counter, buffer_inv sc0 + load poll", rounds);
    run<2>("1 XCD x 32 wgs, write-through stores, agent counter", rounds);
    if (argc > 1 && !strcmp(argv[1], "3")) run<3>("4 XCDs, plain stores + L2 atomics (not coherent: expected wrong)", rounds);
    return 0;
}
