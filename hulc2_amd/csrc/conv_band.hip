// conv_band.hip — LDS-band convolution for the NHWC bf16 layers (conv2 / conv3 forward and every data gradient).
//
// reference arithmetic: nn.Conv2d(+ReLU) of hulc2/models/perceptual_encoders/vision_network.py:41-46 and
// vision_network_gripper.py:15-19, and autograd's conv2d input gradient.
//
// The gather kernel in conv.hip re-reads every input element KH*KW/s^2 times through L1 in 16-byte pieces; here
// a workgroup stages (once, coalesced) the band of input rows its output rows need AND the layer's whole weight
// matrix into LDS, after which the k-loop has no global loads and no barriers: every MFMA operand is one
// ds_read_b128 — pixel fragment straight out of the band (pixel stride padded by 16 B: conflict-free), weight
// fragment out of the resident [Cout][K] image.  Zero padding (data gradients) is written into the band.
//   work unit  = (frame, band of R output rows); workgroups walk units persistently, weights are loaded once
//   wave       = two 32-pixel tiles x all output channels (TM = 2, TN = Cout/32)
//   correlation: in(y,x) = band[(oy*S + ty)][(ox*S + tx)], ty < TH, tx < TW (pad folded into the band origin)
#include "hulc_common.h"
#include "hulc_abi_internal.h"
#include <stdlib.h>

namespace {

struct BandP {
    const void* X; void* Y; const void* Wt; const float* bias; const void* mask;
    int x_dtype, y_dtype, w_dtype, mask_dtype;
    int Nimg, H, W;                 // input tensor dims (NHWC, C = template)
    int OH, OW;                     // output grid of this launch
    int pad_y, pad_x;               // band origin: input row = oy*S + ty - pad_y
    int R;                          // output rows per work unit
    long x_sn, x_sy, x_sx;          // input element strides
    long y_sn, y_sy, y_sx;          // output element strides (channels contiguous)
    long ldw;                       // global weight row stride (elements)
    long w_tap_off[16];             // global offset (elements, inside a weight row) of tap (ty, tx)
    int relu; float mask_scale;
    long long* dbg;                 // optional per-workgroup phase cycle counters (tools/conv_one.py), normally null
};

template <int C, int COUT, int TH, int TW, int S>
__global__ __launch_bounds__(512) void conv_band_kernel(BandP p) {
    constexpr int NT = 512, NWAVE = 8;
    constexpr int K = TH * TW * C;
    constexpr int PS = C * 2 + 16;          // band pixel stride (bytes): +16 B keeps ds_read_b128 conflict-free
    constexpr int WS = K * 2 + 16;          // weight row stride (bytes)
    constexpr int TN = COUT / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* wlds = smem;                      // [COUT][WS]
    long* ooff = (long*)(smem + COUT * WS); // [8 waves][2 tiles][32] output element offsets (no divisions in the epilogue)
    char* band = smem + COUT * WS + 8 * 2 * 32 * 8;   // [rows][Wb][PS]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int Wb = (p.OW - 1) * S + TW;                     // band columns (padding included)
    const int bands = (p.OH + p.R - 1) / p.R;
    const int nunits = p.Nimg * bands;

    long long t_w = 0, t_s = 0, t_c = 0, t0 = clock64();
    // ---- weights -> LDS (once per workgroup), dense [cout][(ty,tx,c)] from the tap table
    constexpr int WCH = COUT * (K / 8);
#pragma unroll 4
    for (int id = tid; id < (WCH + NT - 1) / NT * NT; id += NT) {
        const int idc = id < WCH ? id : 0;
        const int co = idc / (K / 8), kc = idc % (K / 8);
        const int t = (kc * 8) / C, c0 = (kc * 8) % C;
        Chunk8 ch;
        chunk_load_contig(ch, p.Wt, p.w_dtype, (long)co * p.ldw + p.w_tap_off[t] + c0);
        if (id < WCH) chunk_store_lds<bf16_t>(wlds + co * WS + kc * 16, ch);
    }

    for (int unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
        const int n = unit / bands, b = unit % bands;
        const int r0 = b * p.R;
        const int R = (r0 + p.R <= p.OH) ? p.R : p.OH - r0;
        const int rows = (R - 1) * S + TH;
        const int iy0 = r0 * S - p.pad_y, ix0 = -p.pad_x;
        __syncthreads();                                    // previous unit's reads are done (and weights are visible)
        if (unit == (int)blockIdx.x) { t_w = clock64() - t0; }
        long long t1 = clock64();
        // ---- stage the input band (zero outside the tensor).  Loads are issued UNR deep before the first LDS write so
        //      the band arrives at memory-level parallelism instead of one L2/HBM round trip per chunk.
        const int nchunk = rows * Wb * (C / 8);
        constexpr int UNR = 8;
        for (int base = 0; base < nchunk; base += NT * UNR) {
            uint4 v[UNR]; int dst[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {
                const int id = base + u * NT + tid;
                const int idc = id < nchunk ? id : 0;
                const int cc = idc % (C / 8); const int px = idc / (C / 8);
                const int bc = px % Wb, br = px / Wb;
                const int iy = iy0 + br, ix = ix0 + bc;
                dst[u] = id < nchunk ? (br * Wb + bc) * PS + cc * 16 : -1;
                const bool inb = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                // unconditional load from a clamped address, then select: no branch (and no vmcnt(0)) around the load
                const long off = inb ? (long)n * p.x_sn + (long)iy * p.x_sy + (long)ix * p.x_sx + cc * 8 : (long)n * p.x_sn;
                uint4 t;
                if (p.x_dtype == HULC_BF16) t = *(const uint4*)((const uint16_t*)p.X + off);
                else {
                    const float4* q = (const float4*)((const float*)p.X + off);
                    const float4 a = q[0], c = q[1];
                    t.x = pack_bf16x2(a.x, a.y); t.y = pack_bf16x2(a.z, a.w); t.z = pack_bf16x2(c.x, c.y); t.w = pack_bf16x2(c.z, c.w);
                }
                v[u].x = inb ? t.x : 0u; v[u].y = inb ? t.y : 0u; v[u].z = inb ? t.z : 0u; v[u].w = inb ? t.w : 0u;
            }
#pragma unroll
            for (int u = 0; u < UNR; ++u)
                if (dst[u] >= 0) *(uint4*)(band + dst[u]) = v[u];
        }
        __syncthreads();
        t_s += clock64() - t1; t1 = clock64();

        // ---- compute: pairs of 32-pixel tiles per wave
        const int npix = R * p.OW;
        const int ntile = (npix + 31) / 32;
        for (int tp = wave; tp * 2 < ntile; tp += NWAVE) {
            int pix[2], abase[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int q = (tp * 2 + i) * 32 + r;
                pix[i] = q;
                if (q >= npix) q = npix - 1;
                const int oy = q / p.OW, ox = q % p.OW;
                abase[i] = ((oy * S) * Wb + ox * S) * PS + h * 16;
                if (h == 0) ooff[(wave * 2 + i) * 32 + r] = pix[i] < npix ? (long)n * p.y_sn + (long)(r0 + oy) * p.y_sy + (long)ox * p.y_sx : -1;
            }
            f32x16_t acc[2][TN];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
#pragma unroll
            for (int ty = 0; ty < TH; ++ty)
#pragma unroll
                for (int tx = 0; tx < TW; ++tx)
#pragma unroll
                    for (int c0 = 0; c0 < C; c0 += 16) {
                        const int aoff = (ty * Wb + tx) * PS + c0 * 2;
                        const int koff = ((ty * TW + tx) * C + c0) * 2 + h * 16;
                        bf16x8_t a[2], bw[TN];
#pragma unroll
                        for (int i = 0; i < 2; ++i) a[i] = *(const bf16x8_t*)(band + abase[i] + aoff);
#pragma unroll
                        for (int j = 0; j < TN; ++j) bw[j] = *(const bf16x8_t*)(wlds + (j * 32 + r) * WS + koff);
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], bw[j], acc[i][j], 0, 0, 0);
                    }
            // ---- epilogue: lane = output channel, accumulator register = pixel
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int co = j * 32 + r;
                const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const long po = ooff[(wave * 2 + i) * 32 + acc_row(e, lane)];   // written by this wave above: wave-local LDS, in order
                        if (po < 0) continue;
                        const long off = po + co;
                        float v = acc[i][j][e] + bv;
                        if (p.relu) v = fmaxf(v, 0.f);
                        if (p.mask) v = load_elem(p.mask, p.mask_dtype, off) > 0.f ? v * p.mask_scale : 0.f;
                        store_elem(p.Y, p.y_dtype, off, v);
                    }
            }
            (void)pix;
        }
        t_c += clock64() - t1;
    }
    if (p.dbg && tid == 0) { p.dbg[blockIdx.x * 4] = t_w; p.dbg[blockIdx.x * 4 + 1] = t_s; p.dbg[blockIdx.x * 4 + 2] = t_c; p.dbg[blockIdx.x * 4 + 3] = clock64() - t0; }
}

template <int C, int COUT, int TH, int TW, int S>
int launch_band(BandP& p, hipStream_t s) {
    constexpr int K = TH * TW * C, PS = C * 2 + 16, WS = K * 2 + 16;
    const int Wb = (p.OW - 1) * S + TW;
    const long wbytes = (long)COUT * WS;
    const long budget = 160 * 1024 - wbytes - 8 * 2 * 32 * 8 - 256;
    // rows per unit: as many output rows as the LDS band allows (whole frame when it fits)
    int R = p.OH;
    while (R > 1 && (long)((R - 1) * S + TH) * Wb * PS > budget) --R;
    if ((long)((R - 1) * S + TH) * Wb * PS > budget) return -1;
    // balance: equal-ish bands
    const int bands = (p.OH + R - 1) / R;
    R = (p.OH + bands - 1) / bands;
    p.R = R;
    const size_t lds = (size_t)wbytes + 8 * 2 * 32 * 8 + (size_t)((R - 1) * S + TH) * Wb * PS;
    if ((long)R * p.OW < 128) return -1;        // tiny frames: one unit cannot feed 8 waves, the gather kernel is faster
    const int nunits = p.Nimg * bands;
    const int grid = nunits < 256 ? nunits : 256;
    auto kern = conv_band_kernel<C, COUT, TH, TW, S>;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -2;
        attr_set = true;
    }
    kern<<<grid, 512, lds, s>>>(p);
    return 0;
}

}  // namespace

// returns 0 when the band kernel took the launch, 1 when the geometry is not covered (caller falls back to the gather
// kernel), negative on error.  Only bf16 compute; NHWC input with C in {32, 64}.
int hulc_conv_band_dispatch(int C, int COUT, int TH, int TW, int S, const void* x, int x_dtype, int N, int H, int W, int OH, int OW,
                            int pad_y, int pad_x, long x_sn, long x_sy, long x_sx, void* y, int y_dtype, long y_sn, long y_sy, long y_sx,
                            const void* wt, int w_dtype, long ldw, const long* w_tap_off, const float* bias, const void* mask,
                            int mask_dtype, int relu, hipStream_t s) {
    if (getenv("HULC_NO_BAND")) return 1;
    BandP p;
    p.X = x; p.Y = y; p.Wt = wt; p.bias = bias; p.mask = mask;
    p.x_dtype = x_dtype; p.y_dtype = y_dtype; p.w_dtype = w_dtype; p.mask_dtype = mask_dtype;
    p.Nimg = N; p.H = H; p.W = W; p.OH = OH; p.OW = OW; p.pad_y = pad_y; p.pad_x = pad_x; p.R = OH;
    p.x_sn = x_sn; p.x_sy = x_sy; p.x_sx = x_sx; p.y_sn = y_sn; p.y_sy = y_sy; p.y_sx = y_sx;
    p.ldw = ldw; p.relu = relu; p.mask_scale = 1.f;
    { const char* e = getenv("HULC_BAND_DBG"); p.dbg = e ? (long long*)strtoull(e, nullptr, 0) : nullptr; }
    for (int t = 0; t < TH * TW && t < 16; ++t) p.w_tap_off[t] = w_tap_off[t];
    int rc = 1;
    // measured (tools/conv_bench.py, 1024 frames): conv3 forward 0.090 ms vs 0.143 ms gather; conv2 forward on par
    if (C == 32 && COUT == 64 && TH == 4 && TW == 4 && S == 2) rc = launch_band<32, 64, 4, 4, 2>(p, s);
    else if (C == 64 && COUT == 64 && TH == 3 && TW == 3 && S == 1) rc = launch_band<64, 64, 3, 3, 1>(p, s);
    else if (C == 64 && COUT == 32 && TH == 2 && TW == 2 && S == 1) rc = launch_band<64, 32, 2, 2, 1>(p, s);
    else return 1;
    if (rc == -1) return 1;                      // band does not fit LDS: gather kernel
    if (rc < 0) return hulc_fail(-8, "conv band: could not raise the dynamic LDS limit");
    return 0;
}
