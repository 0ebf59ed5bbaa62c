// u8_frames.h — staging of uint8 NHWC camera frames for the conv1 band kernels.
//
// reference arithmetic folded into the load: RandomShiftsAug (replicate-pad by `pad`, integer crop at (sx, sy)),
// ScaleImageTensor (x / 255) and Normalize(0.5, 0.5) — hulc2/utils/transforms.py:8-19, 104-138.
//
// A thread turns 8 consecutive band elements (flat index over band rows x W) of ALL THREE channels into three uint4 of
// bf16 plane data.  Each 4-pixel half reads its 12 source bytes as (at most) four aligned dwords of one source row — the
// row clamp is one v_med3 per half, the column clamp only moves the 4-pixel window, realigned with v_alignbyte — instead of
// 24 single-byte loads.  Halves whose window was clamped (the replicated border columns) pick their pixels with a byte permute.
//
// (b / 255 - 0.5) / 0.5 is evaluated as fma(b, 2/255, -1): for all 256 byte values the two round to the same bf16 (checked
// exhaustively, tests/test_oracle_golden.py::test_u8_normalise_fma_is_bf16_exact), and bf16 is what the bands hold.
#pragma once
#include "hulc_common.h"

// the 12 bytes of a 4-pixel window (R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3) -> bf16 plane data of the three channels.  Bytes are moved with
// v_perm_b32 while they are still bytes: two perms gather a channel's four pixels into one dword, a third applies the border replication
// (element i <- window pixel clamp(i + d, 0, 3); the selector is a byte window of 00 00 00 00 | 00 01 02 03 | 03 03 03 03 at 4 + d, d == 0
// being the identity, so there is no divergent path).  Conversion is v_cvt_f32_ubyteN + a packed fma + v_cvt_pk_bf16: ~55 VALU per half
// where per-element float selects cost ~120 (the staging of uint8 frames is VALU-bound: tools/conv1_u8_probe.py).
HULC_DEVICE void u8_window_planes(uint32_t e0, uint32_t e1, uint32_t e2, int d, bool live, uint32_t (&o)[3][2]) {
    const int k = (d < -3 ? -3 : (d > 3 ? 3 : d)) + 4;             // |d| >= 3: every element is the border pixel
    const uint32_t lo = k < 4 ? 0u : 0x03020100u, hi = k < 4 ? 0x03020100u : 0x03030303u;
    const uint32_t rep = __builtin_amdgcn_alignbyte(hi, lo, (uint32_t)k & 3u);
    const uint32_t ch[3] = {
        __builtin_amdgcn_perm(0u, __builtin_amdgcn_perm(e2, __builtin_amdgcn_perm(e1, e0, 0x00060300u), 0x05020100u), rep),
        __builtin_amdgcn_perm(0u, __builtin_amdgcn_perm(e2, __builtin_amdgcn_perm(e1, e0, 0x00070401u), 0x06020100u), rep),
        __builtin_amdgcn_perm(0u, __builtin_amdgcn_perm(e2, __builtin_amdgcn_perm(e1, e0, 0x00000502u), 0x07040100u), rep)};
    const f32x2_t k2 = {2.0f / 255.0f, 2.0f / 255.0f}, m2 = {-1.f, -1.f};
    const uint32_t lm = live ? 0xffffffffu : 0u;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const uint32_t v = ch[c];
        f32x2_t a = {(float)(v & 0xff), (float)((v >> 8) & 0xff)}, b = {(float)((v >> 16) & 0xff), (float)(v >> 24)};
        a = __builtin_elementwise_fma(a, k2, m2); b = __builtin_elementwise_fma(b, k2, m2);
        o[c][0] = pack_bf16x2(a[0], a[1]) & lm; o[c][1] = pack_bf16x2(b[0], b[1]) & lm;
    }
}

HULC_DEVICE void u8_half4(const unsigned char* img, int H, int W, int row, int x, int dx, int dy, bool live, uint32_t (&o)[3][2]) {
    int yy = row + dy;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    const int xx0 = x + dx;
    const int pw = xx0 < 0 ? 0 : (xx0 > W - 4 ? W - 4 : xx0);        // first pixel of the 4-pixel source window
    const int d = xx0 - pw;                                           // != 0: some of the 4 elements are replicated border pixels
    const int b0 = (yy * W + pw) * 3;                                 // < 2^31 for any camera frame
    const int sh = b0 & 3;
    const uint32_t* q = (const uint32_t*)(img + (b0 - sh));
    uint32_t w0 = 0, w1 = 0, w2 = 0, w3 = 0;
    if (live) { w0 = q[0]; w1 = q[1]; w2 = q[2]; if (sh) w3 = q[3]; }  // the fourth dword is only needed (and only in bounds) when sh > 0
    const uint32_t e0 = __builtin_amdgcn_alignbyte(w1, w0, sh), e1 = __builtin_amdgcn_alignbyte(w2, w1, sh),
                   e2 = __builtin_amdgcn_alignbyte(w3, w2, sh);       // R0 G0 B0 R1 | G1 B1 R2 G2 | B2 R3 G3 B3
    u8_window_planes(e0, e1, e2, d, live, o);
}

// 8 band elements starting at flat index e0 (multiple of 8; W % 4 == 0 so a half never straddles a row); elements >= nflt are zero.
HULC_DEVICE void u8_band_chunk3(const unsigned char* img, int H, int W, int row0, int e0, int nflt, int dx, int dy, uint4& p0, uint4& p1, uint4& p2) {
    const int rr = fast_div(e0, fast_rcp(W)), x = e0 - rr * W;      // (an integer division by a run-time W is ~25 instructions, four times per item)
    const bool wrap = x + 4 >= W;
    uint32_t a[3][2], b[3][2];
    u8_half4(img, H, W, row0 + rr, x, dx, dy, e0 < nflt, a);
    u8_half4(img, H, W, row0 + rr + (wrap ? 1 : 0), wrap ? 0 : x + 4, dx, dy, e0 + 4 < nflt, b);
    p0 = make_uint4(a[0][0], a[0][1], b[0][0], b[0][1]);
    p1 = make_uint4(a[1][0], a[1][1], b[1][0], b[1][1]);
    p2 = make_uint4(a[2][0], a[2][1], b[2][0], b[2][1]);
}

// ---- the same staging split in two: the LOAD half only issues the dword loads (results untouched, so a prefetch really stays in
// flight behind other work), the CONVERT half redoes the cheap index arithmetic and turns the raw windows into plane data.
HULC_DEVICE void u8_half4_load(const unsigned char* img, int H, int W, int row, int x, int dx, int dy, bool live, uint32_t* w) {
    int yy = row + dy;
    yy = yy < 0 ? 0 : (yy > H - 1 ? H - 1 : yy);
    const int xx0 = x + dx;
    const int pw = xx0 < 0 ? 0 : (xx0 > W - 4 ? W - 4 : xx0);
    const int b0 = (yy * W + pw) * 3, sh = b0 & 3;
    const char* q = (const char*)(img + (live ? b0 - sh : 0));
    // four separate dword loads on purpose (the offsets are opaque to the compiler): merged into a dwordx3 the result is a register TUPLE, and
    // under register pressure the allocator split such tuples by copying their dwords right behind the load — a wait for the load where it was
    // issued, i.e. no prefetch at all
    int o1 = 4, o2 = 8, o3 = (live && sh) ? 12 : 0;              // the fourth dword is only in bounds (and only needed) when sh > 0
    asm volatile("" : "+s"(o1), "+s"(o2));
    w[0] = *(const uint32_t*)q; w[1] = *(const uint32_t*)(q + o1); w[2] = *(const uint32_t*)(q + o2);
    w[3] = *(const uint32_t*)(q + o3);
}

HULC_DEVICE void u8_half4_convert(int W, int x, int dx, bool live, const uint32_t* w, uint32_t (&o)[3][2]) {
    const int xx0 = x + dx;
    const int pw = xx0 < 0 ? 0 : (xx0 > W - 4 ? W - 4 : xx0);
    const int d = xx0 - pw;
    const int sh = (pw * 3) & 3;                                 // (yy * W * 3) is a multiple of 4 (W % 4 == 0)
    const uint32_t e0 = __builtin_amdgcn_alignbyte(w[1], w[0], sh), e1 = __builtin_amdgcn_alignbyte(w[2], w[1], sh),
                   e2 = __builtin_amdgcn_alignbyte(w[3], w[2], sh);
    u8_window_planes(e0, e1, e2, d, live, o);
}

HULC_DEVICE void u8_band_chunk3_load(const unsigned char* img, int H, int W, int row0, int e0, int nflt, int dx, int dy, uint32_t* raw) {
    const int rr = fast_div(e0, fast_rcp(W)), x = e0 - rr * W;      // (an integer division by a run-time W is ~25 instructions, four times per item)
    const bool wrap = x + 4 >= W;
    u8_half4_load(img, H, W, row0 + rr, x, dx, dy, e0 < nflt, raw);
    u8_half4_load(img, H, W, row0 + rr + (wrap ? 1 : 0), wrap ? 0 : x + 4, dx, dy, e0 + 4 < nflt, raw + 4);
}

HULC_DEVICE void u8_band_chunk3_convert(int W, int e0, int nflt, int dx, const uint32_t* raw, uint4& p0, uint4& p1, uint4& p2) {
    const int rr = fast_div(e0, fast_rcp(W)), x = e0 - rr * W;      // (an integer division by a run-time W is ~25 instructions, four times per item)
    const bool wrap = x + 4 >= W;
    uint32_t a[3][2], b[3][2];
    u8_half4_convert(W, x, dx, e0 < nflt, raw, a);
    u8_half4_convert(W, wrap ? 0 : x + 4, dx, e0 + 4 < nflt, raw + 4, b);
    p0 = make_uint4(a[0][0], a[0][1], b[0][0], b[0][1]);
    p1 = make_uint4(a[1][0], a[1][1], b[1][0], b[1][1]);
    p2 = make_uint4(a[2][0], a[2][1], b[2][0], b[2][1]);
}
