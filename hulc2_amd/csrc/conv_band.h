// conv_band.h — launch parameters shared by the LDS-band convolution kernels (conv_band.hip: 8 waves, one weight set per wave;
// conv_band4.hip: 4 waves, every weight set in every wave).
#pragma once
#include "hulc_common.h"

namespace hulc_band {

#define BAND_MAXCLS 4
struct BandCls {
    int OH, OW;                     // output grid of this weight set's class
    long y_off;                     // element offset of the class inside Y / mask (parity classes of a data gradient)
    int co_base;                    // first output channel of the set's 32-channel tile
    long w_row0;                    // first global weight row of the set
    long w_tap_off[16];             // global offset (elements, inside a weight row) of tap (ty, tx)
};
struct BandP {
    const void* X; void* Y; const void* Wt; const float* bias; const void* mask;
    const void* add;                // optional residual (bf16, laid out like Y), summed before the ReLU: the ResNet trunk's block outputs
    void* Y16;                      // optional (fp32 Y only): a bf16 copy of Y, same layout, from the same accumulators (hulc_conv_desc.y_bf16)
    int x_dtype, y_dtype, w_dtype, mask_dtype;
    int Nimg, H, W;                 // input tensor dims (NHWC, C = template)
    int OHmax, OWmax;               // largest class grid: defines the staged band
    int pad_y, pad_x;               // band origin: input row = oy*S + ty - pad_y
    int R;                          // output rows per work unit
    int F;                          // > 1: a work unit is F whole frames (small maps; then R == OHmax)
    long x_sn, x_sy, x_sx;          // input element strides
    long y_sn, y_sy, y_sx;          // output element strides (channels contiguous)
    long ldw;                       // global weight row stride (elements)
    int relu;
    unsigned* bits_out;             // optional: ReLU sign planes of Y (forward): dword (co / 32) * bplane + pixel, bit = channel % 32
    const unsigned* bits_in;        // optional: sign planes used as the mask (data gradient) instead of `mask`
    int bshift; long bplane;        // log2(channels of the tensor the planes describe), pixels of that tensor: planes are [channels / 32][pixels]
    int lds_band;                   // bytes of one LDS band (the second one of the double-buffered instances starts there)
    int dbg;                        // timing experiments (HULC_BAND_DBG): 1 skip the MFMA loop, 2 skip the output stores, 4 skip band staging
    BandCls cls[BAND_MAXCLS];
};

// bf16 operands only (input band, weights, ReLU mask): a run-time dtype branch around a load makes hipcc wait for it at the join, which
// turned the 32+ weight-fragment loads of a launch and the 10+ chunk loads of every prefetch into as many serial memory round trips.
// Other storage types take the gather kernel in conv.hip.
HULC_DEVICE uint4 band_load_bits(const void* X, long off) { return *(const uint4*)((const uint16_t*)X + off); }
// XF32 instances (an fp32 gradient entering the gripper stack's conv3 data gradient): converted at load time, compile-time selected
template <bool XF32>
HULC_DEVICE uint4 band_load_x(const void* X, long off) {
    if (!XF32) return band_load_bits(X, off);
    const float4* q = (const float4*)((const float*)X + off);
    const float4 a = q[0], c = q[1];
    return make_uint4(pack_bf16x2(a.x, a.y), pack_bf16x2(a.z, a.w), pack_bf16x2(c.x, c.y), pack_bf16x2(c.z, c.w));
}

// conv_band4.hip: 0 = launched, -1 = geometry / options not covered (the caller tries the other kernels), -2 = LDS limit could not be raised
int launch_band4(BandP& p, int C, int NSET, int TH, int TW, int S, hipStream_t s);
// conv_band_planes.hip (direct-to-LDS loads into a chunk-major band, 64 input channels, the static camera's frame sizes): 0 = launched, -1 = not covered
int launch_band_planes(BandP& p, int NSET, int TH, int TW, hipStream_t s);

}  // namespace hulc_band
