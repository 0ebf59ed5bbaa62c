// windows.hip — play-window sampling over the HBM-resident episode store ("window sampling / padding by repetition").
//
// reference behaviour: hulc2/datasets/base_dataset.py:94-112 (__getitem__: window of `size` frames from frame `start`, padded to
// max_window_size), :121-147 (pad_sequence: observations repeat the last frame; relative actions are zero-padded except the
// gripper dimension, which repeats), :149-165 (pad_with_repetition / pad_with_zeros).  The reference materialises every padded
// window on the host; here a window is a row of store indices (frames are read in place by conv1, hulc_conv_desc.frame_index) and
// only the small per-step vectors (actions, proprioception) are gathered.
#include "hulc_common.h"
#include "hulc_abi_internal.h"

namespace {

__global__ void window_index_kernel(const int* starts, const int* sizes, int B, int S, int* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * S) return;
    const int b = i / S, t = i - b * S, n = sizes[b];
    out[i] = starts[b] + (t < n ? t : n - 1);
}

// out[b][t][:] = store[starts[b] + t] for t < sizes[b]; padded steps repeat the last row, except columns [zero_lo, zero_hi) = 0
__global__ void window_rows_kernel(const float* store, int D, const int* starts, const int* sizes, int B, int S, int zero_lo, int zero_hi,
                                   float* out) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long)B * S * D) return;
    const int c = (int)(i % D);
    const long bt = i / D;
    const int b = (int)(bt / S), t = (int)(bt - (long)b * S), n = sizes[b];
    const bool padded = t >= n;
    const long row = (long)starts[b] + (padded ? n - 1 : t);
    out[i] = (padded && c >= zero_lo && c < zero_hi) ? 0.f : store[row * D + c];
}

}  // namespace

extern "C" int hulc_window_index(const int* starts, const int* sizes, int B, int S, int* index_out, void* stream) {
    if (!starts || !sizes || !index_out) return hulc_fail(-1, "hulc_window_index: null pointer");
    if (B <= 0 || S <= 0) return hulc_fail(-2, "hulc_window_index: bad shape");
    window_index_kernel<<<(B * S + 255) / 256, 256, 0, (hipStream_t)stream>>>(starts, sizes, B, S, index_out);
    return hulc_check_launch("hulc_window_index");
}

extern "C" int hulc_window_rows(const float* store, int D, const int* starts, const int* sizes, int B, int S, int zero_lo, int zero_hi,
                                float* out, void* stream) {
    if (!store || !starts || !sizes || !out) return hulc_fail(-1, "hulc_window_rows: null pointer");
    if (B <= 0 || S <= 0 || D <= 0) return hulc_fail(-2, "hulc_window_rows: bad shape");
    const long n = (long)B * S * D;
    window_rows_kernel<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(store, D, starts, sizes, B, S, zero_lo, zero_hi, out);
    return hulc_check_launch("hulc_window_rows");
}
