/* hulc2_amd.h — C ABI of libhulc2_amd.so: the MI355X (gfx950) kernels under the HULC++ low-level
 * policy training_step (reference: /root/reference/hulc2/models/hulc2.py:336-442).
 *
 * The reference has no FFI of its own: its boundary is the Python class paths Hydra resolves
 * (SURVEY.md §8b).  This header is the new boundary *beneath* those classes; every entry point names
 * the reference code whose arithmetic it replaces.  hulc2_amd/kernels.py binds it with ctypes.
 *
 * Conventions
 *   - plain pointers + sizes; every buffer (inputs, outputs, workspaces) is owned by the caller and
 *     lives in device memory; nothing is allocated, freed or synchronised inside
 *   - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream)
 *   - return 0 on success, a negative code on a rejected call; hulc_last_error() describes it
 *   - dtype codes: HULC_F32 (0) / HULC_BF16 (1); `compute` selects the MFMA arithmetic:
 *     HULC_BF16 = v_mfma_f32_32x32x16_bf16 with fp32 accumulation, HULC_F32 = exact fp32 MFMA
 *   - thread-safe when called on distinct streams
 */
#ifndef HULC2_AMD_H
#define HULC2_AMD_H

#ifdef __cplusplus
extern "C" {
#endif

#define HULC_F32 0
#define HULC_BF16 1

const char* hulc_last_error(void);
int hulc_abi_version(void);

/* ---- dense layers --------------------------------------------------------------------------- */
/* C[M,N] = epi(alpha * A·B^T): A is [M][K] (a_kmajor) or stored [K][M]; B is [N][K] (b_kmajor, the
 * nn.Linear weight layout) or stored [K][N].  Epilogue order: +bias[n], +add[m][n], relu,
 * relu-mask (mask[m][n] > 0 ? v*mask_scale : 0), dropout(drop_p, drop_seed), += C (accumulate).
 * Replaces nn.Linear forward/backward everywhere on the path (plan_proposal_net.py:26-47,
 * plan_recognition_net.py:122-123, goal_encoders.py:21-34,53-71, vision_network.py:49-52,
 * logistic_decoder_rnn.py:60-62,273-277, proj_vis_lang.py:10-21) and the per-timestep recurrence of
 * nn.RNN(relu) (decoders/utils/rnn.py:5-14). */
typedef struct {
    const void* A; const void* B; void* C;
    const float* bias; const void* add; const void* mask;
    int M, N, K;
    long lda, ldb, ldc, ld_add, ld_mask;
    int a_dtype, b_dtype, c_dtype, add_dtype, mask_dtype;
    int a_kmajor, b_kmajor;
    int relu, accumulate;
    float alpha, mask_scale, drop_p;
    unsigned long long drop_seed;
    int compute;
} hulc_gemm_desc;
int hulc_gemm(const hulc_gemm_desc* d, void* stream);


/* ---- convolutions (both camera encoders) ---------------------------------------------------- */
/* Un-padded KHxKW stride-s convolution, replaces nn.Conv2d (+ReLU) of
 * hulc2/models/perceptual_encoders/vision_network.py:36-47 and vision_network_gripper.py:11-20.
 *   x_nchw = 1: x is [N][Cin][H][W] (the batch as the reference delivers it), weights [Cout][Cin*KH*KW]
 *               (= the OIHW parameter, flat), KW must be a power of two >= 8          (conv1)
 *   x_nchw = 0: x is NHWC [N][H][W][Cin], weights [Cout][KH*KW*Cin] (OHWI), Cin power of two >= 8
 * y is always NHWC [N][OH][OW][Cout]; Cout in {32, 64}. */
typedef struct {
    int N, H, W, Cin, Cout, KH, KW, stride;
    int x_nchw;
    int x_dtype, y_dtype, w_dtype;
    int relu;
    int compute;
} hulc_conv_desc;
int hulc_conv2d_fwd(const hulc_conv_desc* d, const void* x, const void* w, const float* bias, void* y, void* stream);
/* dx (NHWC, dtype x_dtype) from dy (NHWC, dtype y_dtype); wt = the weight permuted to [Cin][KH][KW][Cout]
 * (dtype w_dtype); relu_src (same shape/dtype as dx, may be NULL): dx *= (relu_src > 0). */
int hulc_conv2d_bwd_data(const hulc_conv_desc* d, const void* dy, const void* wt, void* dx, const void* relu_src, void* stream);
/* dw [Cout][K] fp32 in the forward k order, db [Cout] fp32 (may be NULL); ws: device workspace of
 * hulc_conv2d_bwd_weight_workspace(d) bytes. */
long hulc_conv2d_bwd_weight_workspace(const hulc_conv_desc* d);
int hulc_conv2d_bwd_weight(const hulc_conv_desc* d, const void* x, const void* dy, float* dw, float* db, void* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif
