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
/* (ABI 7) IEEE half precision — accepted in two places only: as hulc_conv2d_fwd's y_dtype next to hulc_conv_desc.y_bf16 (the finer twin of the
 * bf16 output map: 11 bits of mantissa at bf16's two bytes), and as hulc_spatial_softmax_fwd's x_dtype (the consumer of that twin). */
#define HULC_F16 2

const char* hulc_last_error(void);
int hulc_abi_version(void);
/* (ABI 6) Cooperative launches — the ones whose workgroups wait for each other inside the kernel and therefore must all be resident at once:
 * hulc_mlp_chain / hulc_mlp_chain2, hulc_txl_block_fwd / _bwd with shared sequences, the recurrent sweeps — take one workgroup per CU.  Two of
 * them on two streams (two branches of a captured graph) deadlock unless both fit the device TOGETHER: after hulc_set_coop_share(n), n = 2 or 4,
 * the MLP chains run on 256 / n workgroups (error -9 for a chain wider than 16 x that) and the transformer trunk shares a sequence between as
 * many workgroups as keep its grid within 256 / n; the recurrent sweeps are unaffected (never forked).  Read on the host when a launch is
 * issued: a captured graph keeps what it was captured with.  Returns the previous value; n = 1 restores whole-device launches.
 * reference: the prior and the posterior of hulc2/models/hulc2.py:228-233 do not depend on each other. */
int hulc_set_coop_share(int n);
/* (ABI 6) *out = a new non-blocking stream of the current device (hipStreamCreateWithFlags; the caller owns it and may keep it for the life of
 * the process).  For hipGraph captures: a capture that forks onto side streams must start on a stream no earlier capture has used — a
 * framework's pooled streams come round again (torch: 32 per device), and a later capture on such a stream crashed hipGraphLaunch on this
 * runtime (round 6).  No reference counterpart (Lightning never captures graphs). */
int hulc_stream_create(void** out);

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
    int relu, accumulate;          /* relu: 0 = none, 1 = ReLU, 2 = exact (erf) GELU (the MiniLM feed-forward) */
    float alpha, mask_scale, drop_p;
    unsigned long long drop_seed;
    int compute;
    void* ws; long ws_bytes;   /* optional scratch for the split-K paths: the first 16 KiB are tile counters that must be ZERO before the
                                * first use of the buffer (every launch leaves them zero), fp32 slabs follow; not shared by concurrent streams */
    const unsigned long long* seed_dev;   /* optional device word xor-ed into drop_seed (see hulc_step_state_advance) */
    float* rowsum_a; int rowsum_accumulate;   /* optional, M > 64 only: rowsum_a[m] (+)= sum_k A[m][k] in fp32 — with
                                               * A = dY this is the bias gradient of the weight-gradient GEMM dW = dY^T X, fused */
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
    int dw_oihw;        /* bwd_weight only: write dW in the parameter's OIHW order instead of the forward k order (NHWC layers) */
    int dw_accumulate;  /* bwd_weight only: dW and db are added to the destination (gradient arena) instead of overwriting it */
    /* conv1 fed by the frames as stored: x is uint8 NHWC [N][H][W][3]; the kernel applies, while staging,
     * RandomShiftsAug (hulc2/utils/transforms.py:85-106: replicate-pad by aug_pad, integer shift aug_shift[n] = {sx, sy} in
     * [0, 2*aug_pad]; NULL = no shift), ScaleImageTensor (/255, :8-19) and Normalize(0.5, 0.5) (conf/datamodule/transforms/
     * rand_shift.yaml:7-10): value = (u8[clamp(y+sy-pad)][clamp(x+sx-pad)][c] / 255 - 0.5) / 0.5.  x_nchw must be 1 (k = (c,kh,kw)). */
    int x_u8_nhwc;
    int aug_pad;
    const int* aug_shift;
    /* x_u8_nhwc only: batch frame n is frame frame_index[n] of x — x is then the HBM-resident episode store, not a gathered batch
     * (hulc2/datasets/shm_dataset.py:101-118 reads its windows out of shared memory the same way; pad-by-repetition,
     * base_dataset.py:150-155, is a repeated index).  NULL = frame n. */
    const int* frame_index;
    /* (ABI 3) ReLU sign bit-planes: one bit per output channel, Cout / 32 planes of one dword per pixel, [Cout/32][N*OH*OW], bit c % 32 of plane
     * c / 32 = (y > 0).
     * hulc_conv2d_fwd (relu != 0, bf16 output): written next to y when non-NULL.  hulc_conv2d_bwd_data: the planes of the layer INPUT
     * (Cin / 32 planes over the N*H*W input pixels), used in place of relu_src where the LDS-band kernel takes the launch — the data gradient of conv2 then
     * reads 20 MB of sign bits instead of the 315 MB activation (relu_src stays the mask of the other kernels: pass both).  NULL = none. */
    void* relu_bits;
    /* (ABI 3) hulc_conv2d_fwd, conv1 (fp32 NCHW frames, 3 -> 32, 8 x 8 stride 4) on the LDS-band kernel only: bf16 of the rounding remainders
     * w - bf16(w), laid out like w.  The frames are then split into hi + lo bf16 planes while staged and every product is formed from the
     * splits of both operands (three bf16 MFMAs): fp32-class outputs on the bf16 matrix pipe — the selective-precision site "conv1"
     * (DESIGN §5).  NULL = plain bf16 operands. */
    const void* w_lo;
    /* (ABI 4) conv1 on fp32 NCHW frames — or uint8 NHWC frames without frame_index, aug_shift then holding all N frames' shifts —
     * (hulc_conv2d_fwd and hulc_conv2d_bwd_weight, LDS-band kernels only): a SECOND frame tensor of x's type and geometry — frames
     * n >= n_split are frame n - n_split of x2.  The vision and the language modality of a step are two tensors (hulc2.py:336-361: one
     * batch dict per modality) that are never concatenated (1 GB); with x2 their conv1 runs as ONE launch per direction instead of one per
     * modality (one prologue, one set of weight-gradient slabs, one reduce).  NULL = all N frames in x. */
    const void* x2;
    int n_split;
    /* (ABI 5) conv1 on fp32 NCHW frames (LDS-band kernels, forward and weight gradient): DEVICE slots — each an 8-byte aligned device
     * address holding one device pointer — from which the kernels read the base addresses of the frame tensors when they start, instead of
     * taking them from x / x2 (which then only say "one tensor" / "two tensors" and are checked for alignment).  A captured hipGraph of a
     * training step (hulc2_amd/stepnode.py: the reference's trainer loop, hulc2/training.py:79-82, with a data loader that delivers every
     * batch at new addresses) follows the caller's current batch by updating two pointers instead of copying 1.16 GB of frames into the
     * graph's input buffers.  x_slot NULL = the addresses in x / x2; x2_slot goes with x2 (its address un-offset). */
    const void* x_slot;
    const void* x2_slot;
    /* (ABI 7) hulc_conv2d_fwd with an fp32 or fp16 output (y_dtype = HULC_F32 / HULC_F16; fp16: the direct-to-LDS 64-channel 3 x 3 forward
     * of 23 x 23 frames only — conv3 of the static camera) inside a bf16 step: a bf16 copy of the output map, laid out like y,
     * written from the same accumulators (rounded once, after the ReLU: bit for bit the map a bf16-output launch stores).  The selective-
     * precision site "a3" (DESIGN §5): the consumer of the map's VALUES (the spatial softmax, vision_network.py:74-108; the gripper
     * camera's flatten-linear) reads the fp32 map, the backward pass keeps working on the bf16 one — no second pass over the map and no
     * cast of its gradient.  The LDS-band kernels store both from the epilogue; elsewhere a cast launch follows.  NULL = none. */
    void* y_bf16;
} hulc_conv_desc;
int hulc_conv2d_fwd(const hulc_conv_desc* d, const void* x, const void* w, const float* bias, void* y, void* stream);
/* dx (NHWC, dtype x_dtype) from dy (NHWC, dtype y_dtype); wt = the weight permuted to [Cin][KH][KW][Cout]
 * (dtype w_dtype); relu_src (same shape/dtype as dx, may be NULL): dx *= (relu_src > 0). */
int hulc_conv2d_bwd_data(const hulc_conv_desc* d, const void* dy, const void* wt, void* dx, const void* relu_src, void* stream);
/* dw [Cout][K] fp32 in the forward k order, db [Cout] fp32 (may be NULL); ws: device workspace of
 * hulc_conv2d_bwd_weight_workspace(d) bytes. */
long hulc_conv2d_bwd_weight_workspace(const hulc_conv_desc* d);
int hulc_conv2d_bwd_weight(const hulc_conv_desc* d, const void* x, const void* dy, float* dw, float* db, void* ws, void* stream);

/* ---- reductions of the encoders and the plan-recognition transformer ------------------------ */
/* Spatial softmax over NHWC activations x[N][HW][C] (C <= 64): out[N][2C] = interleaved (E[xmap], E[ymap])
 * per channel, stats[N][C][2] = (max, sum) kept for the backward.  Replaces SpatialSoftmax.forward,
 * vision_network.py:100-108 (xmap/ymap/temperature are that module's registered buffers).
 * bwd writes dx (same layout as x) and, with relu_mask, multiplies by (x > 0) — the ReLU after conv3. */
int hulc_spatial_softmax_fwd(const void* x, int x_dtype, int N, int HW, int C, const float* xmap, const float* ymap,
                             const float* temperature, float* out, float* stats, void* stream);
int hulc_spatial_softmax_bwd(const void* x, int x_dtype, int N, int HW, int C, const float* xmap, const float* ymap,
                             const float* temperature, const float* out, const float* stats, const float* dout,
                             void* dx, int dx_dtype, int relu_mask, void* stream);
/* y = LayerNorm(x + dropout(o)) over rows of D <= 256 (o may be NULL; pre_out receives x + dropout(o)).
 * Replaces nn.LayerNorm (vision_network.py:53, goal_encoders.py:28,61) and the residual+norm of the
 * post-norm nn.TransformerEncoderLayer (plan_recognition_net.py:115-117).  bwd: dpre = grad of the LN
 * input, do_out (optional) = dpre * dropout mask (grad of o), dgamma/dbeta summed deterministically. */
int hulc_layernorm_fwd(const float* x, const float* o, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                       const float* gamma, const float* beta, float eps, int R, int D, float* pre_out, float* y, float* mean,
                       float* rstd, void* stream);
/* the same with the residual branch given as n_o partial slabs o[s] (R x D each, o_stride elements apart) summed in order before the
 * dropout: the hidden-slice partials hulc_ffn_fwd leaves in its workspace (f == NULL) — slice sum + residual + LayerNorm in one launch */
int hulc_layernorm_slab_fwd(const float* x, const float* o, int n_o, long o_stride, float drop_p, unsigned long long seed,
                            const unsigned long long* seed_dev, const float* gamma, const float* beta, float eps, int R, int D,
                            float* pre_out, float* y, float* mean, float* rstd, void* stream);
/* (ABI 3) the same LayerNorm (no residual branch) writing row r at y + r * ld_y: several LayerNorms fill disjoint column / row blocks of one
 * tensor — the cameras' halves of the perceptual embedding (concat_encoders.py:96-107), the modalities' latent goals — with no concat copy;
 * backward reads its block of the wider gradient in place (ld_dy). */
int hulc_layernorm_fwd_ld(const float* x, const float* gamma, const float* beta, float eps, int R, int D, float* y, long ld_y, float* mean,
                          float* rstd, void* stream);
int hulc_layernorm_bwd_ld(const float* dy, long ld_dy, const float* pre, const float* mean, const float* rstd, const float* gamma, int R, int D,
                          float* dpre, float* dgamma, float* dbeta, int accumulate_params, void* ws, void* stream);
long hulc_layernorm_bwd_workspace(int R, int D);
int hulc_layernorm_bwd(const float* dy, const float* pre, const float* mean, const float* rstd, const float* gamma, int R, int D,
                       float* dpre, float* do_out, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                       float* dgamma, float* dbeta, int accumulate_params, void* ws, void* stream);
/* accumulate_params != 0: dgamma / dbeta are added to the destination (the gradient arena) instead of overwriting it */
/* out[n] (+)= sum_m x[m*ld + n]: bias gradients of nn.Linear / position-embedding gradient. */
long hulc_colsum_workspace(long M, int N);
int hulc_colsum(const void* x, int x_dtype, long M, int N, long ld, float* out, int accumulate, void* ws, void* stream);
/* y = scale * mean over the sequence axis of x[B][S][D] (plan_recognition_net.py:145; scale = S gives the
 * plain sum used by the decoder backward) and the gradient of the mean. */
int hulc_seq_mean_fwd(const float* x, float* y, int B, int S, int D, float scale, void* stream);
int hulc_seq_mean_bwd(const float* dy, float* dx, int B, int S, int D, void* stream);
/* y[b*ldy + d] = scale * sum_s x[b*stride_b + s*stride_s + d]: time sums over the time-major recurrent buffers
 * (gradient of the per-sequence constant part of the decoder's input projection). */
int hulc_strided_seq_sum(const void* x, int x_dtype, float* y, int B, int S, int D, long stride_b, long stride_s, long ldy,
                         float scale, void* stream);
/* y = dropout(x[B][S][D] + pos[pos_ids[s]][:]) (plan_recognition_net.py:133-136,142); dropout_bwd: dx = dy * mask. */
int hulc_add_pos_fwd(const float* x, const float* pos, const long* pos_ids, float* y, int B, int S, int D, float drop_p,
                     unsigned long long seed, const unsigned long long* seed_dev, void* stream);
int hulc_dropout_bwd(const float* dy, float* dx, long n, float drop_p, unsigned long long seed, const unsigned long long* seed_dev,
                     void* stream);
/* dx = dy * (y > 0) * scale: gradient through ReLU (+ inverted dropout) from the saved activation. */
int hulc_relu_bwd(const float* dy, const void* y, int y_dtype, float* dx, long n, float scale, void* stream);
/* Multi-head self-attention for S <= 32, head_dim 16: qkv[B*S][3E] (token b*S+s; q|k|v), out[B*S][E],
 * probs[B][H][S][S] (post-dropout).  Replaces nn.MultiheadAttention inside nn.TransformerEncoderLayer. */
int hulc_attention_fwd(const float* qkv, float* out, float* probs, int B, int S, int H, int head_dim, float drop_p,
                       unsigned long long seed, const unsigned long long* seed_dev, void* stream);
int hulc_attention_bwd(const float* qkv, const float* probs, const float* dout, float* dqkv, int B, int S, int H, int head_dim,
                       float drop_p, unsigned long long seed, const unsigned long long* seed_dev, void* stream);

/* ---- losses ---------------------------------------------------------------------------------- */
/* Discretised logistic mixture NLL + gripper cross-entropy over y[T][ld] = [logit_probs(A*n_mix) |
 * means | log_scales | gripper(2)] and act[T][A+1].  The T tokens form nseg equal segments (one per modality when
 * both are batched through the decoder); out (3, nseg) planar = totals | nll means | ce means over the segment's tokens,
 * gout[seg] is the upstream gradient of the total out[seg].
 * Replaces LogisticDecoderRNN._loss/_logistic_loss, logistic_decoder_rnn.py:133-152,181-228. */
typedef struct {
    int T, A, n_mix, num_classes, nseg;
    long ld;
    float log_scale_min, gripper_alpha;
    const float* act_min; const float* act_max;   /* device [A] */
    int time_major_B;   /* 0: rows are batch-major, segment seg = rows [seg T/nseg, (seg+1) T/nseg); B > 0: rows are time-major (row = step * B +
                         * batch row, what the recurrent kernel leaves behind) and segment seg = batch rows [seg B/nseg, (seg+1) B/nseg) of every step */
} hulc_mix_desc;
long hulc_mix_loss_workspace(const hulc_mix_desc* d);
int hulc_mix_loss_fwd(const hulc_mix_desc* d, const float* y, const float* act, float* out, void* ws, void* stream);
int hulc_mix_loss_bwd(const hulc_mix_desc* d, const float* y, const float* act, const float* gout, float* dy, long ld_dy, void* stream);
/* KL-balanced categorical KL (hulc2.py:444-466): out[seg] = beta * mean over the segment's rows of sum_g KL(post_g || prior_g);
 * the B rows are nseg equal segments (the modalities of a step batched together; nseg = 1: one mean over all rows).
 * bwd: gout[seg] upstream; dpp gets mix * d/d prior, dpr gets (1 - mix) * d/d posterior. pp/pr: [B][G*CLS], CLS == 32. */
int hulc_cat_kl_fwd(const float* pp, const float* pr, int B, int G, int CLS, float beta, int nseg, float* out, float* kl_group, void* stream);
int hulc_cat_kl_bwd(const float* pp, const float* pr, const float* kl_group, int B, int G, int CLS, float beta, float mix,
                    const float* gout, int nseg, float* dpp, float* dpr, void* stream);
/* Straight-through one-hot sample of each of NG categorical groups (distributions.py:23-27, hulc2.py:235-237):
 * plan = one_hot(idx); idx from idx_in (injected) or inverse-CDF sampling with the counter RNG. */
int hulc_plan_sample_fwd(const float* logits, const long* idx_in, unsigned long long seed, const unsigned long long* seed_dev,
                         int NG, int CLS, long* idx_out, float* plan, void* stream);
int hulc_plan_sample_bwd(const float* logits, const float* dplan, int NG, int CLS, float* dlogits, int accumulate, void* stream);
/* CLIP-style symmetric contrastive loss on projected features im/tx [M][32] restricted to rows with
 * use[m] != 0 (hulc2.py:472-508); dscale = d loss / d logit_scale.  out[2] = {loss, number of rows with use != 0 (1 when none): the weight
 * `batch_size["aux_lang"]` the step logs the loss with, hulc2.py:391-394 — a device value, no host synchronisation}. */
int hulc_clip_loss_fwd(const float* im, const float* tx, const unsigned char* use, int row0, const float* logit_scale, int M, int D,
                       float* out, void* stream);
int hulc_clip_loss_bwd(const float* im, const float* tx, const unsigned char* use, int row0, const float* logit_scale, int M, int D,
                       const float* gout, float* dim, float* dtx, float* dscale, void* stream);
/* (ABI 3) row0: rows 0 .. row0-1 of im / tx never take part and use[] describes rows row0 .. M-1 — the stacked modalities of a step go through
 * the projection heads together, only the language rows enter the loss (no slicing of the pooled features, no gradient scatter). */
/* The scalar tail of Hulc2.training_step (hulc2.py:400-430): total = (sum_m action_loss[m] + sum_m kl_loss[m]) / n + beta * clip (clip may be
 * NULL).  out (4 + n) = {total, kl mean, action mean, beta * clip, action_loss[m] + kl_loss[m] ...}: everything the step logs.  bwd: g = d total. */
int hulc_loss_combine_fwd(const float* kls, const float* acts, const float* clip, int n, float beta, float* out, void* stream);
int hulc_loss_combine_bwd(const float* g, int n, float beta, float* dkls, float* dacts, float* dclip, void* stream);
/* Fan-out of the perceptual embedding emb (N, S, D) to its consumers (hulc2.py:380-387, :228-231): e0 (N, D) = emb[:, 0] (prior),
 * elast (n_last, D) = emb[:n_last, -1] (visual goal encoder), edec_t (S, N, hi - lo) = emb[:, :, lo:hi] time-major (action decoder,
 * logistic_decoder_rnn.py:262-266); the posterior reads emb itself.  bwd: demb = g_rec + the three scattered gradients (any may be NULL). */
int hulc_emb_fanout_fwd(const float* emb, int N, int S, int D, int n_last, int lo, int hi, float* e0, float* elast, float* edec_t, void* stream);
int hulc_emb_fanin_bwd(const float* g_rec, const float* g0, const float* g_last, const float* g_dec_t, int N, int S, int D, int n_last, int lo, int hi,
                       float* demb, void* stream);
/* Relative actions world -> tcp frame, act[n][7], robot_obs[n][obs_dim] (euler angles in 3:6),
 * gripper_control.py:16-36 (pytorch3d XYZ convention restated; parity unpinned). */
int hulc_world_to_tcp(const float* act, const float* robot_obs, int n, int obs_dim, float* out, void* stream);
/* The decoder's targets of a step in the time-major row order of the recurrent kernel's outputs: nseg (<= 4) modality batches act[i] (B, S, 7)
 * with robot_obs[i] (B, S, obs_dim) -> out row (s * nseg * B + i * B + b); to_tcp != 0 applies world_to_tcp_frame (gripper_control.py:16-36) on the
 * way, else the rows are copied (logistic_decoder_rnn.py:118-131 with gripper_control false).  act / robot_obs: HOST arrays of device pointers. */
int hulc_actions_time_major(const float* const* act, const float* const* robot_obs, int nseg, int B, int S, int obs_dim, int to_tcp, float* out,
                            void* stream);
/* inverse frame change for sampled actions, tcp_to_world_frame (gripper_control.py:39-63; without the NaN quaternion fallback) */
int hulc_tcp_to_world(const float* act, const float* robot_obs, int n, int obs_dim, float* out, void* stream);
/* LogisticDecoderRNN._sample (logistic_decoder_rnn.py:231-255) on the fused head output y (T rows: [logit_probs | means |
 * log_scales | gripper 2], row stride d->ld): Gumbel-max over the n_mix mixtures, inverse-CDF draw from the selected logistic,
 * gripper_bounds[argmax].  u_mix (T, A, n_mix) / u_inv (T, A): optional raw uniforms in [0,1) standing in for torch.rand
 * (parity tests); NULL -> counter RNG on (seed ^ *seed_dev).  act_out (T, A+1); idx_out (T, A) selected mixture (optional). */
int hulc_mix_sample(const hulc_mix_desc* d, const float* y, const float* u_mix, const float* u_inv, unsigned long long seed,
                    const unsigned long long* seed_dev, const float* gripper_bounds, float* act_out, long* idx_out, void* stream);

/* ---- play windows over the HBM-resident episode store ---------------------------------- */
/* A window = `sizes[b]` consecutive store frames from `starts[b]`, padded to S steps (hulc2/datasets/base_dataset.py:94-112).
 * hulc_window_index: index_out[b][t] = starts[b] + min(t, sizes[b] - 1) — pad_with_repetition (base_dataset.py:149-154) as a repeated
 * index; feed it to hulc_conv_desc.frame_index so conv1 reads the store in place.
 * hulc_window_rows: out (B, S, D) fp32 gathered from store (n_frames, D): padded steps repeat the last row, except columns
 * [zero_lo, zero_hi) which are zero — relative actions pad with zeros in dims 0..5 and repeat the gripper dim (base_dataset.py:132-142);
 * observations / state_info repeat everything (zero_lo = zero_hi = 0).  1 <= sizes[b] <= S; starts/sizes are device int32. */
int hulc_window_index(const int* starts, const int* sizes, int B, int S, int* index_out, void* stream);
int hulc_window_rows(const float* store, int D, const int* starts, const int* sizes, int B, int S, int zero_lo, int zero_hi,
                     float* out, void* stream);

/* ---- sentence encoder: paraphrase-MiniLM-L3-v2 = BERT (3 layers, hidden 384, 12 heads, GELU, eps 1e-12) + mean
 * pooling, as sentence_transformers runs it for hulc2/affordance/models/language_encoders/sbert_lang_encoder.py:13-71 (frozen, inference
 * only).  The dense layers are hulc_gemm (relu = 2 for the intermediate GELU); these are the pieces around them, all fp32 in/out. */
/* out[t] = LayerNorm(word[ids[t]] + pos[t % S] + type0), T tokens of D <= 1024 features (BertEmbeddings) */
int hulc_embed_ln_fwd(const long* ids, const float* word, const float* pos, const float* type0, const float* gamma, const float* beta,
                      float eps, int T, int S, int D, float* out, void* stream);
/* y[r] = LayerNorm(x[r] + add[r]) over D <= 1024 (add may be NULL): BertSelfOutput / BertOutput */
int hulc_ln_wide_fwd(const float* x, const float* add, const float* gamma, const float* beta, float eps, int R, int D, float* y, void* stream);
/* multi-head self attention with a key padding mask: qkv (B*S, 3*nhead*hd) = [q | k | v], mask (B, S) int32 (1 = token), out (B*S, nhead*hd);
 * softmax(q k^T / sqrt(hd) + (1 - mask) * finfo(float).min) v as BertSelfAttention; S <= 128, hd <= 64 */
int hulc_mha_masked_fwd(const float* qkv, const int* mask, int B, int S, int nhead, int hd, float* out, void* stream);
/* out[b] = sum_s mask[b][s] x[b][s] / max(sum_s mask[b][s], 1e-9): sentence_transformers' mean Pooling */
int hulc_masked_mean_fwd(const float* x, const int* mask, int B, int S, int D, float* out, void* stream);

/* ---- frozen ResNet-18 trunk of VisionR3M ---------------------------------------- */
/* hulc2/models/perceptual_encoders/vision_r3m.py:24-27 runs `self.r3m(x)` under no_grad on frames in [0, 255].  r3m (un-vendored
 * submodule, parity unpinned — SURVEY §8c) computes obs / 255 -> Normalize(ImageNet) -> torchvision resnet18 with fc = Identity.
 * hulc_r3m_normalize: x fp32 NCHW [N][3][H][W] -> y NHWC [N][H][W][8] (dtype y_dtype), channel c < 3 = (x / 255 - mean3[c]) / std3[c],
 * channels 3..7 zero (one 16-byte bf16 chunk per pixel; the stem's weights are zero-padded to match).  mean3 / std3: HOST arrays.
 * hulc_conv2d_padded_fwd: zero-padded convolution, NHWC x [N][H][W][Cin] (Cin a power of two >= 8), w [Cout][KH*KW*Cin] with the
 * BatchNorm scale folded in, bias = the folded shift, y = [relu](conv + bias [+ add]) NHWC, add (optional residual) shaped and typed
 * like y; Cout a multiple of 32; uses d->{N,H,W,Cin,Cout,KH,KW,stride,x_dtype,y_dtype,w_dtype,relu,compute}.
 * hulc_maxpool_nhwc: nn.MaxPool2d(k, stride, pad) on NHWC (C % 8 == 0), same dtype in and out.
 * The global average pool is hulc_strided_seq_sum over the H*W axis with scale 1 / (H*W). */
int hulc_r3m_normalize(const float* x, int N, int H, int W, const float* mean3, const float* std3, void* y, int y_dtype, void* stream);
int hulc_conv2d_padded_fwd(const hulc_conv_desc* d, int pad, const void* x, const void* w, const float* bias, const void* add, void* y,
                           void* stream);
int hulc_maxpool_nhwc(const void* x, int dtype, int N, int H, int W, int C, int k, int stride, int pad, void* y, void* stream);
/* (ABI 4) nn.BatchNorm2d in TRAINING mode over NHWC rows z [M][C] (M = N*H*W pixels), forward only: y = relu?(gamma (z - mean_batch) /
 * sqrt(var_biased + eps) + beta + add); run_mean / run_var (optional) are updated with `momentum` and the unbiased variance.  The affordance
 * model's trunk as the reference runs it (hulc2/affordance/models/visual_lang_encoders/r3m_rn18.py:27-43 freezes the PARAMETERS of
 * layer1..4 only; pixel_aff_lang_detector.py:51-53 leaves train mode on, so the trunk's BatchNorms use batch statistics).  z / add / y: fp32
 * or bf16; C a multiple of 8 (a divisor of 256 below 256); ws: hulc_nhwc_bn_train_workspace(M, C) bytes.  Fixed summation order. */
long hulc_nhwc_bn_train_workspace(long M, int C);
int hulc_nhwc_bn_train_fwd(const void* z, int z_dtype, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                           float* run_mean, float* run_var, const void* add, int add_dtype, int relu, void* y, int y_dtype, void* ws, void* stream);
/* (ABI 5) The same launch, additionally leaving the batch mean and 1 / sqrt(var + eps) in saved[0..C) / saved[C..2C) for the backward. */
int hulc_nhwc_bn_train_fwd_saved(const void* z, int z_dtype, long M, int C, const float* gamma, const float* beta, float eps, float momentum,
                                 float* run_mean, float* run_var, const void* add, int add_dtype, int relu, void* y, int y_dtype, float* saved, void* ws,
                                 void* stream);
/* (ABI 5) Backward of nn.BatchNorm2d in training mode (+ the ReLU behind it when y is given) over NHWC rows — the data gradient through the
 * FROZEN layers of the affordance trunk, which the reference's trainable stem needs (r3m_rn18.py:34-38 freezes layer1..4 only; the modules stay
 * in train mode, pixel_aff_lang_detector.py:51-53): g = dy * (y > 0), dz = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)) with
 * xhat = (z - mean) * rstd from the forward's `saved`; g_out (optional) receives g — the gradient of the block's shortcut branch;
 * dgamma / dbeta (optional; accumulate_params: added to) = sum g * xhat / sum g (the stem's bn1).  z fp32 (the bias-free convolution's output),
 * dz / g_out in o_dtype; ws = hulc_nhwc_bn_train_workspace(M, C) bytes.  Fixed summation order: bit-reproducible. */
int hulc_nhwc_bn_train_bwd(const void* dy, int dy_dtype, const void* y, int y_dtype, const float* z, long M, int C, const float* gamma,
                           const float* saved, void* dz, void* g_out, int o_dtype, float* dgamma, float* dbeta, int accumulate_params, void* ws,
                           void* stream);
/* (ABI 5) Backward of hulc_maxpool_nhwc: dx[pixel] = sum of dy over the windows whose first maximum (scan order kh, kw, strict >: the index
 * nn.MaxPool2d records) is that pixel; x, dy, dx share `dtype`; a gather, no atomics. */
int hulc_maxpool_nhwc_bwd(const void* x, const void* dy, int dtype, int N, int H, int W, int C, int k, int stride, int pad, void* dx, void* stream);
/* (ABI 5) y [N][Hy][Wy][C] = x [N][H][W][C] placed at (off + step * row, off + step * col), zeros elsewhere: step 2 = the zero-inserted gradient
 * of a stride-2 convolution (whose data gradient is then a stride-1 convolution with flipped taps), step 1 = zero padding by `off`. */
int hulc_nhwc_scatter(const void* x, int dtype, int N, int H, int W, int C, int Hy, int Wy, int step, int off, void* y, void* stream);
/* The stem (7x7, stride 2, padding 3, 3 -> Cout) on a packed input, bf16: hulc_r3m_normalize_packed writes xp [N][H+6][Wp][4] with
 * Wp = hulc_r3m_packed_width(W), pixel (y, x) at [y+3][x+3] = normalised (R, G, B, 0), zero border; hulc_r3m_stem_fwd computes
 * y NHWC [N][OH][OW][Cout] = [relu](conv + bias) from it with w [Cout][7][8][4] bf16 = the (BatchNorm-folded) stem weight as [o][kh][kw][c],
 * zero at kw = 7 and c = 3: two neighbouring pixels are one aligned 16-byte piece, K = 224 instead of 392, no bounds checks. */
int hulc_r3m_packed_width(int W);
int hulc_r3m_normalize_packed(const float* x, int N, int H, int W, const float* mean3, const float* std3, void* xp, void* stream);
int hulc_r3m_stem_fwd(const void* xp, const void* w, const float* bias, void* y, int y_dtype, int N, int H, int W, int Cout, int relu, void* stream);

/* ---- transformer feed-forward block, fused (bf16 compute) ----------------------------------------------- */
/* f = relu(x W1^T + b1) [dropout] W2^T + b2 of nn.TransformerEncoderLayer (plan_recognition_net.py:108-117), d_model 128,
 * dim_feedforward FF (multiple of 128); the (T x FF) hidden activation stays on chip, backward recomputes it.
 * x, f, df, dx: (T, 128) fp32; W1 [FF][128], W2 [128][FF] bf16; W1T = W1^T [128][FF], W2T = W2^T [FF][128] bf16.
 * Dropout: counter RNG on (seed ^ *seed_dev, token*FF + unit), the stream hulc_gemm's epilogue uses for the unfused block.
 * bwd: dx (+)= df-path input gradient; dW1 [FF][128], db1 [FF], dW2 [128][FF] (+)= when accumulate_params (db2 = column sums of df:
 * hulc_colsum).  ws: hulc_ffn_workspace(T, FF) bytes.  f == NULL (fwd) / dx == NULL (bwd): skip the pass that sums the FF / 128
 * hidden-slice partials; they stay at the start of ws as [FF / 128][T][128] fp32 (b2 rides in slice 0) for a consumer that sums them
 * itself (hulc_layernorm_slab_fwd, hulc_txl_attn_bwd). */
long hulc_ffn_workspace(int T, int FF);
int hulc_ffn_fwd(const float* x, const void* W1, const float* b1, const void* W2, const float* b2, int T, int D, int FF, float drop_p,
                 unsigned long long seed, const unsigned long long* seed_dev, float* f, void* ws, void* stream);
int hulc_ffn_bwd(const float* x, const float* df, const void* W1, const float* b1, const void* W1T, const void* W2T, int T, int D, int FF,
                 float drop_p, unsigned long long seed, const unsigned long long* seed_dev, float* dx, int dx_accumulate,
                 float* dW1, float* db1, float* dW2, int accumulate_params, void* ws, void* stream);

/* ---- transformer layer, attention half, fused (bf16 compute) ----------------------------------------------- */
/* y1 = LayerNorm1(x + dropout(out_proj(MHA(x)))) of the post-norm nn.TransformerEncoderLayer behind PlanRecognitionTransformersNetwork
 * (hulc2/models/plan_encoders/plan_recognition_net.py:108-121; d_model E = 128, H = 8 heads of 16, S <= 32 tokens per sequence) as ONE
 * launch: workgroup = sequence, wave = 2 heads; in_proj, QK^T (one 32x32x16 MFMA per head), softmax + dropout in registers, PV,
 * out_proj, residual + dropout + LayerNorm.  Rows of x are tokens b*S + s.  Weights bf16: Wqkv [3E][E] (in_proj_weight), Wo [E][E];
 * the backward also reads their transposes WqkvT [E][3E], WoT [E][E].
 * fwd : writes y (T, E) fp32 and, when pre != NULL, what backward needs: pre (T, E) fp32 = x + dropout(o), mean / rstd (T), and
 *       ctx (T, E) bf16 (attention output before out_proj: operand of the out_proj weight gradient).
 * bwd : dy (T, E) fp32 [+ n_slab partials dy_slab[s] (T, E) fp32, slab_stride elements apart, summed in order: the hidden-slice
 *       input gradients of hulc_ffn_bwd] -> dx (T, E) fp32 (gradient of x through both the residual and the attention branch),
 *       d_o (T, E) bf16 and dqkv (T, 3E) bf16 (left operands of the weight-gradient GEMMs dWo = d_o^T ctx, dWqkv = dqkv^T x, whose
 *       row sums are the bias gradients) and ln_partial (B, 2, E) fp32 = per-sequence {dgamma, dbeta} sums (hulc_colsum finishes them).
 * Dropout streams are those of hulc_attention_fwd (seed_attn) and hulc_layernorm_fwd (seed_ln): the unfused kernels give the same masks. */
typedef struct hulc_txl_attn_desc {
    const float* x;
    const void *Wqkv, *Wo, *WqkvT, *WoT;
    const float *bqkv, *bo, *gamma, *beta;
    float eps;
    int B, S, H, E;
    float drop_p;
    unsigned long long seed_attn, seed_ln;
    const unsigned long long* seed_dev;
    float *y, *pre, *mean, *rstd;
    void* ctx;
    const float *dy, *dy_slab;
    int n_slab;
    long slab_stride;
    float* dx;
    void *d_o, *dqkv;
    float* ln_partial;
} hulc_txl_attn_desc;
int hulc_ln_partial_reduce(const float* partial, int P, int D, float* dgamma, float* dbeta, int accumulate, void* stream);
/* (ABI 3) the same for n <= 8 LayerNorms in one launch: partial (n, P, 2, D); dgamma / dbeta / accumulate: HOST arrays of n entries
 * (hulc_txl_block_bwd's lnp1 / lnp2 of every layer, carved from one buffer). */
int hulc_ln_partial_reduce_multi(const float* partial, int n, int P, int D, float* const* dgamma, float* const* dbeta, const int* accumulate,
                                 void* stream);
int hulc_txl_attn_fwd(const hulc_txl_attn_desc* d, void* stream);
int hulc_txl_attn_bwd(const hulc_txl_attn_desc* d, void* stream);

/* ---- (ABI 3) Linear(128, H) + ReLU + Linear(H, OUT) over many rows as one launch per direction (bf16 compute) ------------------------ */
/* The head of both camera encoders, fc2(relu(fc1(x))) (vision_network.py:49-52,60-66, vision_network_gripper.py:36-39,52-55), on the T
 * frames of a step: x (T, K = 128) fp32, W1 bf16 [H][128] (H a multiple of 128), W2 bf16 [OUT][H] (OUT in {32, 64, 96, 128}) -> y (T, OUT)
 * fp32; the hidden activation stays in registers.  Backward (W1T [128][H], W2T [H][OUT] = the transposes) recomputes it, writes
 * dx (T, 128) fp32 (null: not wanted) and the bf16 operands h, dh (T, H) of dW1 = dh^T x, dW2 = dy^T h (row sums of dh / dy = the bias
 * gradients; hulc_wgrad_group). */
/* W1_lo / W2_lo (both or none): bf16 of the rounding remainders w - bf16(w), same layouts — the forward then forms both products from hi / lo
 * splits of both operands (three bf16 MFMAs: fp32-class values; the selective-precision site "encfc"). */
int hulc_mlp2_rows_fwd(const float* x, const void* W1, const float* b1, const void* W2, const float* b2, const void* W1_lo, const void* W2_lo,
                       int T, int K, int H, int OUT, float* y, void* stream);
int hulc_mlp2_rows_bwd(const float* x, const float* dy, const void* W1, const float* b1, const void* W1T, const void* W2T, int T, int K, int H, int OUT,
                       float* dx, void* h, void* dh, void* stream);

/* ---- (ABI 3) the whole plan-recognition transformer trunk as ONE launch per direction (bf16 compute) -------------------------------- */
/* PlanRecognitionTransformersNetwork.forward up to the sequence mean (plan_recognition_net.py:125-146): dropout(emb + pos[position_ids]) ->
 * L post-norm nn.TransformerEncoderLayer (d_model 128, 8 heads, ReLU feed-forward FF, dropout drop_p) -> mean over the S <= 32 positions.
 * Every operation is independent per sequence: one workgroup owns one sequence for the whole trunk (hulc_txl_attn_* as its attention
 * stages; the feed-forward block keeps its hidden activation in MFMA registers).  Weights are the bf16 shadows hulc_txl_attn_* / hulc_ffn_*
 * take; the tensors marked "kept" are what backward reads, layer l+1's x IS layer l's y2 (same pointer).  Backward additionally leaves,
 * per layer, the bf16 operands of the weight-gradient products over all T = B S tokens — dWqkv = dqkv^T x, dWo = d_o^T ctx, dW1 = dh^T y1,
 * dW2 = df^T h (row sums of the left operands = the bias gradients; hulc_wgrad_group) — and the per-sequence LayerNorm partials
 * lnp1 / lnp2 (B, 2, E) (hulc_ln_partial_reduce).  Dropout streams: the unfused kernels' (same masks for the same seeds). */
#define HULC_TXL_MAX_LAYERS 4
typedef struct hulc_txl_block_layer {
    const void *Wqkv, *Wo, *W1, *W2;                 /* bf16 [3E][E], [E][E], [FF][E], [E][FF] */
    const void *WqkvT, *WoT, *W1T, *W2T;             /* backward: their transposes */
    const void *W1p, *W2p, *W2Tp, *W1Tp;             /* optional fragment-packed copies (hulc_ffn_frag_perm layouts 0, 1, 2, 3): every weight fragment
                                                        of the feed-forward loops is then one coalesced 16-byte load per lane; null = gather from the matrices */
    const void *Wqkv_lo, *Wo_lo, *W1p_lo, *W2p_lo;   /* optional (all four or none, on every layer or none): bf16 of the rounding remainders w - bf16(w) of Wqkv, Wo
                                                        (same layouts) and of W1, W2 in the packed layouts of W1p / W2p.  The FORWARD launch then forms every product
                                                        from hi / lo splits of both operands (a_hi b_hi + a_lo b_hi + a_hi b_lo, three bf16 MFMAs): forward values of
                                                        fp32 class (~2^-16) on the bf16 matrix pipe — the selective-precision site "txl" */
    const float *bqkv, *bo, *b1, *b2, *g1, *be1, *g2, *be2;
    unsigned long long seed_attn, seed_ln1, seed_ffn, seed_ln2;
    float* x;                                        /* (T, E) layer input; layer 0: written by the forward launch = dropout(emb + pos) */
    float *y1, *pre1, *mean1, *rstd1;                /* kept: attention half's output, LayerNorm1 input and statistics (pre1 .. may be null: inference) */
    void* ctx;                                       /* kept: bf16 (T, E) attention context */
    float *y2, *pre2, *mean2, *rstd2;                /* kept: the layer's output, LayerNorm2 input and statistics */
    void *d_o, *dqkv, *df, *h, *dh;                  /* backward out: bf16 (T, E), (T, 3E), (T, E), (T, FF), (T, FF) */
    float *lnp1, *lnp2;                              /* backward out: (B, 2, E) */
} hulc_txl_block_layer;
typedef struct hulc_txl_block_desc {
    int L, B, S, H, E, FF;
    float eps, drop_p;
    unsigned long long seed_pos;
    const unsigned long long* seed_dev;
    const float* emb;                                /* (B, S, E) */
    const float* pos;                                /* position table (rows, E) */
    const long* pos_ids;                             /* (S,) */
    float* pooled;                                   /* (B, E) */
    const float* dpooled;                            /* backward in: (B, E) */
    float* demb;                                     /* backward out: (B, S, E) gradient of emb (its sum over the batch is the table's) */
    /* Sharing a sequence between 2 or 4 workgroups (each takes a part of the feed-forward hidden units; partial tiles exchanged through ws,
     * one arrival counter per sequence) while all of them fit the device at once: ws = hulc_txl_block_workspace(B, L) bytes whose first
     * 8 KiB (the counters: a fixed-size area, so one buffer serves launches of any B) were ZERO before the first use (every launch that does not time out leaves them zero), exclusive != 0 = the stream runs
     * nothing else concurrently (the members wait for each other: they must be co-resident).  ws null or exclusive 0: one workgroup per
     * sequence, no waiting.  A member that waits too long ORs bit 2 (value 4) into *err_sticky (see hulc_rnn_wave_desc.err_sticky). */
    void* ws;
    int exclusive;
    int* err_sticky;
    hulc_txl_block_layer layers[HULC_TXL_MAX_LAYERS];
} hulc_txl_block_desc;
long hulc_txl_block_workspace(int B, int L);
int hulc_txl_block_fwd(const hulc_txl_block_desc* d, void* stream);
int hulc_txl_block_bwd(const hulc_txl_block_desc* d, void* stream);

/* ---- a stack of Linear(+ReLU) layers on M <= 64 rows as one persistent launch (bf16 compute) -------------------------------------- */
/* y_l = f_l(y_{l-1} W_l^T + b_l), l = 0 .. nl-1 (nl <= 8): the per-sequence MLPs of the policy — PlanProposalNetwork (plan_proposal_net.py:
 * 26-47), the goal encoders (goal_encoders.py:21-34,53-71), ProjVisLang (proj_vis_lang.py:10-21), the posterior's fc -> fc_state
 * (plan_recognition_net.py:122-123,144-148) — and, with W = the transposed weights and mask = the stored activations, the data-gradient
 * chain of their backward (f_l = keep where mask > 0, scaled by mask_scale).  x0: fp32 (M, K0 <= 4096) row-major (the launch's first stage
 * rounds it to bf16 in the layout the layers exchange their outputs in); W_l: bf16 [N_l][K_l] with
 * K_l = N_{l-1}; every layer's fp32 output goes to out_l (M, N_l).  N_l: multiple of 16, <= 4096; K: multiple of 8 rounding up to
 * 128 x {1, 2, 3, 4, 8, 16, 32}.  256 workgroups meet at a device-wide barrier between layers: the stream must not run another kernel
 * concurrently; a barrier timeout ORs bit 1 (value 2) into *err_sticky (see hulc_rnn_wave_desc.err_sticky).  ws: hulc_mlp_chain_workspace(d) bytes
 * whose first 36 KiB (barrier counters) were ZERO before the first use of the buffer; every launch that does not time out leaves them zero,
 * so one buffer serves all launches of a stream (a single launch, no memset). */
typedef struct hulc_mlp_chain_layer {
    const void* W; long ldw;
    const void* W_lo;                                /* (ABI 3) optional, on every layer of a chain or none: bf16 of the remainders w - bf16(w), laid out like W.
                                                        The chain — the SECOND one of hulc_mlp_chain2, or a single chain of <= 32 rows — then forms its products
                                                        from hi / lo splits of both operands (three MFMAs: fp32-class values; precision site "goal") */
    const float* bias;
    const float* mask; long ld_mask; float mask_scale;
    float* out; long ld_out;
    int N, relu;
} hulc_mlp_chain_layer;
typedef struct hulc_mlp_chain_desc {
    int nl, M, K0;
    const float* x0; long ld_x0;
    hulc_mlp_chain_layer layers[8];
} hulc_mlp_chain_desc;
long hulc_mlp_chain_workspace(const hulc_mlp_chain_desc* d);
int hulc_mlp_chain(const hulc_mlp_chain_desc* d, void* ws, int* err_sticky, void* stream);
/* (ABI 3) two independent chains, <= 32 rows each, as ONE launch: VisualGoalEncoder.mlp + LanguageGoalEncoder.mlp (goal_encoders.py:21-34 /
 * :53-71; own weights, inputs of different width) and the data-gradient chains of their backward.  b->nl <= a->nl, and layer l of b has
 * layer l of a's width N (b sits out a's trailing layers).  ws: hulc_mlp_chain_workspace(a) + hulc_mlp_chain_workspace(b) bytes. */
int hulc_mlp_chain2(const hulc_mlp_chain_desc* a, const hulc_mlp_chain_desc* b, void* ws, int* err_sticky, void* stream);

/* ---- all small weight gradients of a backward pass in one launch (csrc/wgrad_group.hip) ------------------- */
/* For every item i:  C[M][N] (+)= sum_k A[k][m] B[k][n]  and, when rowsum != NULL,  rowsum[m] (+)= sum_k A[k][m]  — the weight and bias
 * gradient of an nn.Linear (A = gradient of its output, B = its input, both row-major with the token index k outer; reference: what
 * autograd computes for plan_proposal_net.py:26-47, goal_encoders.py:21-34,53-71, plan_recognition_net.py:115-148, vision_network.py:43-47,
 * logistic_decoder_rnn.py:81-84).  bf16 MFMA with fp32 accumulation (fp32 operands are rounded to bf16 as hulc_gemm does; the bias sums use
 * the unrounded values); split-K partials are summed in slice order, the result does not depend on scheduling.  M, N multiples of 8,
 * K a multiple of 32, operand rows 16-byte aligned.  Two items of one call must not write the same C / rowsum.
 * ws: hulc_wgrad_group_workspace(items, n) bytes whose first 256 KiB (tile counters) were zero before the FIRST use of the buffer; every
 * launch leaves them zero again. */
typedef struct hulc_wgrad_item {
    const void* A; const void* B; float* C; float* rowsum;
    int M, N, K;
    int lda, ldb, ldc;
    int a_dtype, b_dtype;            /* HULC_F32 / HULC_BF16 */
    int accumulate, rowsum_accumulate;
    int col_perm;                    /* > 0: column n of the product is stored at column (n % col_perm) * (N / col_perm) + n / col_perm — the
                                      * (h, w, c) -> (c, h, w) order of a Linear that follows nn.Flatten on an NHWC map (vision_network_gripper.py:16-17) */
    int col_mul;                     /* > 1 (with col_perm 0): column n of the product is stored at column n * col_mul — one tap of a convolution
                                      * weight kept as OIHW: C points at tap t, col_mul = KH KW */
    int store_rows;                  /* > 0: only rows m < store_rows of the product (and of rowsum) are stored — A padded with zero columns */
    int conv_taps_wp;                /* > 0 (with col_mul 9, no rowsum): the item is the nine taps of a 3 x 3 convolution's weight gradient on the padded
                                      * grid of width conv_taps_wp = W + 2 — tap u = 3 (dy + 1) + (dx + 1) multiplies A with B shifted by dy conv_taps_wp + dx
                                      * rows and writes C + u.  bf16 operands with M, N multiples of 64 (or 32) run as nine-tap tiles (wgrad_taps.hip: dZ staged once and X as
                                      * three row windows per k-step, a second launch sums the fp32 slabs of split tiles in a fixed order); other shapes as
                                      * nine neighbouring 64 x 64 tiles of the grouped kernel */
} hulc_wgrad_item;
long hulc_wgrad_group_workspace(const hulc_wgrad_item* items, int n);
int hulc_wgrad_group(const hulc_wgrad_item* items, int n, void* ws, long ws_bytes, void* stream);

/* ---- affordance model (BASELINE configs[4]): the trainable part behind the frozen R3M trunk ---------------------------- */
/* PixelAffLangDetector.training_step (hulc2/affordance/pixel_aff_lang_detector.py:51-69) in the shipped variant
 * (conf/affordance/aff_detection/r3m.yaml): UnetLangFusionDecoder (models/core/unet_decoder.py:83-146), the segmentation head
 * (visual_lang_encoders/r3m_rn18.py:64-69), cross_entropy_with_logits over the pixels (utils/losses.py:6-13).
 * PADDED GRID layout: a map (N, H, W, C) is stored as rows of C bf16 channels on the grid (N, H + 2, W + 2) — pixel (n, y, x) is grid row
 * n (H+2)(W+2) + (y+1)(W+2) + (x+1), border rows are ZERO, and at least W + 3 zero rows precede and follow the tensor (pointers below address
 * grid row 0).  A 3 x 3 / padding-1 convolution is then a shifted GEMM without bounds tests, its data gradient the same kernel with flipped
 * taps and transposed weights, its weight gradient nine items of hulc_wgrad_group (A = dY rows, B = X rows shifted by the tap, col_mul = 9).
 * hulc_gridconv3x3: y[r][co] = sum_{t, ci} x[r + off_t][ci] wt[co][t * Cin + ci] (wt bf16 [Cout][9 Cin], Cin / Cout multiples of 32; flip_taps:
 *   off_{8-t} instead of off_t — the data gradient reading the UNflipped weights as [ci][kh][kw][co]), border
 *   rows of y forced to zero; stats (optional, hulc_gridconv_stats_bytes): per row-tile partial sums of y and y^2 over the pixels from the fp32
 *   accumulators; out0 (optional): fp32 [rows] = channel 0 + bias0[0] (the one-channel head; y may then be NULL).
 * hulc_grid_bn_finalize: nn.BatchNorm2d in training mode from those partials (nb = row tiles, count = N H W): bn[4][C] = mean, rstd,
 *   scale = gamma rstd, shift = beta - mean scale; running statistics updated (momentum, unbiased variance) when given.  mid (32 x 2 x C floats)
 *   + counters (C / 64 words, zero before the first use, left zero): the sum is spread over up to 32 workgroups per 64 channels, the last to
 *   arrive finalises (both NULL: one workgroup per 64 channels).
 * hulc_grid_bn_relu_fwd: out = relu(y scale + shift) on the pixels, zero on the border.   hulc_grid_bn_relu_bwd: dz = BatchNorm backward of
 *   (dout masked by out > 0) on the pixels, zero on the border; dgamma / dbeta (+)=; ws: hulc_grid_bn_bwd_workspace bytes; counters as in
 *   hulc_grid_bn_finalize (or NULL).
 * hulc_grid_upcat_fwd: DecoderBlock's input (unet_decoder.py:60-80): out grid (N, Ho, Wo) rows of Cx + Cs = [nearest-up-sampled x * g | skip];
 *   x (N, Ho/s, Wo/s, Cx) and skip (N, Ho, Wo, Cs) bf16 with element strides (n, y, x) — a grid tensor's pixels or a plain NHWC map; g (N, Cx)
 *   fp32 or NULL = lang_proj(l) of FusionMult (core/fusion.py:64-73).   hulc_grid_upcat_bwd: dsmall (grid (N, Hi, Wi) rows of Cx, pixels only)
 *   = g * the s x s block sums of dX's first Cx channels, dg (N, Cx) (+)= sum over pixels of x * block sums (either may be NULL).
 * hulc_pixel_ce_fwd / _bwd: log-sum-exp over an image's H W logits (logit0: fp32 per grid row) and the labelled pixel's logit
 *   (p0 (N, 2) int32 = row, col); backward writes upstream (softmax - onehot) / (N H W) into channel 0 of a grid tensor of C channels. */
/* hulc_gridconv3x3_fused: the same convolution with the epilogue of a frozen ResNet BasicBlock (torchvision resnet18 as r3m wraps it; BatchNorm
 *   folded into wt and bias by the host): y = [relu](conv(x) + bias[co] [+ add[r][co]]) on the pixels, border rows zero; add = the residual
 *   branch as a grid tensor (rows ldadd apart), bias fp32 [Cout] — each may be NULL.   hulc_grid_from_nhwc: a dense (N, H, W, C) bf16 map onto
 *   the grid (border rows zero). */
int hulc_gridconv3x3_fused(const void* x, long ldx, const void* wt, void* y, long ldy, int N, int H, int W, int Cin, int Cout, const float* bias,
                           const void* add, long ldadd, int relu, void* stream);
int hulc_grid_from_nhwc(const void* x, int N, int H, int W, int C, void* y, long ldy, void* stream);
long hulc_gridconv_stats_bytes(int N, int H, int W, int Cout);
int hulc_gridconv3x3(const void* x, long ldx, const void* wt, void* y, long ldy, int N, int H, int W, int Cin, int Cout, int flip_taps, float* stats,
                     float* out0, const float* bias0, void* stream);
int hulc_grid_bn_finalize(const float* part, int nb, int C, long count, const float* gamma, const float* beta, float eps, float momentum, float* bn,
                          float* run_mean, float* run_var, float* mid, unsigned* counters, void* stream);
int hulc_grid_bn_relu_fwd(const void* y, long ldy, const float* bn, int N, int H, int W, int C, void* out, long ldo, void* stream);
long hulc_grid_bn_bwd_workspace(int N, int H, int W, int C);
int hulc_grid_bn_relu_bwd(const void* dout, long ldd, const void* out, long ldo, const void* y, long ldy, const float* bn, int N, int H, int W, int C,
                          void* dz, long ldz, float* dgamma, float* dbeta, int accumulate_params, void* ws, unsigned* counters, void* stream);
int hulc_grid_upcat_fwd(const void* x, long xsn, long xsy, long xsx, const float* g, const void* skip, long ssn, long ssy, long ssx, int N, int Ho, int Wo,
                        int s, int Cx, int Cs, void* out, void* stream);
int hulc_grid_upcat_bwd(const void* dX, long ldd, const void* x, long xsn, long xsy, long xsx, const float* g, int N, int Hi, int Wi, int s, int Cx,
                        void* dsmall, float* dg, int accumulate_dg, void* stream);
int hulc_pixel_ce_fwd(const float* logit0, const int* p0, int N, int H, int W, float* lse, float* picked, void* stream);
int hulc_pixel_ce_bwd(const float* logit0, const int* p0, const float* lse, const float* upstream, int N, int H, int W, int C, void* dz, void* stream);
/* The one-channel segmentation head (visual_lang_encoders/r3m_rn18.py:64-69: nn.Conv2d(C, 1, 3, padding = 1), C = 8 / 16 / 32 / 64) on the padded
 *   grid, as streaming kernels (a 32-output-channel hulc_gridconv3x3 with 31 zero channels moves the same bytes for 1 / 32 of the work):
 *   hulc_head_conv_fwd: out0[r] = bias[0] + sum_{t, ci} x[r + off_t][ci] w[ci][t] on the pixels, 0 on the border rows (w = the parameter
 *     (1, C, 3, 3) fp32 as it lies, x bf16 grid rows);  hulc_pixel_ce_bwd_rows: g[r] = upstream (softmax - onehot) / (N H W) fp32 per grid row,
 *     zero on the border;  hulc_head_conv_dgrad: dx[r][ci] = sum_t g[r - off_t] w[ci][t] (bf16 grid tensor, border rows zero);
 *   hulc_head_conv_wgrad: dw[ci][t] (+)= sum_r g[r] x[r + off_t][ci], summed in a fixed order through ws (hulc_head_conv_wgrad_workspace bytes);
 *     the bias gradient is sum_r g[r] = 0 for the cross-entropy (softmax - onehot) and is not computed. */
int hulc_head_conv_fwd(const void* x, long ldx, const float* w, const float* bias, int N, int H, int W, int C, float* out0, void* stream);
int hulc_pixel_ce_bwd_rows(const float* logit0, const int* p0, const float* lse, const float* upstream, int N, int H, int W, float* g, void* stream);
int hulc_head_conv_dgrad(const float* g, const float* w, int N, int H, int W, int C, void* dx, long lddx, void* stream);
long hulc_head_conv_wgrad_workspace(int N, int H, int W, int C);
int hulc_head_conv_wgrad(const void* x, long ldx, const float* g, int N, int H, int W, int C, float* dw, int accumulate, void* ws, void* stream);
/* hulc_depth_nll_fwd / _bwd: the tail of DepthEstimationGaussian (hulc2/affordance/models/core/depth_gaussian.py:67-69,94-102) on x (B, D) fp32 =
 *   the output of fc3 + ReLU: mu = x w_mu^T + b_mu, log_sigma = x w_sigma^T + b_sigma, sigma = exp(clamp(log_sigma, -20, 2)),
 *   loss[0] = mean over B of 0.5 (log var + (mu - target)^2 / var) with var = max(sigma, 1e-6) (nn.GaussianNLLLoss fed sigma as the variance).
 *   Backward from gout[0] = d/d loss: dx (B, D) (may be NULL), dw_* (D) and db_* (1) stored, or added when their bit of accumulate_mask is set
 *   (1 dw_mu, 2 db_mu, 4 dw_sigma, 8 db_sigma); the clamps pass the gradient inside their ranges (inclusive), as torch.clamp does. */
int hulc_depth_nll_fwd(const float* x, int B, int D, const float* w_mu, const float* b_mu, const float* w_sigma, const float* b_sigma, const float* target,
                       float* mu, float* sigma, float* log_sigma, float* loss, void* stream);
int hulc_depth_nll_bwd(const float* x, int B, int D, const float* w_mu, const float* w_sigma, const float* mu, const float* sigma, const float* log_sigma,
                       const float* target, const float* gout, float* dx, float* dw_mu, float* db_mu, float* dw_sigma, float* db_sigma, int accumulate_mask,
                       void* stream);

/* ---- recurrent decoder: both RNN layers of one direction as a persistent wavefront kernel ----------------- */
/* nn.RNN(num_layers=2, nonlinearity="relu") of hulc2/models/decoders/logistic_decoder_rnn.py:70-79 and its backward.
 * State rows z_t = [first half | second half] (B x 2H fp32, time-major, consecutive wave steps z_step elements apart,
 * negative for the reversed backward sweep).  Wave step tau = 0..S reads row tau (row 0 must be zero and is not read)
 * and writes row tau+1:
 *     first  half (written for tau < S):  f1( z_tau[:, :H] wA^T + add1[tau] + bias1a + bias1b )
 *     second half (written for tau >= 1, zero at tau = 0):  f2( z_tau[:, :H] wB1^T + z_tau[:, H:] wB2^T + bias2a + bias2b )
 * f = keep where mask[tau] > 0 when a mask is given (backward: stored activations), else ReLU when relu != 0.
 * Weights are bf16, element (n, k) at w[n*ld + k] (t = 0) or w[k*ld + n] (t = 1); H must be 2048, B <= 64.
 * ws: hulc_rnn_wavefront_workspace(S, B, H) bytes; after the launch ws + 256 holds a bf16 mirror of the S+2 state rows (row r of
 * the buffer the sweep walks, in buffer order for both directions) — the operands of the weight-gradient GEMMs at half the bytes.  A device-wide barrier separates wave steps: the stream must not run
 * another kernel concurrently; a barrier timeout writes NaN into the last state row instead of hanging. */
typedef struct hulc_rnn_wave_desc {
    float* z; long z_step;
    const void *wA, *wB1, *wB2; long ldA, ldB1, ldB2; int tA, tB1, tB2;
    const float* add1; long add1_step, ld_add1;
    const float *bias1a, *bias1b, *bias2a, *bias2b;
    const float* mask1; long mask1_step, ld_mask1;
    const float* mask2; long mask2_step, ld_mask2;
    int relu, S, B, H;
    int mirror_t;   /* also write the transposed bf16 mirror (hulc_rnn_wavefront_mirror_t_offset); needs B % 8 == 0 */
    int* err_sticky; /* optional device word: bit 0 (value 1) is OR-ed in on a barrier timeout and never cleared by the kernels: the host polls it
                      * (and hulc_adam_step skips its update while it is set) so a timeout ends the job instead of feeding NaN to Adam */
    const float* add1c; long ld_add1c;   /* optional (B x H, row stride ld_add1c): a per-row term of the first half that is the same at every
                                          * wave step — the plan / goal part of the layer-0 input projection, constant over a sequence */
    int zero_edges;   /* also clear the two fp32 pieces the sweep reads / exposes without writing: row 0 (the zero initial state) and the first
                       * half of row S+1 — the caller then hands in an uninitialised buffer */
} hulc_rnn_wave_desc;
long hulc_rnn_wavefront_workspace(int S, int B, int H);
long hulc_rnn_wavefront_mirror_offset(void);   /* byte offset of the bf16 state mirror (S+2, B, 2H) inside ws */
/* byte offset of the TRANSPOSED bf16 mirror (2H, (S+2)*B), token = row_of_the_sweep * B + batch row — the k-major operand of the
 * weight-gradient GEMMs dW = delta^T h (k = tokens contiguous); 0 when B % 8 != 0 (not produced) */
long hulc_rnn_wavefront_mirror_t_offset(int S, int B, int H);
int hulc_rnn_wavefront(const hulc_rnn_wave_desc* d, void* ws, void* stream);

/* ---- optimizer ------------------------------------------------------------------------------- */
/* torch.optim.Adam semantics (hulc2.py:185-198, conf/model/optimizer/adam.yaml) over a flat fp32 arena;
 * bf16_shadow (optional) receives the updated weights rounded to bf16 for the MFMA kernels. */
int hulc_adam_step(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                   float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                   void* stream);   /* skip_flag (optional device word): non-zero = leave parameters and moments untouched (kernel fault upstream) */
/* (ABI 4) The same step, additionally writing the ROUNDING REMAINDERS lo = bf16(w - float(bf16(w))) of the updated weights into a second
 * shadow arena (same element offsets as bf16_shadow) inside up to 8 element ranges lo_ranges[2 i] <= k < lo_ranges[2 i + 1] (HOST array,
 * starts multiples of 4): the second halves of the split operands of the fp32-class forwards (conv1, transformer trunk, camera heads,
 * language goal encoder).  Replaces the separate hulc_residual_bf16 launch behind the optimizer. */
int hulc_adam_step_lo(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                      float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                      void* lo_shadow, const long* lo_ranges, int n_ranges, void* stream);
/* (ABI 5) The same step under torch.amp.GradScaler WITHOUT a host synchronisation (reference: conf/trainer/play_trainer.yaml:3 `precision: 16`
 * -> Lightning's scaler.step(optimizer), hulc2/training.py:79-82; an optimizer that sets `_step_supports_amp_scaling` is handed the scaler's
 * device scalars instead of a `found_inf.item()`): loss_scale (optional device float S: the gradients are multiplied by grad_scale / S, the
 * reciprocal formed in double as torch's unscale_ forms it) and found_inf (optional device float: non-zero = the step is skipped, parameters,
 * moments and shadows untouched).  hulc_step_count_advance_if bumps state[1] unless found_inf is set (a skipped step does not count). */
int hulc_adam_step_amp(float* p, const float* g, float* m, float* v, void* bf16_shadow, long n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, const unsigned long long* step_state, float grad_scale, const int* skip_flag,
                       void* lo_shadow, const long* lo_ranges, int n_ranges, const float* loss_scale, const float* found_inf, void* stream);
int hulc_step_count_advance_if(unsigned long long* state, const float* found_inf, void* stream);
/* Device-resident step state {rng word, optimizer step count}: advanced by one kernel per training step so that
 * a captured hipGraph replays with fresh dropout masks / plan samples and the right Adam bias correction.
 * RNG kernels xor state[0] into their site seed (seed_dev = state); hulc_adam_step reads state[1] when
 * step_state != NULL (the `step` argument is then ignored). */
int hulc_step_state_advance(unsigned long long* state, void* stream);
/* the same with the two words advanced separately: Hulc2.training_step (hulc2.py:336) walks the RNG word whenever it runs in training
 * mode (under Lightning too), the native trainer alone bumps the optimizer step count (its Adam bias correction) */
int hulc_step_state_advance_words(unsigned long long* state, int rng, int step, void* stream);
int hulc_cast_f32_to_bf16(const float* src, void* dst, long n, void* stream);
/* Gradient all-reduce helpers of the data-parallel path (replaces the NCCL ring inside Lightning's DDPStrategy, hulc2/training.py:72-75;
 * SURVEY §8e): bf16 payloads widen back to the fp32 arena; the direct algorithm (all-to-all -> local sum -> all-gather, every xGMI link
 * busy at once) sums the W rank contributions of this rank's chunk in rank order with fp32 accumulation. */
int hulc_cast_bf16_to_f32(const void* src, float* dst, long n, void* stream);
int hulc_sum_chunks(const void* src, int dtype, int W, long chunk, void* dst, void* stream);
/* Transposed bf16 shadows of the 2-D weights of the arena in one launch: tiles[q] = {element offset, rows, cols, tile row,
 * tile col} over 64 x 64 tiles; dst + offset receives W^T ([cols][rows]).  The data-gradient GEMMs (dX = dY W of every
 * nn.Linear) then read W k-major like the forward pass does. */
int hulc_transpose_bf16_tiles(const void* src, void* dst, const long* tiles, int ntiles, void* stream);
/* (ABI 3) Fragment-packed copies of the transformer feed-forward weights for hulc_txl_block_*: out[i] (FF * 128 entries) = the element of
 * the source matrix that element i of the packed array holds.  layout 0: source W1 [FF][128] (forward / backward A operand of z^T),
 * 1: source W2 [128][FF] (forward), 2: source W2^T [FF][128], 3: source W1^T [128][FF] (backward); element order in csrc/optim.hip.
 * Runs on the host (no device work); every permutation moves runs of 4 consecutive source elements. */
int hulc_ffn_frag_perm(int layout, int FF, int* out);
/* dst chunk c (8 bytes = 4 bf16) = chunk (idx[c] & 0x7fffffff) of src1 when bit 31 of idx[c] is set, else of src0: every packed weight copy
 * of a step from the bf16 arena (src0) and its transposed shadow (src1) in one launch. */
int hulc_gather_chunks(const void* src0, const void* src1, void* dst, const unsigned* idx, long nchunks, void* stream);
/* (ABI 3) lo[dst + i] = bf16(p32[src + i] - float(hi[src + i])) for nseg segments {src offset, count, dst offset} (device table of longs): the
 * rounding remainders of the bf16 weight shadows, the second halves of the split operands of hulc_txl_block_fwd's fp32-class forward. */
int hulc_residual_bf16(const float* p32, const void* hi, void* lo, const long* segments, int nseg, void* stream);
/* All conv-weight repacks of a step in one launch: table[q] = {src offset (fp32 arena elements), dst offset (bf16 elements), Cout, Cin,
 * KH, KW, mode}; mode 0 = OIHW flat (conv1 forward), 1 = OHWI (NHWC forward, k = (kh,kw,c)), 2 = IHWO (data gradient,
 * rows = input channel, k = (kh,kw,cout)), 3 = [tap][c][o] (the transposed flatten-linear operand); mode + 8 (ABI 5): the rounding remainder
 * bf16(w - float(bf16(w))) in the same layout — the second operand of a two-product (fp32-class) forward on bf16 activations.  Replaces a
 * permute copy + cast per layer and layout. */
int hulc_repack_conv_weights(const float* src, void* dst, const long* table, int n, void* stream);
/* (ABI 4) hulc_transpose_bf16_tiles and hulc_repack_conv_weights of a step as ONE launch (independent jobs on disjoint workgroup ranges);
 * either half may be empty (count 0). */
int hulc_derive_copies(const void* bf16, void* bf16_t, const long* tiles, int ntiles, const float* p32, void* conv_dst, const long* conv_table,
                       int nconv, void* stream);
/* (ABI 4) two chunk gathers as ONE launch: the first as hulc_gather_chunks (sources a0 / a1), the second from the single source b0. */
int hulc_gather_chunks2(const void* a0, const void* a1, void* ad, const unsigned* ai, long an, const void* b0, void* bd, const unsigned* bi,
                        long bn, void* stream);

#ifdef __cplusplus
}
#endif
#endif
