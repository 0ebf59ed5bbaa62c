// txl_fused.hip — the attention half of the plan-recognition transformer layer as ONE launch per direction (bf16 MFMA).
//
// reference arithmetic: nn.TransformerEncoderLayer (post-norm, d_model 128, 8 heads of 16, dropout p) as configured by
// hulc2/models/plan_encoders/plan_recognition_net.py:108-121 —
//     y1 = LayerNorm1( x + dropout( out_proj( softmax_dropout( (x Wq^T + bq) / 4 · (x Wk^T + bk)^T ) (x Wv^T + bv) ) ) )
// and its autograd backward.  Sequences are short (S <= 32) and independent: one workgroup (4 waves) owns one sequence, wave w owns
// heads 2w and 2w+1, i.e. the 32 q-, k- and v-columns 32w..32w+31.  Nothing but x, the weights and the results touches memory:
// the 32 x 32 score tile of a head is ONE v_mfma_f32_32x32x16_bf16 (k = head dimension 16), softmax runs in registers, P V is two more.
//
// The trick that keeps everything in registers: an MFMA accumulator tile D (lane <-> column, 16 registers <-> rows
// (e&3) + 8(e>>2) + 4(lane>>5)) IS an MFMA operand fragment for a product that reduces over D's rows — registers 0..7 are the 8 k-slots
// of k-step 0 (rows 0-3, 8-11 | 4-7, 12-15 for the two lane halves), registers 8..15 those of k-step 1 — as long as the OTHER operand
// enumerates the reduction index in the same order.  Both operands of QK^T come out of transposed projection tiles
// (D[qcol][token] = Wq x^T), P^T feeds P V directly, ctx^T feeds the output projection (Wo read with the matching column order), and
// in the backward pass every tile is produced in the orientation its consumer wants by swapping the MFMA operands (a second MFMA is
// cheaper than any transpose).  Exchanges between waves (ctx, d_o, dqkv: each wave owns 32 of the 128 / 384 columns of the next
// reduction) are 16-byte fragment copies through LDS.
//
// Dropout draws the same counter-RNG streams as the unfused kernels (attention probabilities: ((b H + h) S + i) S + j; residual
// branch: token * 128 + feature), so fused and unfused layers give the same masks.
#include "hulc_common.h"
#include "hulc_abi_internal.h"

#include "txl_attn.h"

namespace {

__global__ __launch_bounds__(256) void txl_attn_fwd_kernel(TxlP p) {
    __shared__ __attribute__((aligned(16))) char lds[TXL_FWD_LDS];
    txl_attn_fwd_body(p, blockIdx.x, lds);
}

__global__ __launch_bounds__(256) void txl_attn_bwd_kernel(TxlP p) {
    __shared__ __attribute__((aligned(16))) char lds[TXL_BWD_LDS];
    txl_attn_bwd_body(p, blockIdx.x, lds);
}

int txl_check(const hulc_txl_attn_desc* d, const char* who) {
    if (!d || !d->x || !d->Wqkv || !d->bqkv || !d->gamma) return hulc_fail(-1, who);
    if (d->E != E || d->H != NH || d->S < 1 || d->S > SMAX || d->B < 1) return hulc_fail(-2, who);
    return 0;
}

TxlP txl_params(const hulc_txl_attn_desc* d) {
    TxlP p = {};
    p.x = d->x; p.Wqkv = (const uint16_t*)d->Wqkv; p.Wo = (const uint16_t*)d->Wo; p.WqkvT = (const uint16_t*)d->WqkvT; p.WoT = (const uint16_t*)d->WoT;
    p.bqkv = d->bqkv; p.bo = d->bo; p.gamma = d->gamma; p.beta = d->beta; p.eps = d->eps; p.B = d->B; p.S = d->S;
    p.drop_p = d->drop_p; p.seed_attn = d->seed_attn; p.seed_ln = d->seed_ln; p.seed_dev = d->seed_dev;
    p.y = d->y; p.pre = d->pre; p.mean = d->mean; p.rstd = d->rstd; p.ctx = (uint16_t*)d->ctx;
    p.dy = d->dy; p.dy_slab = d->dy_slab; p.n_slab = d->dy_slab ? d->n_slab : 0; p.slab_stride = d->slab_stride;
    p.dx = d->dx; p.d_o = (uint16_t*)d->d_o; p.dqkv = (uint16_t*)d->dqkv; p.ln_partial = d->ln_partial;
    return p;
}

}  // namespace

// see include/hulc2_amd.h
extern "C" int hulc_txl_attn_fwd(const hulc_txl_attn_desc* d, void* stream) {
    if (int rc = txl_check(d, "hulc_txl_attn_fwd: needs d_model 128, 8 heads, 1 <= S <= 32 and non-null operands")) return rc;
    if (!d->Wo || !d->bo || !d->beta || !d->y) return hulc_fail(-1, "hulc_txl_attn_fwd: null pointer");
    if ((d->pre != nullptr) != (d->mean != nullptr) || (d->mean != nullptr) != (d->rstd != nullptr))
        return hulc_fail(-3, "hulc_txl_attn_fwd: pre / mean / rstd are saved together or not at all");
    txl_attn_fwd_kernel<<<(unsigned)d->B, 256, 0, (hipStream_t)stream>>>(txl_params(d));
    return hulc_check_launch("hulc_txl_attn_fwd");
}

extern "C" int hulc_txl_attn_bwd(const hulc_txl_attn_desc* d, void* stream) {
    if (int rc = txl_check(d, "hulc_txl_attn_bwd: needs d_model 128, 8 heads, 1 <= S <= 32 and non-null operands")) return rc;
    if (!d->WqkvT || !d->WoT || !d->dy || !d->pre || !d->mean || !d->rstd || !d->dx || !d->d_o || !d->dqkv || !d->ln_partial)
        return hulc_fail(-1, "hulc_txl_attn_bwd: null pointer");
    txl_attn_bwd_kernel<<<(unsigned)d->B, 256, 0, (hipStream_t)stream>>>(txl_params(d));
    return hulc_check_launch("hulc_txl_attn_bwd");
}
