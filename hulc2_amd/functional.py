"""Differentiable operators of the HULC++ low-level policy, each a torch.autograd.Function whose forward
and backward are calls into libhulc2_amd.so (hulc2_amd/kernels.py).  Python only allocates buffers and
orders launches; there is no eager-PyTorch arithmetic fallback.

Layout conventions: camera frames arrive NCHW fp32 as the reference delivers them, every
conv activation is NHWC; token tensors are (B, S, D) row-major with token index b*S+s; weights keep the
reference's nn.Linear / nn.Conv2d / nn.RNN layouts so state_dicts interchange.
"""
from typing import List, Optional, Sequence, Tuple

import os

import torch

from . import gradsink
from . import kernels as kn
from .shadow import weight_operand


def _scoped(cls):
    """class decorator: the Function's backward runs in the compute mode its forward ran in (kernels.compute_scope)"""
    fwd, bwd = cls.forward, cls.backward

    def forward(ctx, *args):
        ctx._compute = kn.backward_compute()       # (differs from the forward's mode inside a forward-only precision scope: 'mixed' mode)
        return fwd(ctx, *args)

    def backward(ctx, *grads):
        if ctx._compute == kn.get_compute():
            return bwd(ctx, *grads)
        with kn.compute_scope(ctx._compute):
            return bwd(ctx, *grads)

    forward.__doc__, backward.__doc__ = fwd.__doc__, bwd.__doc__
    cls.forward, cls.backward = staticmethod(forward), staticmethod(backward)
    return cls


def _f32(*shape, like: torch.Tensor) -> torch.Tensor:
    return torch.empty(*shape, dtype=torch.float32, device=like.device)


def _c(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_contiguous() else t.contiguous()


def _act_dtype() -> torch.dtype:
    """Storage type of the conv-stack activations and their gradients: the MFMA kernels round operands to the
    compute type while staging anyway, so in bf16 mode keeping them bf16 in HBM halves the traffic at identical
    arithmetic (fp32 mode keeps fp32)."""
    return torch.bfloat16 if kn.get_compute() == "bf16" else torch.float32


# ------------------------------------------------------------------------------------------------
# MLP chain: y = L_n(...relu(L_1(x))) — all Linear(+ReLU(+Dropout)) stacks of the policy
# reference: plan_proposal_net.py:26-47, goal_encoders.py:21-34,53-71, vision_network.py:49-52,
#            proj_vis_lang.py:10-21, nn.TransformerEncoderLayer feed-forward block
# ------------------------------------------------------------------------------------------------
@_scoped
class MLPFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, relus: Tuple[bool, ...], drops: Tuple[float, ...], seed: int, *params):
        x3 = len(relus) > 0 and relus[-1] == "x3"          # caller's hint (mlp(..., x3=True)): split-operand chain inside an exact-forward scope
        relus = tuple(relus[:-1]) if x3 else tuple(relus)
        L = len(relus)
        assert len(params) == 2 * L and not relus[-1], "chain must end with a plain Linear"
        # a 2-D input with unit inner stride is consumed in place (row stride = lda): the recurrent decoder hands in a column slice
        x2 = x if (x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0) else _c(x.reshape(-1, x.shape[-1]))
        M = x2.shape[0]
        acts: List[torch.Tensor] = []
        widths = [params[2 * i].shape[0] for i in range(L)]
        pre_ok = L >= 2 and not any(d > 0 for d in drops) and x2.dtype == torch.float32 and x2.stride(0) % 4 == 0 and x2.data_ptr() % 16 == 0
        chain = pre_ok and kn.mlp_chain_ok(M, x2.shape[1], widths, x2.device)
        if (not chain) and x3 and pre_ok and M <= 32 and max(widths[:-1] + [x2.shape[1]]) <= 2048 and kn.exact_site_in_bf16_step():
            # exact-forward scope of a bf16 step: the chain launch with split operands (three MFMAs per product, fp32-class values) instead of
            # per-layer exact-fp32 GEMMs — the arithmetic of the paired launch's second stack (DualMLPFn), bit for bit
            with kn.compute_scope("bf16", fwd_only=True):
                if kn.mlp_chain_ok(M, x2.shape[1], widths, x2.device):
                    acts = [_f32(M, n, like=x2) for n in widths]
                    kn.mlp_chain(x2, [(weight_operand(params[2 * i]), params[2 * i + 1], relus[i], None, 1.0, acts[i], weight_operand(params[2 * i], "lo"))
                                      for i in range(L)], M)
                    chain = True
        elif chain:    # the whole stack as one persistent launch (csrc/mlp_chain.hip): no per-layer launch, no split-K epilogues
            acts = [_f32(M, n, like=x2) for n in widths]
            kn.mlp_chain(x2, [(weight_operand(params[2 * i]), params[2 * i + 1], relus[i], None, 1.0, acts[i]) for i in range(L)], M)
        inp = x2
        for i in range(L if not chain else 0):
            W, b = params[2 * i], params[2 * i + 1]
            N, K = W.shape
            out = _f32(M, N, like=x2)
            kn.gemm(inp, weight_operand(W), out, M, N, K, inp.stride(0), K, N, bias=b, relu=relus[i],
                    drop_p=drops[i], drop_seed=seed + i)
            acts.append(out)
            inp = out
        ctx.save_for_backward(x2, *acts[:-1], *params)
        ctx.meta = (relus, drops, seed, L, x.shape)
        ctx.chain = chain
        ctx.share = kn.coop_share()                        # (a forked branch of the step: the backward's chain launch keeps to the same share of the device)
        return acts[-1].reshape(*x.shape[:-1], acts[-1].shape[-1])

    @staticmethod
    def backward(ctx, dy):
        relus, drops, seed, L, xshape = ctx.meta
        saved = ctx.saved_tensors
        x2, acts, params = saved[0], saved[1:L], saved[L:]
        g = _c(dy.reshape(-1, dy.shape[-1]))
        need_x = ctx.needs_input_grad[0]
        # (the data-gradient chain does not depend on how the forward ran: an exact-fp32 forward — selective precision, mixed mode — leaves the
        # same fp32 activations behind, and the backward products run in bf16 either way)
        with kn.coop_share_scope(ctx.share):
            dx, grads = _mlp_backward(x2, acts, params, relus, drops, g, need_x, ctx.chain or not any(d > 0 for d in drops))
        return (dx.reshape(xshape) if need_x else None, None, None, None, *grads)


def _dgrad_chain_layers(acts, params, relus, drops, g, lo_layer, ks):
    """the data-gradient chain  g_{i-1} = (g_i W_i) * (act_{i-1} > 0)  of layers L-1 .. lo_layer as mlp_chain layer tuples; -> (layers, {i-1: out})"""
    L = len(relus)
    layers, gs = [], {}
    for i in range(L - 1, lo_layer - 1, -1):
        out = _f32(g.shape[0], ks[i], like=g)
        masked = i > 0 and relus[i - 1]
        layers.append((weight_operand(params[2 * i], "t"), None, False, acts[i - 1] if masked else None,
                       1.0 / (1.0 - drops[i - 1]) if masked else 1.0, out))
        gs[i - 1] = out
    return layers, gs


def _mlp_backward(x2, acts, params, relus, drops, g, need_x, chain, gs=None):
    """backward of a Linear(+ReLU(+Dropout)) stack: -> (dx or None, [dW0, db0, ...] with None where the gradient went into the arena).
    gs: data gradients {layer index: tensor} a caller already produced (DualMLPFn's paired chain launch)"""
    L = len(relus)
    M = g.shape[0]
    grads = [None] * (2 * L)
    lo_layer = 0 if need_x else 1                       # lowest layer whose input gradient is wanted
    n_chain = L - lo_layer
    ks = [x2.shape[1]] + [params[2 * i].shape[0] for i in range(L - 1)]      # input width of layer i
    if gs is None:
        gs = {}
        # the data-gradient chain as one persistent launch when the forward took that path
        if (chain and n_chain >= 2 and g.stride(0) % 4 == 0 and g.data_ptr() % 16 == 0
                and kn.mlp_chain_ok(M, g.shape[1], [ks[i] for i in range(L - 1, lo_layer - 1, -1)], g.device)):
            layers, gs = _dgrad_chain_layers(acts, params, relus, drops, g, lo_layer, ks)
            kn.mlp_chain(g, layers, M)
    gs[L - 1] = g
    for i in range(L - 1, -1, -1):
        g = gs[i] if i in gs else g
        W = params[2 * i]
        N, K = W.shape
        inp = x2 if i == 0 else acts[i - 1]
        sw, sb = gradsink.get(W), gradsink.get(params[2 * i + 1])
        dW = sw if sw is not None else _f32(N, K, like=g)
        db = sb if sb is not None else _f32(N, like=g)
        acc_w = sw is not None and not gradsink.first_write(W)
        acc_b = sb is not None and not gradsink.first_write(params[2 * i + 1])
        # dW (+)= g^T inp, db (+)= column sums of g: part of the pass's grouped weight-gradient launch when both land in the arena
        kn.wgrad(g, inp, dW, N, K, M, g.stride(0), inp.stride(0), K, accumulate=acc_w, rowsum=db, rowsum_accumulate=acc_b,
                 defer=sw is not None and sb is not None)
        grads[2 * i] = None if sw is not None else dW          # written straight into the gradient arena
        grads[2 * i + 1] = None if sb is not None else db
        if (i - 1) in gs:
            continue                                       # produced by the chain launch
        if i > 0 or need_x:
            dinp = _f32(M, K, like=g)
            wt = weight_operand(W, "t")                      # (K, N): the reduction index contiguous, like the forward pass
            if i > 0 and relus[i - 1]:
                kn.gemm(g, wt, dinp, M, K, N, g.stride(0), N, K, mask=acts[i - 1], ld_mask=K, mask_scale=1.0 / (1.0 - drops[i - 1]))
            else:
                kn.gemm(g, wt, dinp, M, K, N, g.stride(0), N, K)
            gs[i - 1] = dinp
    return (gs.get(-1) if need_x else None), grads


@_scoped
class DualMLPFn(torch.autograd.Function):
    """Two independent Linear(+ReLU) stacks of the same hidden widths on <= 32 rows each — VisualGoalEncoder.mlp and LanguageGoalEncoder.mlp
    (reference goal_encoders.py:21-34 / :53-71) — as ONE persistent launch forward (hulc_mlp_chain2) and one for the two data-gradient
    chains backward; the weight gradients join the pass's grouped launch as for MLPFn.  params: the first stack's (W, b) pairs, then the
    second's.  Only built by dual_mlp() after kernels.mlp_chain2_ok."""

    @staticmethod
    def forward(ctx, xa, xb, relus, *params):
        """relus: tuple of flags; a trailing string "exact_b" in it selects split operands (fp32-class forward) for the SECOND stack"""
        exact_b = len(relus) > 0 and relus[-1] == "exact_b"
        relus = tuple(relus[:-1]) if exact_b else tuple(relus)
        L = len(relus)
        pa, pb = params[:2 * L], params[2 * L:]
        outs = []
        for x, ps in ((xa, pa), (xb, pb)):
            outs.append([_f32(x.shape[0], ps[2 * i].shape[0], like=x) for i in range(L)])
        lo = [(weight_operand(pb[2 * i], "lo"),) if exact_b else () for i in range(L)]
        kn.mlp_chain2(xa, [(weight_operand(pa[2 * i]), pa[2 * i + 1], relus[i], None, 1.0, outs[0][i]) for i in range(L)], xa.shape[0],
                      xb, [(weight_operand(pb[2 * i]), pb[2 * i + 1], relus[i], None, 1.0, outs[1][i]) + lo[i] for i in range(L)], xb.shape[0])
        ctx.save_for_backward(xa, xb, *outs[0][:-1], *outs[1][:-1], *params)
        ctx.relus = relus
        ctx.share = kn.coop_share()
        return outs[0][-1], outs[1][-1]

    @staticmethod
    def backward(ctx, dya, dyb):
        with kn.coop_share_scope(ctx.share):
            return DualMLPFn._backward(ctx, dya, dyb)

    @staticmethod
    def _backward(ctx, dya, dyb):
        relus = ctx.relus
        L = len(relus)
        saved = ctx.saved_tensors
        xs = saved[:2]
        acts = (saved[2:L + 1], saved[L + 1:2 * L])
        params = (saved[2 * L:4 * L], saved[4 * L:])
        drops = (0.0,) * L
        need = (ctx.needs_input_grad[0], ctx.needs_input_grad[1])
        g = [_c(dya), _c(dyb)]
        order = (0, 1) if (need[0] or not need[1]) else (1, 0)        # the chain that goes down to its input is the deeper one: it leads
        descr = []
        for s in order:
            lo = 0 if need[s] else 1
            ks = [xs[s].shape[1]] + [params[s][2 * i].shape[0] for i in range(L - 1)]
            descr.append((lo, ks))
        widths = [[ks[i] for i in range(L - 1, lo - 1, -1)] for lo, ks in descr]
        gs = [None, None]
        if (all(len(w) >= 1 for w in widths) and all(t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0 for t in g)
                and kn.mlp_chain2_ok(g[order[0]].shape[0], g[order[0]].shape[1], widths[0], g[order[1]].shape[0], g[order[1]].shape[1], widths[1],
                                     g[0].device)):
            built = [_dgrad_chain_layers(acts[s], params[s], relus, drops, g[s], descr[j][0], descr[j][1]) for j, s in enumerate(order)]
            kn.mlp_chain2(g[order[0]], built[0][0], g[order[0]].shape[0], g[order[1]], built[1][0], g[order[1]].shape[0])
            gs[order[0]], gs[order[1]] = built[0][1], built[1][1]
        res = [_mlp_backward(xs[s], acts[s], params[s], relus, drops, g[s], need[s], True, gs=gs[s]) for s in (0, 1)]
        return (res[0][0], res[1][0], None, *res[0][1], *res[1][1])


@_scoped
class FlattenLinearFn(torch.autograd.Function):
    """relu(Linear(nn.Flatten(x_nchw))) computed on the NHWC activation a (N, H, W, C) as it lies in memory: the reference flattens
    (C, H, W) (vision_network_gripper.py:16-17), so the weight's columns are reordered to (h, w, c) once per optimizer step
    (weight_operand(W, "hwc")) instead of permuting 2048 x 3136 activations forward and their gradients backward.  The incoming ReLU of
    `a` (conv3's) is applied to the returned gradient in the data-gradient GEMM's epilogue: da is already masked by (a > 0)."""

    @staticmethod
    def forward(ctx, a, W, b, a_exact=None):
        N = a.shape[0]
        K = a[0].numel()
        O = W.shape[0]
        ctx.a_dtype = a.dtype
        chw = (a.shape[3], a.shape[1], a.shape[2])
        if kn.exact_site_in_bf16_step() and a.dtype == torch.bfloat16 and os.environ.get("HULC_FLATLIN_X2"):
            # OPT-IN (HULC_FLATLIN_X2=1; round 5): the exact-forward site "encfc" behind a bf16 conv stack as two bf16 MFMA GEMMs, a w_hi + a w_lo
            # (the activation IS bf16: nothing to split), on the trainer's layout shadows — instead of a cast copy of the activation, a per-step
            # fp32 relayout of the weight and an fp32-MFMA GEMM (38 + 9 + 5 us per step).  Off by default: the 2^-17 weight remainder it drops
            # moved one recorded gradient error of the B = 2 parity case past its 1.5 x ratchet (tests/golden/error_budget.json), and the
            # recorded errors are not re-recorded for a 0.8 % step-time gain
            x2 = _c(a).reshape(N, K)
            out = _f32(N, O, like=x2)
            with kn.compute_scope("bf16", fwd_only=True):
                part = _f32(N, O, like=x2)
                kn.gemm(x2, weight_operand(W, "hwc_lo", chw=chw), part, N, O, K, K, K, O)
                kn.gemm(x2, weight_operand(W, "hwc", chw=chw), out, N, O, K, K, K, O, bias=b, add=part, ld_add=O, relu=True)
            ctx.save_for_backward(x2, out, W, b)
            ctx.ashape = a.shape
            return out
        if kn.get_compute() != "bf16" and a.dtype == torch.bfloat16:
            # an exact-fp32 site behind a bf16 conv stack (HULC_FP32_SITES "encfc"): the fp32 GEMM takes fp32 operands — the stack's own
            # exact map when it kept one (site "a3"), else a cast copy of the bf16 map
            if a_exact is not None and a_exact.dtype == torch.float32:
                # forward on the exact map; the backward's operand stays the bf16 map (what it was when a cast copy of it was saved: the
                # weight-gradient and masked data-gradient products round their operands to bf16 anyway, and every path rounds a value that
                # is already bf16 the same way)
                x2 = _c(a_exact).reshape(N, K)
                out = _f32(N, O, like=x2)
                kn.gemm(x2, weight_operand(W, "hwc", chw=chw), out, N, O, K, K, K, O, bias=b, relu=True)
                ctx.save_for_backward(_c(a).reshape(N, K), out, W, b)
                ctx.ashape = a.shape
                return out
            else:
                a32 = torch.empty(a.shape, dtype=torch.float32, device=a.device)
                kn.cast_bf16_to_f32(_c(a), a32, a.numel())
                a = a32
        x2 = _c(a).reshape(N, K)
        out = _f32(N, O, like=x2)
        kn.gemm(x2, weight_operand(W, "hwc", chw=chw), out, N, O, K, K, K, O, bias=b, relu=True)
        ctx.save_for_backward(x2, out, W, b)
        ctx.ashape = a.shape
        return out

    @staticmethod
    def backward(ctx, dy):
        x2, out, W, b = ctx.saved_tensors
        N, K = x2.shape
        O = W.shape[0]
        C = ctx.ashape[3]
        g = _f32(N, O, like=out)
        kn.relu_bwd(_c(dy), out, g, g.numel())
        sw, sb = gradsink.get(W), gradsink.get(b)
        if sw is not None and sb is not None and kn.wgrad_group_ok(g, x2, sw, O, K, N, O, K, K):
            # straight into the arena with the parameter's (c, h*w) column order: one item of the pass's grouped weight-gradient launch
            kn.wgrad(g, x2, sw.view(O, K), O, K, N, O, K, K, accumulate=not gradsink.first_write(W), rowsum=sb,
                     rowsum_accumulate=not gradsink.first_write(b), defer=True, col_perm=C)
            dW = db = None
        else:
            dWh = _f32(O, K, like=g)                                   # (h, w, c) column order
            db = sb if sb is not None else _f32(O, like=g)
            acc_b = sb is not None and not gradsink.first_write(b)
            fused = kn.gemm_fuses_rowsum(O, False)
            kn.gemm(g, x2, dWh, O, K, N, O, K, K, a_kmajor=False, b_kmajor=False, rowsum=db if fused else None, rowsum_accumulate=acc_b)
            if not fused:
                kn.colsum(g, N, O, O, db, accumulate=acc_b)
            src = dWh.view(O, K // C, C).transpose(1, 2)                # -> the parameter's (c, h*w) order
            if sw is not None:
                dst = sw.view(O, C, K // C)
                if gradsink.first_write(W):
                    dst.copy_(src)
                else:
                    dst.add_(src)
                dW = None
            else:
                dW = src.reshape(O, K)
            if sb is not None:
                db = None
        da = None
        if ctx.needs_input_grad[0]:
            # in the activation's own storage type: autograd would otherwise cast the 2048 x 3136 gradient in a launch of its own
            da = torch.empty(N, K, dtype=ctx.a_dtype, device=g.device)
            kn.gemm(g, weight_operand(W, "hwc_t", chw=(C, ctx.ashape[1], ctx.ashape[2])), da, N, K, O, O, O, K, mask=x2, ld_mask=K, mask_scale=1.0)   # x (a > 0): conv3's ReLU
            da = da.view(ctx.ashape)
        return da, dW, db, None


def flatten_linear_relu(a_nhwc, W, b):
    return FlattenLinearFn.apply(a_nhwc, W, b, exact_map(a_nhwc))


def mlp(x, layers: Sequence[Tuple[torch.Tensor, torch.Tensor, bool]], drops: Optional[Sequence[float]] = None, seed: int = 0, x3: bool = False):
    """layers: [(weight, bias, relu), ...]; dropout (inverted, after the ReLU) per layer optional.  x3: inside an exact-forward scope of a
    bf16 step, prefer the split-operand chain launch to the exact-fp32 GEMMs (<= 32 rows)."""
    relus = tuple(bool(r) for _, _, r in layers) + (("x3",) if x3 else ())
    drops = tuple(float(d) for d in (drops or [0.0] * len(layers)))
    params = [t for W, b, _ in layers for t in (W, b)]
    return MLPFn.apply(x, relus, drops, int(seed), *params)


@_scoped
class Mlp2RowsFn(torch.autograd.Function):
    """fc2(relu(fc1(x))) over many rows — the head of both camera encoders (vision_network.py:49-52,60-66) — as one launch per direction
    (csrc/mlp2_rows.hip): the (rows x 512) hidden activation stays in registers, backward recomputes it and leaves the bf16 operands of the
    two weight-gradient products for the pass's grouped launch."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, x3: bool = False):
        y = _f32(x.shape[0], W2.shape[0], like=x)
        with kn.compute_scope("bf16", fwd_only=True):       # (x3: called inside an exact-forward scope — the operands are the bf16 shadows + remainders)
            lo = (weight_operand(W1, "lo"), weight_operand(W2, "lo")) if x3 else (None, None)
            kn.mlp2_rows_fwd(x, weight_operand(W1), b1, weight_operand(W2), b2, y, *lo)
        ctx.save_for_backward(x, W1, b1, W2, b2)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, W1, b1, W2, b2 = ctx.saved_tensors
        T, H, OUT = x.shape[0], W1.shape[0], W2.shape[0]
        dy = _c(dy)
        if dy.dtype != torch.float32:
            dy = dy.float()
        dx = _f32(T, 128, like=dy) if ctx.needs_input_grad[0] else None
        h = torch.empty(T, H, dtype=torch.bfloat16, device=dy.device)
        dh = torch.empty(T, H, dtype=torch.bfloat16, device=dy.device)
        kn.mlp2_rows_bwd(x, dy, weight_operand(W1), b1, weight_operand(W1, "t"), weight_operand(W2, "t"), dx, h, dh)
        ret = []
        for left, right, W, b, M, N in ((dh, x, W1, b1, H, 128), (dy, h, W2, b2, OUT, H)):
            (dW, a1, rW), (db, a2, rb) = _sink_or_new(W, (M, N), dy), _sink_or_new(b, (M,), dy)
            kn.wgrad(left, right, dW, M, N, T, M, N, N, accumulate=a1, rowsum=db, rowsum_accumulate=a2, defer=rW is None and rb is None)
            ret += [rW, rb]
        return (dx, *ret, None)


def mlp2_rows(x, fc1_w, fc1_b, fc2_w, fc2_b):
    """fc2(relu(fc1(x))) for x (rows, 128): one launch per direction where hulc_mlp2_rows_* take the shape, else the GEMM chain.  Inside an
    exact-forward scope of a bf16 step (site "encfc") the launch forms its products from split operands instead of leaving for fp32 GEMMs."""
    if kn.mlp2_rows_ok(x, fc1_w, fc2_w):
        return Mlp2RowsFn.apply(x, fc1_w, fc1_b, fc2_w, fc2_b, kn.exact_site_in_bf16_step())
    return mlp(x, [(fc1_w, fc1_b, True), (fc2_w, fc2_b, False)])


def dual_mlp(xa, layers_a, xb, layers_b, exact_b: bool = False):
    """mlp(xa, layers_a), mlp(xb, layers_b) — as one launch where the pair fits hulc_mlp_chain2 (both <= 32 rows, the same ReLU pattern
    and hidden widths), else one after the other.  exact_b: the second stack's forward from split operands (fp32-class values, three MFMAs
    per product; precision site "goal") — only with the paired launch (the caller falls back to an exact-fp32 scope otherwise: None)."""
    relus = tuple(bool(r) for _, _, r in layers_a)
    wa, wb = [W.shape[0] for W, _, _ in layers_a], [W.shape[0] for W, _, _ in layers_b]
    fits = (relus == tuple(bool(r) for _, _, r in layers_b) and not relus[-1] and len(relus) >= 2 and xa.dim() == 2 and xb.dim() == 2
            and all(x.is_cuda and x.dtype == torch.float32 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0 for x in (xa, xb))
            and kn.mlp_chain2_ok(xa.shape[0], xa.shape[1], wa, xb.shape[0], xb.shape[1], wb, xa.device))
    if not fits:
        return None if exact_b else (mlp(xa, layers_a), mlp(xb, layers_b))
    params = [t for W, b, _ in layers_a for t in (W, b)] + [t for W, b, _ in layers_b for t in (W, b)]
    return DualMLPFn.apply(xa, xb, relus + ("exact_b",) if exact_b else relus, *params)


# ------------------------------------------------------------------------------------------------
# conv stack of a camera encoder -> NHWC ReLU activations of conv3
# reference: vision_network.py:36-47, vision_network_gripper.py:11-20
# ------------------------------------------------------------------------------------------------
def _conv1_pair_ok(xs, u8, bits, cout, shifts=None, indices=None) -> bool:
    """two frame tensors of one geometry in the bf16 modes — fp32 NCHW, or uint8 NHWC with both or neither augmentation shift (plain tensors, or
    both inputs windows of ONE episode store: then the launch indexes the store with the two index lists back to back): conv1 takes both in one launch (the band kernels' x2); the sign plane of a 32-channel conv1 is one plane, so the two inputs'
    slices are one contiguous tensor"""
    if len(xs) != 2 or xs[0].dtype != xs[1].dtype or xs[0].shape[1:] != xs[1].shape[1:] or not (xs[0].is_contiguous() and xs[1].is_contiguous()):
        return False
    if cout != 32 or kn.base_mode() == "fp32" or kn.get_compute() != "bf16" or os.environ.get("HULC_CONV1_PER_INPUT"):
        return False
    if u8:
        H, W = xs[0].shape[1], xs[0].shape[2]
        sh, ix = shifts or [None, None], indices or [None, None]
        same_store = ix[0] is not None and ix[1] is not None and xs[0].data_ptr() == xs[1].data_ptr() and xs[0].shape == xs[1].shape
        return (xs[0].dtype == torch.uint8 and W % 4 == 0 and (H - 8) % 4 == 0 and (W - 8) % 4 == 0 and ((ix[0] is None and ix[1] is None) or same_store)
                and (sh[0] is None) == (sh[1] is None) and xs[0].data_ptr() % 4 == 0 and xs[1].data_ptr() % 4 == 0)
    return (xs[0].dtype == torch.float32 and xs[0].shape[-1] % 4 == 0 and (xs[0].shape[-2] - 8) % 4 == 0
            and xs[0].data_ptr() % 16 == 0 and xs[1].data_ptr() % 16 == 0)


def _pair_not_covered(e: Exception) -> bool:
    """the library's answer to a two-tensor conv1 launch (hulc_conv_desc.x2) on a geometry its band kernels do not take — small frames
    (W <= 40: fewer than two 8-pixel blocks per output row), LDS / workspace limits, HULC_NO_BAND_WGRAD: nothing was launched, the caller
    runs the per-input loop instead"""
    return "(x2) is taken by the conv1 band kernel only" in str(e)


@_scoped
class ConvStackFn(torch.autograd.Function):
    """xs: one or more (N_i,3,H,W) NCHW frame tensors -> a3 (sum N_i, OH3, OW3, 64) NHWC.  Several inputs (the vis and lang
    modalities of a step) share one activation buffer from conv1 on: conv1 runs per input (no 1 GB concat of the frames),
    every later layer once over all frames.  `grad_premasked`: the incoming gradient has already been multiplied by
    (a3 > 0) by the consumer (spatial softmax backward does it for free)."""

    GEOM = ((8, 4), (4, 2), (3, 1))   # (kernel, stride) of the three layers

    @staticmethod
    def forward(ctx, grad_premasked, aug, w1, b1, w2, b2, w3, b3, *xs):
        """aug: None, or (pad, [shift_i or None per input], [index_i or None per input]) for uint8 NHWC frame tensors (N_i, H, W, 3):
        conv1 applies RandomShiftsAug / ScaleImageTensor / Normalize while staging.  With index_i the tensor is
        the HBM-resident episode store and the batch frames are store frames index_i (windows padded by repeating an index)."""
        xs = [_c(x) for x in xs]
        pad, shifts, indices = (aug if aug is not None else (0, [None] * len(xs), [None] * len(xs)))
        Ns = [x.shape[0] if ix is None else ix.numel() for x, ix in zip(xs, indices)]
        N = sum(Ns)
        u8 = xs[0].dtype == torch.uint8
        if u8:
            _, H, W, C = xs[0].shape
        else:
            _, C, H, W = xs[0].shape
        ws, bs = (w1, w2, w3), (b1, b2, b3)
        acts, dims = [], []
        inp, h, w_, cin = None, H, W, C
        # ReLU sign planes of conv1's and conv2's outputs (one bit per element, written by the forward kernels' epilogues): all the data
        # gradients need of those activations is the sign — conv2's data gradient then reads 20 MB instead of 315 MB per 2048 frames
        want_bits = (_act_dtype() == torch.bfloat16 and any(ctx.needs_input_grad[2:8]) and not os.environ.get("HULC_NO_RELU_BITS"))
        bits = [None, None, None]
        # selective precision site "a3" (DESIGN §5): the stack's OUTPUT (conv3's ReLU map, the spatial softmax's / flatten-linear's input) is
        # kept in fp32 inside a bf16 step — its rounding to bf16 alone costs as much gradient fidelity as all of conv1's operand rounding
        # (round 6: `grad_premasked` may be the pair (grad_premasked, exact_out) — the consumer of the values asks for the twin: the gripper
        #  camera's flatten-linear under site "encfc", every map under site "a3")
        exact_out = False
        if isinstance(grad_premasked, tuple):
            grad_premasked, exact_out = grad_premasked
        a3_exact = _act_dtype() == torch.bfloat16 and kn.base_mode() == "bf16" and (exact_out or "a3" in kn.fp32_sites())
        for li, (k, s) in enumerate(ConvStackFn.GEOM):
            cout = ws[li].shape[0]
            nchw = li == 0
            w2d = weight_operand(ws[li], "oihw_flat" if nchw else "ohwi")
            oh, ow = kn.conv_out_hw(h, w_, k, k, s)
            # (site "a3", round 6) conv3 stores TWO maps from its accumulators: a finer one for whoever consumes the values, the bf16 one the
            # backward pass of a bf16 step works on (hulc_conv_desc.y_bf16) — the stack's differentiable output stays bf16, so neither autograd
            # nor the stack casts the incoming gradient (returning an fp32 map cost two passes over it: 0.145 ms per step).  The static
            # camera's map (23 x 23 -> 21 x 21, the spatial softmax's input) gets an fp16 twin — 11 bits of mantissa at the bf16 map's two
            # bytes: an fp32 twin's 16-byte pieces at 256-byte pixel pitch cost conv3's launch 48 us per 2048 frames and the softmax twice the
            # bytes —, the gripper camera's small map (the exact-fp32 flatten-linear's operand) an fp32 one
            twin = a3_exact and li == 2
            f16_twin = twin and (h, w_, cin, cout, k, s) == (23, 23, 64, 64, 3, 1) and not os.environ.get("HULC_A3_F32_TWIN")
            y = torch.empty(N, oh, ow, cout, dtype=(torch.float16 if f16_twin else torch.float32) if twin else _act_dtype(), device=xs[0].device)
            y16 = torch.empty(N, oh, ow, cout, dtype=torch.bfloat16, device=xs[0].device) if twin else None
            if want_bits and li < 2 and cout % 32 == 0 and (li > 0 or cout == 32):     # (conv1 writes per input tensor: one plane, pixel-major slices)
                bits[li] = torch.empty(N * oh * ow * (cout // 32), dtype=torch.int32, device=xs[0].device)
            if li == 0:
                off = 0
                # selective precision site "conv1": fp32 frames and weights as hi + lo bf16 splits, three MFMAs per product (fp32-class a1)
                w_lo = weight_operand(ws[0], "oihw_flat_lo") if (not u8 and _act_dtype() == torch.bfloat16 and kn.base_mode() == "bf16"
                                                                  and "conv1" in kn.fp32_sites()) else None
                paired = False
                if _conv1_pair_ok(xs, u8, bits[li], cout, shifts, indices):
                    # the two modalities of a step (two frame tensors, never concatenated) as ONE conv1 launch: hulc_conv_desc.x2 (round 4)
                    # (uint8 frames: the per-frame shifts of the two inputs as one small tensor, kept for the weight gradient)
                    pair_shift = torch.cat([shifts[0], shifts[1]]) if (u8 and shifts[0] is not None) else None
                    pair_index = torch.cat([indices[0], indices[1]]) if (u8 and indices[0] is not None) else None
                    ctx.pair_aug = (pair_shift, pair_index)
                    try:
                        kn.conv2d_fwd(xs[0], w2d, bs[li], y, N, h, w_, cin, cout, k, k, s, nchw, relu=True, w_lo=w_lo, relu_bits=bits[li],
                                      x2=None if pair_index is not None else xs[1], aug_shift=pair_shift, aug_pad=pad, frame_index=pair_index)
                        paired = True
                    except kn._L.HulcKernelError as e:
                        if not _pair_not_covered(e):
                            raise
                if not paired:
                    for x, n, sh, ix in zip(xs, Ns, shifts, indices):
                        kn.conv2d_fwd(x, w2d, bs[li], y[off:off + n], n, h, w_, cin, cout, k, k, s, nchw, relu=True, aug_shift=sh, aug_pad=pad,
                                      frame_index=ix, w_lo=w_lo,
                                      relu_bits=None if bits[li] is None else bits[li][off * oh * ow * (cout // 32):(off + n) * oh * ow * (cout // 32)])
                        off += n
            else:
                try:
                    kn.conv2d_fwd(inp, w2d, bs[li], y, N, h, w_, cin, cout, k, k, s, nchw, relu=True, relu_bits=bits[li], y_bf16=y16)
                except kn._L.HulcKernelError:
                    if not (twin and y.dtype == torch.float16):
                        raise
                    # (the fp16 twin is stored by the direct-to-LDS kernel only — e.g. HULC_BAND_PLANES=0 —: the fp32 twin is served everywhere)
                    y = torch.empty(N, oh, ow, cout, dtype=torch.float32, device=xs[0].device)
                    kn.conv2d_fwd(inp, w2d, bs[li], y, N, h, w_, cin, cout, k, k, s, nchw, relu=True, relu_bits=bits[li], y_bf16=y16)
            dims.append((h, w_, cin, cout, k, s, nchw))
            acts.append(y16 if y16 is not None else y)
            inp, h, w_, cin = y, oh, ow, cout
        a3_f32 = y if (a3_exact and y16 is not None) else None
        saved = acts
        if kn.backward_compute() == "bf16" and acts[0].dtype != torch.bfloat16:
            # exact-fp32 forward, bf16 backward ('mixed' mode): the backward kernels get what a bf16 step would have saved — bf16 maps — only
            # rounded ONCE from the exact values (as operands of the gradient products their rounding costs < 1 %)
            saved = []
            for a in acts:
                h = torch.empty(a.shape, dtype=torch.bfloat16, device=a.device)
                kn.cast_f32_to_bf16(a, h, a.numel())
                saved.append(h)
        ctx.g_dtype = saved[1].dtype                      # storage type of the gradient maps the backward kernels work on
        ctx.save_for_backward(saved[0], saved[1], saved[2], w2, w3, *xs)
        ctx.conv_w, ctx.conv_b = ws, bs                   # identities for the gradient sinks
        ctx.relu_bits = bits
        ctx.aug = (pad, shifts, indices)
        ctx.meta = (dims, grad_premasked, Ns)
        if a3_f32 is not None:
            # (the twin carries no gradient, and autograd must not make one up: with materialised gradients the engine filled a 231 MB fp32
            #  zero tensor for it in front of every backward)
            ctx.mark_non_differentiable(a3_f32)
            ctx.set_materialize_grads(False)
            return acts[2], a3_f32
        return acts[2]

    @staticmethod
    def backward(ctx, da3, *unused):
        a1, a2, a3, w2, w3, *xs = ctx.saved_tensors
        dims, premasked, Ns = ctx.meta
        if da3 is None:                                   # (gradients are not materialised when the stack handed out its exact twin)
            da3 = torch.zeros_like(a3)
        N = sum(Ns)
        g = _c(da3)
        if premasked and g.dtype != ctx.g_dtype:          # ('mixed' mode / site "a3": the consumer worked on the fp32 map) -> the saved maps' storage type
            h = torch.empty(g.shape, dtype=ctx.g_dtype, device=g.device)
            kn.cast_f32_to_bf16(g, h, g.numel())
            g = h
        if not premasked:
            gz = torch.empty(g.shape, dtype=torch.float32, device=g.device)
            kn.relu_bwd(g.float() if g.dtype != torch.float32 else g, a3, gz, g.numel())
            g = gz
        inputs = (None, a1, a2)
        weights = (None, w2, w3)
        grads_w, grads_b = [None] * 3, [None] * 3
        for li in (2, 1, 0):
            h, w_, cin, cout, k, s, nchw = dims[li]
            sw, sb = gradsink.get(ctx.conv_w[li]), gradsink.get(ctx.conv_b[li])
            sunk = sw is not None and sb is not None      # the reduce pass writes OIHW straight into the gradient arena
            dw = sw.view(cout, cin * k * k) if sunk else _f32(cout, cin * k * k, like=g)
            db = sb if sunk else _f32(cout, like=g)
            acc = sunk
            if sunk:                                     # dW and db leave one launch: one flag for both
                fw, fb = gradsink.first_write(ctx.conv_w[li]), gradsink.first_write(ctx.conv_b[li])
                acc = not (fw and fb)
            if li == 0:                                  # per input tensor: the second one accumulates
                off = 0
                pad, shifts, indices = ctx.aug
                paired = False
                if _conv1_pair_ok(xs, xs[0].dtype == torch.uint8, None, cout, shifts, indices) and g.dtype == torch.bfloat16:
                    pair_shift, pair_index = getattr(ctx, "pair_aug", (None, None))
                    try:
                        kn.conv2d_bwd_weight(xs[0], g, dw, db, N, h, w_, cin, cout, k, k, s, nchw, dw_oihw=True, accumulate=acc,
                                             x2=None if pair_index is not None else xs[1], aug_shift=pair_shift, aug_pad=pad, frame_index=pair_index)
                        paired = True
                    except kn._L.HulcKernelError as e:
                        if not _pair_not_covered(e):
                            raise
                if not paired:
                    for j, (x, n, sh, ix) in enumerate(zip(xs, Ns, shifts, indices)):
                        kn.conv2d_bwd_weight(x, g[off:off + n], dw, db, n, h, w_, cin, cout, k, k, s, nchw, dw_oihw=True, accumulate=acc or j > 0,
                                             aug_shift=sh, aug_pad=pad, frame_index=ix)
                        off += n
            else:
                kn.conv2d_bwd_weight(inputs[li], g, dw, db, N, h, w_, cin, cout, k, k, s, nchw, dw_oihw=True, accumulate=acc)
            grads_w[li] = None if sunk else dw.view(cout, cin, k, k)
            grads_b[li] = None if sunk else db
            if li > 0:
                inp = inputs[li]
                wt = weight_operand(weights[li], "ihwo")
                dx = torch.empty(N, h, w_, cin, dtype=inp.dtype, device=g.device)
                kn.conv2d_bwd_data(g, wt, dx, inp, N, h, w_, cin, cout, k, k, s, relu_bits=ctx.relu_bits[li - 1])   # masked by relu of the layer input
                g = dx
        return (None, None, grads_w[0], grads_b[0], grads_w[1], grads_b[1], grads_w[2], grads_b[2], *([None] * len(xs)))


def conv_stack(x, params, grad_premasked=False, aug_pad=0, aug_shifts=None, frame_index=None, exact_out=False):
    """x: a frame tensor or a list of them (batched from conv1's output on); fp32 NCHW frames in [-1, 1], or uint8 NHWC frames as
    stored, with aug_shifts (per tensor: (N, 2) int32 or None) and aug_pad applied inside conv1; frame_index (per tensor: int32 store
    frame numbers or None) when x is the episode store itself."""
    xs = list(x) if isinstance(x, (list, tuple)) else [x]
    aug = None
    if xs[0].dtype == torch.uint8:
        sh = list(aug_shifts) if isinstance(aug_shifts, (list, tuple)) else [aug_shifts] * len(xs)
        ix = list(frame_index) if isinstance(frame_index, (list, tuple)) else [frame_index] * len(xs)
        aug = (int(aug_pad), [None if t is None else _c(t.reshape(-1, 2).to(torch.int32)) for t in sh],
               [None if t is None else _c(t.reshape(-1).to(torch.int32)) for t in ix])
    out = ConvStackFn.apply((bool(grad_premasked), True) if exact_out else grad_premasked, aug, *params, *xs)
    if isinstance(out, tuple):             # site "a3": the bf16 map carries its exact twin for the consumers that read values (exact_map())
        out[0]._hulc_f32 = out[1]
        return out[0]
    return out


def exact_map(a: torch.Tensor):
    """the finer twin (fp16 or fp32) of a conv stack's bf16 output map when the stack kept one (precision site "a3"), else None"""
    t = getattr(a, "_hulc_f32", None)
    return t if (t is not None and t.shape == a.shape) else None


@_scoped
class SpatialSoftmaxFn(torch.autograd.Function):
    """NHWC (N,H,W,C) -> (N,2C); backward also applies the ReLU mask of its input (vision_network.py:100-108)."""

    @staticmethod
    def forward(ctx, a, xmap, ymap, temperature, a_exact=None):
        N, H, W, C = a.shape
        out = _f32(N, 2 * C, like=a)
        stats = _f32(N, C, 2, like=a)
        # (site "a3": the expectation is taken over the exact map; the backward recomputes its softmax from the bf16 map, as a bf16 step does)
        kn.spatial_softmax_fwd(a_exact if a_exact is not None else a, N, H * W, C, xmap, ymap, temperature, out, stats)
        ctx.save_for_backward(a, xmap, ymap, temperature, out, stats)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, xmap, ymap, temperature, out, stats = ctx.saved_tensors
        N, H, W, C = a.shape
        # (an fp32 map inside a bf16 step — site "a3": the gradient leaves in the conv stack's gradient type, no cast launch behind it)
        gdt = torch.bfloat16 if (a.dtype == torch.float32 and kn.base_mode() == "bf16" and kn.get_compute() == "bf16") else a.dtype
        dx = torch.empty(a.shape, dtype=gdt, device=a.device)
        kn.spatial_softmax_bwd(a, N, H * W, C, xmap, ymap, temperature, out, stats, _c(dout), dx, relu_mask=True)
        return dx, None, None, None, None


def spatial_softmax(a, xmap, ymap, temperature):
    return SpatialSoftmaxFn.apply(a, xmap, ymap, temperature, exact_map(a))


# ------------------------------------------------------------------------------------------------
# LayerNorm (optionally fused with residual add + dropout of the residual branch)
# ------------------------------------------------------------------------------------------------
@_scoped
class LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, o, gamma, beta, eps: float, drop_p: float, seed: int):
        D = x.shape[-1]
        x2 = _c(x.reshape(-1, D))
        o2 = _c(o.reshape(-1, D)) if o is not None else None
        R = x2.shape[0]
        y = _f32(R, D, like=x2)
        mean, rstd = _f32(R, like=x2), _f32(R, like=x2)
        pre = _f32(R, D, like=x2) if o2 is not None else None
        kn.layernorm_fwd(x2, o2, drop_p, seed, gamma, beta, eps, R, D, pre, y, mean, rstd)
        ctx.save_for_backward(pre if pre is not None else x2, mean, rstd, gamma)
        ctx.beta = beta                                           # identity only (gradient sink lookup)
        ctx.meta = (x.shape, o is not None, drop_p, seed)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        pre, mean, rstd, gamma = ctx.saved_tensors
        shape, has_o, drop_p, seed = ctx.meta
        R, D = pre.shape
        dy2 = _c(dy.reshape(R, D))
        dpre = _f32(R, D, like=dy2)
        do = _f32(R, D, like=dy2) if (has_o and drop_p > 0) else None
        sg, sb = gradsink.get(gamma), gradsink.get(ctx.beta)
        sunk = sg is not None and sb is not None                  # straight into the gradient arena (no AccumulateGrad adds)
        dg, db = (sg, sb) if sunk else (_f32(D, like=dy2), _f32(D, like=dy2))
        acc = sunk
        if sunk:
            fg, fb = gradsink.first_write(gamma), gradsink.first_write(ctx.beta)
            acc = not (fg and fb)
        kn.layernorm_bwd(dy2, pre, mean, rstd, gamma, R, D, dpre, do, drop_p, seed, dg, db, accumulate_params=acc)
        dx = dpre.reshape(shape)
        d_o = None
        if has_o:
            d_o = (do if do is not None else dpre).reshape(shape)
        return dx, d_o, (None if sunk else dg), (None if sunk else db), None, None, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return LayerNormFn.apply(x, None, gamma, beta, eps, 0.0, 0)


def add_layer_norm(x, o, gamma, beta, eps=1e-5, drop_p=0.0, seed=0):
    """LayerNorm(x + dropout(o)) — the post-norm residual step of nn.TransformerEncoderLayer."""
    return LayerNormFn.apply(x, o, gamma, beta, eps, drop_p, seed)


@_scoped
class LayerNormCatFn(torch.autograd.Function):
    """cat([LayerNorm_i(x_i)], dim) without the concat: every LayerNorm writes its block of ONE output tensor (rows ld apart) and its backward
    reads its block of the incoming gradient in place — the cameras' 64 + 64 halves of the perceptual embedding (dim = -1,
    concat_encoders.py:96-107) and the modalities' latent goals stacked on the batch axis (dim = 0).  args = x_0 .. x_{n-1}, gamma_0, beta_0, ..."""

    @staticmethod
    def forward(ctx, dim: int, eps: float, n: int, *args):
        xs = [_c(x.reshape(-1, x.shape[-1])) for x in args[:n]]
        params = args[n:]
        if dim == 0:
            D = xs[0].shape[1]
            R = sum(x.shape[0] for x in xs)
            offs, acc = [], 0
            for x in xs:
                offs.append(acc * D)
                acc += x.shape[0]
            ld = D
        else:
            R = xs[0].shape[0]
            D = sum(x.shape[1] for x in xs)
            offs, acc = [], 0
            for x in xs:
                offs.append(acc)
                acc += x.shape[1]
            ld = D
        y = _f32(R, D, like=xs[0])
        flat = y.view(-1)
        stats = []
        for i, x in enumerate(xs):
            r, d = x.shape
            mean, rstd = _f32(r, like=x), _f32(r, like=x)
            kn.layernorm_fwd_ld(x, params[2 * i], params[2 * i + 1], eps, r, d, flat[offs[i]:], ld, mean, rstd)
            stats += [mean, rstd]
        ctx.save_for_backward(*xs, *stats, *[params[2 * i] for i in range(n)])
        ctx.betas = [params[2 * i + 1] for i in range(n)]
        ctx.meta = (n, offs, ld, [tuple(a.shape) for a in args[:n]])
        return y

    @staticmethod
    def backward(ctx, dy):
        n, offs, ld, shapes = ctx.meta
        sv = ctx.saved_tensors
        xs, stats, gammas = sv[:n], sv[n:3 * n], sv[3 * n:]
        dyf = _c(dy).view(-1)
        dxs, dps = [], []
        for i in range(n):
            x, gamma, beta = xs[i], gammas[i], ctx.betas[i]
            r, d = x.shape
            dx = _f32(r, d, like=dyf)
            sg, sb = gradsink.get(gamma), gradsink.get(beta)
            sunk = sg is not None and sb is not None
            dg, db = (sg, sb) if sunk else (_f32(d, like=dyf), _f32(d, like=dyf))
            acc = sunk
            if sunk:
                fg, fb = gradsink.first_write(gamma), gradsink.first_write(beta)
                acc = not (fg and fb)
            kn.layernorm_bwd_ld(dyf[offs[i]:], ld, x, stats[2 * i], stats[2 * i + 1], gamma, r, d, dx, dg, db, accumulate_params=acc)
            dxs.append(dx.reshape(shapes[i]))
            dps += [None if sunk else dg, None if sunk else db]
        return (None, None, None, *dxs, *dps)


def layer_norm_cat(xs, norms, dim: int):
    """cat([F.layer_norm(x_i, ...)], dim) for nn.LayerNorm modules `norms` (2-D results: rows = all leading axes flattened)"""
    params = [t for m in norms for t in (m.weight, m.bias)]
    return LayerNormCatFn.apply(int(dim), float(norms[0].eps), len(xs), *xs, *params)


@_scoped
class PlanSampleKLFn(torch.autograd.Function):
    """The posterior's two consumers in one node: straight-through plan sample (hulc2.py:235-237) + balanced KL against the prior
    (hulc2.py:444-466) -> (plan, idx, kl (nseg,)).  Backward adds the sample's gradient onto the KL's in the sampling kernel (`accumulate`):
    no gradient fan-in add on the (rows, 1024) logits."""

    @staticmethod
    def forward(ctx, pp, pr, idx_in, G: int, CLS: int, seed: int, beta: float, mix: float, nseg: int):
        pp, pr = _c(pp), _c(pr)
        B = pr.shape[0]
        plan = _f32(B, G * CLS, like=pr)
        idx = torch.empty(B, G, dtype=torch.long, device=pr.device)
        kn.plan_sample_fwd(pr, _c(idx_in) if idx_in is not None else None, seed, B * G, CLS, idx, plan)
        out = _f32(nseg, like=pp)
        klg = _f32(B * G, like=pp)
        kn.cat_kl_fwd(pp, pr, B, G, CLS, beta, out, klg, nseg)
        ctx.save_for_backward(pp, pr, klg)
        ctx.meta = (B, G, CLS, beta, mix, nseg)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return plan, idx, out

    @staticmethod
    def backward(ctx, dplan, _didx=None, gkl=None):
        pp, pr, klg = ctx.saved_tensors
        B, G, CLS, beta, mix, nseg = ctx.meta
        dpp = dpr = None
        if gkl is not None:
            dpp, dpr = torch.empty_like(pp), torch.empty_like(pr)
            kn.cat_kl_bwd(pp, pr, klg, B, G, CLS, beta, mix, _c(gkl.reshape(nseg)), dpp, dpr, nseg)
        if dplan is not None:
            if dpr is None:
                dpr = torch.empty_like(pr)
                kn.plan_sample_bwd(pr, _c(dplan), B * G, CLS, dpr)
            else:
                kn.plan_sample_bwd(pr, _c(dplan), B * G, CLS, dpr, accumulate=True)
        return dpp, dpr, None, None, None, None, None, None, None


# ------------------------------------------------------------------------------------------------
# transformer pieces
# ------------------------------------------------------------------------------------------------
@_scoped
class AddPosFn(torch.autograd.Function):
    """dropout(x + pos[position_ids]) — plan_recognition_net.py:133-136,142.  identity: position_ids is arange(S) (what the reference passes):
    the table's gradient is then the batch sum itself and goes straight into the trainer's gradient sink."""

    @staticmethod
    def forward(ctx, x, pos, pos_ids, drop_p: float, seed: int, identity: bool = False):
        B, S, D = x.shape
        y = _f32(B, S, D, like=x)
        kn.add_pos_fwd(_c(x), pos, pos_ids, y, B, S, D, drop_p, seed)
        ctx.save_for_backward(pos_ids)
        ctx.meta = (B, S, D, drop_p, seed, pos.shape)
        ctx.pos = pos if identity else None
        return y

    @staticmethod
    def backward(ctx, dy):
        (pos_ids,) = ctx.saved_tensors
        B, S, D, drop_p, seed, pshape = ctx.meta
        dy = _c(dy)
        if drop_p > 0:
            dx = torch.empty_like(dy)
            kn.dropout_bwd(dy, dx, dy.numel(), drop_p, seed)
        else:
            dx = dy
        sink = gradsink.get(ctx.pos) if ctx.pos is not None else None
        if sink is not None and dx.dtype == torch.float32:
            # rows 0 .. S-1 of the table's gradient = the sum over the batch, written (first writer of the step) or added in place
            first = gradsink.first_write(ctx.pos)
            kn.colsum(dx, B, S * D, S * D, sink[:S].view(-1), accumulate=not first)
            if first and pshape[0] > S:
                sink[S:].zero_()
            return dx, None, None, None, None, None
        dsum = _f32(S, D, like=dy)
        kn.colsum(dx, B, S * D, S * D, dsum)                 # sum over the batch
        dpos = torch.zeros(pshape, dtype=torch.float32, device=dy.device)
        dpos.index_copy_(0, pos_ids, dsum)                   # row placement by (unique) position id: a copy, no arithmetic
        return dx, dpos, None, None, None, None


@_scoped
class SeqMeanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        B, S, D = x.shape
        y = _f32(B, D, like=x)
        kn.seq_mean_fwd(_c(x), y, B, S, D)
        ctx.meta = (B, S, D)
        return y

    @staticmethod
    def backward(ctx, dy):
        B, S, D = ctx.meta
        dx = _f32(B, S, D, like=dy)
        kn.seq_mean_bwd(_c(dy), dx, B, S, D)
        return dx


@_scoped
class AttentionFn(torch.autograd.Function):
    """softmax(q k^T / sqrt(dh)) v per (batch, head) on packed qkv (B*S, 3E)."""

    @staticmethod
    def forward(ctx, qkv, B: int, S: int, H: int, drop_p: float, seed: int):
        E = qkv.shape[-1] // 3
        dh = E // H
        qkv = _c(qkv)
        out = _f32(B * S, E, like=qkv)
        probs = _f32(B, H, S, S, like=qkv)
        kn.attention_fwd(qkv, out, probs, B, S, H, dh, drop_p, seed)
        ctx.save_for_backward(qkv, probs)
        ctx.meta = (B, S, H, dh, drop_p, seed)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, probs = ctx.saved_tensors
        B, S, H, dh, drop_p, seed = ctx.meta
        dqkv = torch.empty_like(qkv)
        kn.attention_bwd(qkv, probs, _c(dout), dqkv, B, S, H, dh, drop_p, seed)
        return dqkv, None, None, None, None, None


@_scoped
class FFNFn(torch.autograd.Function):
    """linear1 -> ReLU -> dropout -> linear2 of the transformer layer as one fused kernel per direction (csrc/ffn_fused.hip):
    the (tokens x 2048) hidden activation never reaches HBM, backward recomputes it.  x (T, 128) fp32 -> f (T, 128) fp32."""

    @staticmethod
    def forward(ctx, x, W1, b1, W2, b2, drop_p: float, seed: int):
        x2 = _c(x.reshape(-1, x.shape[-1]))
        T, D = x2.shape
        FF = W1.shape[0]
        f = _f32(T, D, like=x2)
        kn.ffn_fwd(x2, weight_operand(W1), b1, weight_operand(W2), b2, T, D, FF, drop_p, seed, f)
        ctx.save_for_backward(x2, W1, b1, W2, b2)
        ctx.meta = (x.shape, T, D, FF, drop_p, seed)
        return f.reshape(x.shape)

    @staticmethod
    def backward(ctx, df):
        x2, W1, b1, W2, b2 = ctx.saved_tensors
        shape, T, D, FF, drop_p, seed = ctx.meta
        df2 = _c(df.reshape(T, D))
        sinks = [gradsink.get(t) for t in (W1, b1, W2, b2)]
        sunk = all(s is not None for s in sinks)
        dW1, db1, dW2, db2 = sinks if sunk else (_f32(FF, D, like=df2), _f32(FF, like=df2), _f32(D, FF, like=df2), _f32(D, like=df2))
        dx = _f32(T, D, like=df2)
        acc, acc_b2 = sunk, sunk
        if sunk:
            f = [gradsink.first_write(t) for t in (W1, b1, W2, b2)]
            acc, acc_b2 = not all(f[:3]), not f[3]
        kn.ffn_bwd(x2, df2, weight_operand(W1), b1, weight_operand(W1, "t"), weight_operand(W2, "t"), T, D, FF, drop_p, seed, dx, dW1, db1, dW2,
                   accumulate_params=acc)
        kn.colsum(df2, T, D, D, db2, accumulate=acc_b2)
        g = (None, None, None, None) if sunk else (dW1, db1, dW2, db2)
        return (dx.reshape(shape), *g, None, None)


def _sink_or_new(param, shape, like):
    """-> (tensor the kernels write, accumulate flag, None-or-tensor to hand back to autograd)"""
    sk = gradsink.get(param)
    if sk is None:
        t = _f32(*shape, like=like)
        return t, False, t
    return sk.view(*shape), not gradsink.first_write(param), None


@_scoped
class TxlLayerFn(torch.autograd.Function):
    """One post-norm nn.TransformerEncoderLayer (plan_recognition_net.py:115-117) on tokens x (B*S, 128), bf16 compute, as
         forward : hulc_txl_attn_fwd (in_proj + MFMA attention + out_proj + residual + dropout + LayerNorm1, one workgroup per sequence)
                   -> hulc_ffn_fwd (hidden-slice partials) -> hulc_layernorm_slab_fwd (slice sum + residual + dropout + LayerNorm2)
         backward: hulc_layernorm_bwd -> hulc_ffn_bwd (dx left as slice partials) -> hulc_txl_attn_bwd (sums them, LayerNorm1 backward,
                   attention / projection data gradients) -> two weight-gradient GEMMs over all tokens + the small reductions.
    Autograd sees one node per layer: no gradient fan-in adds, no intermediate tensors between the kernels."""

    NP = 12

    @staticmethod
    def forward(ctx, x, B: int, S: int, H: int, drop_p: float, seed: int, w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2):
        x2 = _c(x.reshape(B * S, x.shape[-1]))
        T, E = x2.shape
        FF = w1.shape[0]
        keep = any(ctx.needs_input_grad)               # (grad mode is off inside Function.forward: ask which inputs want gradients)
        y1 = _f32(T, E, like=x2)
        pre1, mean1, rstd1 = (_f32(T, E, like=x2), _f32(T, like=x2), _f32(T, like=x2)) if keep else (None, None, None)
        ctxb = torch.empty(T, E, dtype=torch.bfloat16, device=x2.device) if keep else None
        kn.txl_attn_fwd(x2, weight_operand(w_in), b_in, weight_operand(w_out), b_out, g1, be1, 1e-5, B, S, H, drop_p, seed + 11, seed + 12,
                        y1, pre1, mean1, rstd1, ctxb)
        ws = kn.ffn_fwd(y1, weight_operand(w1), b1, weight_operand(w2), b2, T, E, FF, drop_p, seed + 13, None)
        y2, pre2, mean2, rstd2 = _f32(T, E, like=x2), _f32(T, E, like=x2), _f32(T, like=x2), _f32(T, like=x2)
        kn.layernorm_slab_fwd(y1, ws, FF // 128, T * E, drop_p, seed + 15, g2, be2, 1e-5, T, E, pre2, y2, mean2, rstd2)
        if keep:
            ctx.save_for_backward(x2, y1, pre1, mean1, rstd1, ctxb, pre2, mean2, rstd2, w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2)
            ctx.meta = (B, S, H, drop_p, seed, x.shape)
        return y2.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (x2, y1, pre1, mean1, rstd1, ctxb, pre2, mean2, rstd2, w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2) = ctx.saved_tensors
        B, S, H, drop_p, seed, xshape = ctx.meta
        T, E = x2.shape
        FF = w1.shape[0]
        dy2 = _c(dy.reshape(T, E))
        like = dy2
        outs = {}

        def sink(name, param, shape):
            t, acc, ret = _sink_or_new(param, shape, like)
            outs[name] = ret
            return t, acc
        # LayerNorm2 backward: dpre2 = gradient of y1 through the residual, df = gradient of the feed-forward output
        dpre2 = _f32(T, E, like=like)
        df = _f32(T, E, like=like) if drop_p > 0 else dpre2
        (dg2, a1), (db2n, a2) = sink("g2", g2, (E,)), sink("be2", be2, (E,))
        kn.layernorm_bwd(dy2, pre2, mean2, rstd2, g2, T, E, dpre2, df if drop_p > 0 else None, drop_p, seed + 15, dg2, db2n, accumulate_params=a1 or a2)
        # feed-forward backward: weight gradients complete, input gradient left as FF / 128 slice partials
        (dW1, c1), (db1, c2), (dW2, c3), (db2, c4) = sink("w1", w1, (FF, E)), sink("b1", b1, (FF,)), sink("w2", w2, (E, FF)), sink("b2", b2, (E,))
        ws = kn.ffn_bwd(y1, df, weight_operand(w1), b1, weight_operand(w1, "t"), weight_operand(w2, "t"), T, E, FF, drop_p, seed + 13, None, dW1, db1,
                        dW2, accumulate_params=c1 or c2 or c3)
        kn.colsum(df, T, E, E, db2, accumulate=c4)
        # attention half backward
        dx = _f32(T, E, like=like)
        d_o = torch.empty(T, E, dtype=torch.bfloat16, device=like.device)
        dqkv = torch.empty(T, 3 * E, dtype=torch.bfloat16, device=like.device)
        lnp = _f32(B, 2, E, like=like)
        kn.txl_attn_bwd(x2, weight_operand(w_in), weight_operand(w_in, "t"), weight_operand(w_out, "t"), b_in, g1, 1e-5, B, S, H, drop_p, seed + 11,
                        seed + 12, pre1, mean1, rstd1, dpre2, ws, FF // 128, T * E, dx, d_o, dqkv, lnp)
        (dg1, e1), (db1n, e2) = sink("g1", g1, (E,)), sink("be1", be1, (E,))
        kn.ln_partial_reduce(lnp, B, E, dg1, db1n, accumulate=e1 or e2)
        # weight gradients of the two projections over all tokens; the row sums of the left operand are the bias gradients
        (dWin, f1), (dbin, f2) = sink("w_in", w_in, (3 * E, E)), sink("b_in", b_in, (3 * E,))
        kn.wgrad(dqkv, x2, dWin, 3 * E, E, T, 3 * E, E, E, accumulate=f1, rowsum=dbin, rowsum_accumulate=f2,
                 defer=outs["w_in"] is None and outs["b_in"] is None)
        (dWo, h1), (dbo, h2) = sink("w_out", w_out, (E, E)), sink("b_out", b_out, (E,))
        kn.wgrad(d_o, ctxb, dWo, E, E, T, E, E, E, accumulate=h1, rowsum=dbo, rowsum_accumulate=h2,
                 defer=outs["w_out"] is None and outs["b_out"] is None)
        g = [outs[k] for k in ("w_in", "b_in", "w_out", "b_out", "w1", "b1", "w2", "b2", "g1", "be1", "g2", "be2")]
        return (dx.reshape(xshape), None, None, None, None, None, *g)


def _txl_fused_ok(x, p: dict, S: int, nhead: int) -> bool:
    return (kn.get_compute() == "bf16" and x.shape[-1] == 128 and nhead == 8 and 1 <= S <= 32 and p["linear1.weight"].shape[0] % 128 == 0
            and x.dtype == torch.float32 and not os.environ.get("HULC_NO_FUSED_TXL") and not os.environ.get("HULC_NO_FUSED_FFN"))


def _ffn_fused_ok(x, W1) -> bool:
    return (kn.get_compute() == "bf16" and x.shape[-1] == 128 and W1.shape[0] % 128 == 0 and x.dtype == torch.float32
            and not os.environ.get("HULC_NO_FUSED_FFN"))


def transformer_encoder_layer(x, p: dict, B: int, S: int, nhead: int, drop_p: float, seed: int):
    """Post-norm nn.TransformerEncoderLayer (ReLU, eps 1e-5) on tokens x (B*S, E) — plan_recognition_net.py:115-117."""
    if _txl_fused_ok(x, p, S, nhead):
        return TxlLayerFn.apply(x, B, S, nhead, float(drop_p), int(seed), p["in_proj_weight"], p["in_proj_bias"], p["out_proj.weight"],
                                p["out_proj.bias"], p["linear1.weight"], p["linear1.bias"], p["linear2.weight"], p["linear2.bias"],
                                p["norm1.weight"], p["norm1.bias"], p["norm2.weight"], p["norm2.bias"])
    qkv = mlp(x, [(p["in_proj_weight"], p["in_proj_bias"], False)])
    att = AttentionFn.apply(qkv, B, S, nhead, drop_p, seed + 11)
    o = mlp(att, [(p["out_proj.weight"], p["out_proj.bias"], False)])
    x = add_layer_norm(x, o, p["norm1.weight"], p["norm1.bias"], 1e-5, drop_p, seed + 12)
    if _ffn_fused_ok(x, p["linear1.weight"]):
        ff = FFNFn.apply(x, p["linear1.weight"], p["linear1.bias"], p["linear2.weight"], p["linear2.bias"], float(drop_p), int(seed + 13))
    else:
        ff = mlp(x, [(p["linear1.weight"], p["linear1.bias"], True), (p["linear2.weight"], p["linear2.bias"], False)],
                 drops=[drop_p, 0.0], seed=seed + 13)
    return add_layer_norm(x, ff, p["norm2.weight"], p["norm2.bias"], 1e-5, drop_p, seed + 15)


_TXL_KEYS = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "linear1.weight", "linear1.bias", "linear2.weight",
             "linear2.bias", "norm1.weight", "norm1.bias", "norm2.weight", "norm2.bias")


@_scoped
class TxlBlockFn(torch.autograd.Function):
    """PlanRecognitionTransformersNetwork.forward up to the sequence mean (plan_recognition_net.py:125-146) as ONE launch per direction
    (csrc/txl_block.hip): dropout(emb + pos) -> L transformer layers -> mean over the sequence, one workgroup per sequence.  Backward leaves
    the bf16 operands of the 4 L weight-gradient products (they join the pass's grouped launch) and the LayerNorm partials.
    params: 12 per layer in _TXL_KEYS order.  Same dropout streams as AddPosFn / TxlLayerFn with the same seeds."""

    @staticmethod
    def forward(ctx, emb, pos, pos_ids, H: int, drop_p: float, seed: int, *params):
        B, S, E = emb.shape
        L = len(params) // 12
        T = B * S
        FF = params[4].shape[0]
        emb = _c(emb)
        keep = any(ctx.needs_input_grad)
        dev = emb.device
        x = _f32(T, E, like=emb)
        recs, kept = [], []
        x3 = "txl" in kn.fp32_sites()
        for li in range(L):
            w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
            sd = seed + 100 * (li + 1)
            t = {"y1": _f32(T, E, like=emb), "y2": _f32(T, E, like=emb)}
            if keep:
                t.update(pre1=_f32(T, E, like=emb), mean1=_f32(T, like=emb), rstd1=_f32(T, like=emb), pre2=_f32(T, E, like=emb),
                         mean2=_f32(T, like=emb), rstd2=_f32(T, like=emb), ctx=torch.empty(T, E, dtype=torch.bfloat16, device=dev))
            lo = {}
            if x3:                 # fp32-class forward: every product from hi / lo splits of both operands (the remainders of the bf16 shadows)
                lo = dict(Wqkv_lo=weight_operand(w_in, "lo"), Wo_lo=weight_operand(w_out, "lo"), W1p_lo=weight_operand(w1, "ffn_p0_lo"),
                          W2p_lo=weight_operand(w2, "ffn_p1_lo"))
            recs.append(dict(Wqkv=weight_operand(w_in), Wo=weight_operand(w_out), W1=weight_operand(w1), W2=weight_operand(w2),
                             W1p=weight_operand(w1, "ffn_p0"), W2p=weight_operand(w2, "ffn_p1"), **lo, bqkv=b_in, bo=b_out,
                             b1=b1, b2=b2, g1=g1, be1=be1, g2=g2, be2=be2, seed_attn=sd + 11, seed_ln1=sd + 12, seed_ffn=sd + 13, seed_ln2=sd + 15,
                             x=x, **t))
            kept.append(t)
            x = t["y2"]
        pooled = _f32(B, E, like=emb)
        d = kn.txl_block_desc(emb, pos, pos_ids, B, S, H, FF, drop_p, seed, 1e-5, recs, pooled=pooled)
        kn.txl_block_fwd(d, B, S, H, E, FF, L)
        if keep:
            flat = [recs[0]["x"]]
            for t in kept:
                flat += [t[k] for k in ("y1", "pre1", "mean1", "rstd1", "ctx", "y2", "pre2", "mean2", "rstd2")]
            ctx.save_for_backward(emb, pos, pos_ids, *flat, *params)
            ctx.meta = (B, S, E, H, FF, L, drop_p, seed)
            ctx.pos_identity = bool(getattr(pos_ids, "_hulc_arange", False)) and pos_ids.numel() == S
            ctx.share = kn.coop_share()                    # (the posterior as a forked branch: the backward launch keeps to the same share of the device)
        return pooled

    @staticmethod
    def backward(ctx, dpooled):
        B, S, E, H, FF, L, drop_p, seed = ctx.meta
        saved = ctx.saved_tensors
        emb, pos, pos_ids, x0 = saved[:4]
        acts, params = saved[4:4 + 9 * L], saved[4 + 9 * L:]
        T = B * S
        dev = emb.device
        like = dpooled = _c(dpooled)
        bf = dict(dtype=torch.bfloat16, device=dev)
        recs, outs = [], []
        x = x0
        lnp = _f32(2 * L, B, 2, E, like=like)              # every LayerNorm's per-sequence partials: one reduce launch
        for li in range(L):
            w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
            y1, pre1, mean1, rstd1, ctxb, y2, pre2, mean2, rstd2 = acts[9 * li:9 * li + 9]
            sd = seed + 100 * (li + 1)
            o = dict(d_o=torch.empty(T, E, **bf), dqkv=torch.empty(T, 3 * E, **bf), df=torch.empty(T, E, **bf), h=torch.empty(T, FF, **bf),
                     dh=torch.empty(T, FF, **bf), lnp1=lnp[2 * li], lnp2=lnp[2 * li + 1])
            recs.append(dict(Wqkv=weight_operand(w_in), Wo=weight_operand(w_out), W1=weight_operand(w1), W2=weight_operand(w2),
                             WqkvT=weight_operand(w_in, "t"), WoT=weight_operand(w_out, "t"), W1T=weight_operand(w1, "t"), W2T=weight_operand(w2, "t"),
                             W1p=weight_operand(w1, "ffn_p0"), W2Tp=weight_operand(w2, "ffn_p2"), W1Tp=weight_operand(w1, "ffn_p3"),
                             bqkv=b_in, bo=b_out, b1=b1, b2=b2, g1=g1, be1=be1, g2=g2, be2=be2,
                             seed_attn=sd + 11, seed_ln1=sd + 12, seed_ffn=sd + 13, seed_ln2=sd + 15,
                             x=x, y1=y1, pre1=pre1, mean1=mean1, rstd1=rstd1, ctx=ctxb, y2=y2, pre2=pre2, mean2=mean2, rstd2=rstd2, **o))
            outs.append((o, x, y1, ctxb))
            x = y2
        demb = _f32(B, S, E, like=like)
        d = kn.txl_block_desc(emb, pos, pos_ids, B, S, H, FF, drop_p, seed, 1e-5, recs, dpooled=dpooled, demb=demb)
        kn.txl_block_bwd(d, B, S, H, E, FF, L, share=ctx.share)
        grads = []
        rets = []
        dgs, dbs, accs = [], [], []
        for li in range(L):
            w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
            ret = {}
            rets.append(ret)
            for gm, bt, kg, kb in ((g1, be1, "g1", "be1"), (g2, be2, "g2", "be2")):
                (dg, a1, ret[kg]), (db, a2, ret[kb]) = _sink_or_new(gm, (E,), like), _sink_or_new(bt, (E,), like)
                dgs.append(dg); dbs.append(db); accs.append(a1 or a2)
        kn.ln_partial_reduce_multi(lnp, B, E, dgs, dbs, accs)
        for li in range(L):
            w_in, b_in, w_out, b_out, w1, b1, w2, b2, g1, be1, g2, be2 = params[12 * li:12 * li + 12]
            o, xin, y1, ctxb = outs[li]
            ret = rets[li]

            def sink(name, param, shape):
                t, acc, r = _sink_or_new(param, shape, like)
                ret[name] = r
                return t, acc
            # dW = (left operand)^T (right operand) over all T tokens, bias gradient = the left operand's column sums
            for left, right, W, bias, kw, kb_, M, N in ((o["dqkv"], xin, w_in, b_in, "w_in", "b_in", 3 * E, E), (o["d_o"], ctxb, w_out, b_out, "w_out", "b_out", E, E),
                                                        (o["dh"], y1, w1, b1, "w1", "b1", FF, E), (o["df"], o["h"], w2, b2, "w2", "b2", E, FF)):
                (dW, f1), (dbias, f2) = sink(kw, W, (M, N)), sink(kb_, bias, (M,))
                kn.wgrad(left, right, dW, M, N, T, M, N, N, accumulate=f1, rowsum=dbias, rowsum_accumulate=f2, defer=ret[kw] is None and ret[kb_] is None)
            grads += [ret[k] for k in ("w_in", "b_in", "w_out", "b_out", "w1", "b1", "w2", "b2", "g1", "be1", "g2", "be2")]
        # the position table: rows pos_ids of its gradient = the sum of demb over the batch (AddPosFn.backward)
        dpos = None
        sk = gradsink.get(pos)
        identity = ctx.needs_input_grad[1] and sk is not None and ctx.pos_identity
        if ctx.needs_input_grad[1]:
            if identity:
                first = gradsink.first_write(pos)
                kn.colsum(demb, B, S * E, S * E, sk[:S].view(-1), accumulate=not first)
                if first and pos.shape[0] > S:
                    sk[S:].zero_()
            else:
                dsum = _f32(S, E, like=like)
                kn.colsum(demb, B, S * E, S * E, dsum)
                dpos = torch.zeros(pos.shape, dtype=torch.float32, device=dev)
                dpos.index_copy_(0, pos_ids, dsum)
        return (demb if ctx.needs_input_grad[0] else None, dpos, None, None, None, None, *grads)


def txl_block_ok(emb, layer_params, S: int, nhead: int) -> bool:
    """the whole-trunk launch takes the configured posterior (d_model 128, 8 heads, S <= 32, <= 4 layers, bf16 compute in every site it spans)"""
    p0 = layer_params[0]
    return (emb.is_cuda and kn.get_compute() == "bf16" and kn.base_mode() == "bf16"
            and emb.shape[-1] == 128 and nhead == 8 and 1 <= S <= 32 and 1 <= len(layer_params) <= 4 and emb.dtype == torch.float32
            and p0["linear1.weight"].shape[0] % 128 == 0 and all(p["linear1.weight"].shape == p0["linear1.weight"].shape for p in layer_params)
            and not os.environ.get("HULC_NO_TXL_BLOCK") and not os.environ.get("HULC_NO_FUSED_TXL") and not os.environ.get("HULC_NO_FUSED_FFN"))


def transformer_trunk_pooled(emb, pos, pos_ids, layer_params, nhead: int, drop_p: float, seed: int):
    """mean_s(TransformerEncoder(dropout(emb + pos[pos_ids]))) — only call after txl_block_ok"""
    params = [p[k] for p in layer_params for k in _TXL_KEYS]
    return TxlBlockFn.apply(emb, pos, pos_ids, int(nhead), float(drop_p), int(seed), *params)


# ------------------------------------------------------------------------------------------------
# action decoder recurrence: 2-layer ReLU RNN over [plan | emb_slice | goal]
# reference: logistic_decoder_rnn.py:257-270 + decoders/utils/rnn.py:5-14
# ------------------------------------------------------------------------------------------------
def _rnn_persistent(B: int, Hd: int, state_dtype, device) -> bool:
    """the persistent wavefront kernel covers the benchmarked geometry (bf16 compute, H = 2048, <= 64 rows, fp32 state) on a whole
    MI355X (its 256 workgroups must all be resident, one per CU: a CPX/DPX partition or a CU-masked device shows fewer CUs);
    other geometries and the exact-fp32 mode use the per-step GEMMs.  HULC_NO_RNN_WAVEFRONT=1 forces the per-step path."""
    return (kn.get_compute() == "bf16" and Hd == 2048 and B <= 64 and state_dtype == torch.float32
            and kn.device_cu_count(device) >= 256
            and not kn.concurrent_streams()          # its device-wide barrier needs the GPU to itself (kernels.set_concurrent_streams)
            and not os.environ.get("HULC_NO_RNN_WAVEFRONT"))


def _decoder_rnn_forward(plan, emb, goal, lo: int, hi: int, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, h0, emb_tm: bool = False):
    """Forward sweep shared by training (DecoderRNNFn) and inference (decoder_rnn_infer).  h0: None (zero initial state, the
    training path) or (2, B, H) initial hidden states of the two layers (stateful `act`).  Returns the time-major state buffer
    zbuf (S+2, B, 2H) with zbuf[t+1] = [h0_t | h1_{t-1}], the operands backward needs, and whether the persistent kernel ran."""
    # emb_tm: `emb` already IS the decoder's column slice in time-major order, (S, B, hi - lo) (EmbFanoutFn)
    if emb_tm:
        S, B, _ = emb.shape
    else:
        B, S, _ = emb.shape
    Hd = w_hh0.shape[0]
    P, G, E = plan.shape[1], goal.shape[1], hi - lo
    plan, goal = _c(plan), _c(goal)
    dev = emb.device
    Kin = w_ih0.shape[1]
    wih0 = weight_operand(w_ih0)
    # per-sequence constant part c = plan Wp^T + goal Wg^T + b_ih0 (b_hh0 is added in the recurrent step)
    c = torch.empty(B, Hd, dtype=torch.float32, device=dev)
    kn.gemm(plan, wih0, c, B, Hd, P, P, Kin, Hd, bias=b_ih0)
    kn.gemm(goal, wih0[:, P + E:], c, B, Hd, G, G, Kin, Hd, accumulate=True)
    emb_t = _c(emb) if emb_tm else emb[:, :, lo:hi].permute(1, 0, 2).contiguous()                   # (S, B, E) time-major
    zdt0 = _act_dtype() if os.environ.get("HULC_RNN_STATE_BF16") else torch.float32
    persistent0 = h0 is None and _rnn_persistent(B, Hd, zdt0, dev)
    if persistent0:       # the per-sequence constant c enters the recurrent kernel as its own (step-independent) term: no (S, B, H) expand
        pre0 = torch.empty(S, B, Hd, dtype=torch.float32, device=dev)
        kn.gemm(emb_t, wih0[:, P:P + E], pre0, S * B, Hd, E, E, Kin, Hd)
    else:
        pre0 = c.unsqueeze(0).expand(S, B, Hd).contiguous()
        kn.gemm(emb_t, wih0[:, P:P + E], pre0, S * B, Hd, E, E, Kin, Hd, accumulate=True)
    # state kept fp32: a bf16 state halves the step traffic but the big reduction-major wgrad GEMMs over it then run on
    # 2-byte strided loads and lose more than the steps gain (tools/decoder_bench.py: 3.07 ms fp32 vs 3.69 ms bf16)
    zdt = _act_dtype() if os.environ.get("HULC_RNN_STATE_BF16") else torch.float32
    persistent = h0 is None and _rnn_persistent(B, Hd, zdt, dev)
    if persistent:
        # the persistent kernel writes every row it owns (rows 1..S+1, zeros included): only the initial row and the half of the last
        # row it never produces are cleared, not 36 MB
        zbuf = torch.empty(S + 2, B, 2 * Hd, dtype=zdt, device=dev)            # zbuf[t+1] = [h0_t | h1_{t-1}]; edges cleared by the launcher
    else:
        zbuf = torch.zeros(S + 2, B, 2 * Hd, dtype=zdt, device=dev)
    if h0 is not None:                                                         # carried state: h0_{-1} and h1_{-1}
        zbuf[0][:, :Hd] = h0[0]
        zbuf[1][:, Hd:] = h0[1]
    whh0 = weight_operand(w_hh0)
    meta = (B, S, Hd, P, G, E, lo, hi, emb.shape[2], emb_tm)
    if persistent:
        # both layers, all S steps: one persistent kernel with register-resident weights (csrc/rnn_wavefront.hip)
        z16 = kn.rnn_wavefront(zbuf[0], B * 2 * Hd, S, B, Hd, whh0, weight_operand(w_ih1), weight_operand(w_hh1), False,
                               add1=pre0, add1_step=B * Hd, ld_add1=Hd, bias1=(b_hh0, None), bias2=(b_ih1, b_hh1), relu=True,
                               mirror_t=bool(os.environ.get("HULC_RNN_WGRAD_TMIRROR")), add1c=c, zero_edges=True)
        return zbuf, plan, emb_t, goal, z16, meta
    w1cat = weight_operand(torch.cat([w_ih1.detach(), w_hh1.detach()], dim=1))   # (H, 2H) = [W_ih1 | W_hh1]
    s0, s1 = torch.cuda.current_stream(dev), kn.side_stream(dev)
    s1.wait_stream(s0)
    for t in range(S):
        kn.gemm(zbuf[t][:, :Hd], whh0, zbuf[t + 1][:, :Hd], B, Hd, Hd, 2 * Hd, Hd, 2 * Hd, bias=b_hh0, add=pre0[t], ld_add=Hd, relu=True)
        ev = torch.cuda.Event()
        ev.record(s0)
        with torch.cuda.stream(s1):
            s1.wait_event(ev)
            kn.gemm(zbuf[t + 1], w1cat, zbuf[t + 2][:, Hd:], B, Hd, 2 * Hd, 2 * Hd, 2 * Hd, 2 * Hd, bias=b_ih1, add=b_hh1, ld_add=0,
                    relu=True)
    s0.wait_stream(s1)
    return zbuf, plan, emb_t, goal, None, meta


@torch.no_grad()
def decoder_rnn_infer(plan, emb, goal, lo: int, hi: int, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, h0=None):
    """Inference forward of the decoder RNN (logistic_decoder_rnn.py:257-271 with h_0): -> (h1 (B,S,H), h_n (2,B,H))."""
    zbuf, _, _, _, _, meta = _decoder_rnn_forward(plan, emb, goal, lo, hi, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, h0)
    S, Hd = meta[1], meta[2]
    h1 = zbuf[2:S + 2, :, Hd:].permute(1, 0, 2).contiguous()
    h_n = torch.stack([zbuf[S][:, :Hd], zbuf[S + 1][:, Hd:]]).float()             # final states of layer 0 and layer 1
    return h1, h_n


@_scoped
class DecoderRNNFn(torch.autograd.Function):
    """2-layer ReLU RNN over x_t = [plan | emb_t[lo:hi] | goal], h_{-1} = 0  ->  h1 (B, S, H).

    Restructured for a latency-bound recurrence on 256 CUs (same sums, different association; fp32-parity-checked):
      * layer 0's input projection is split by linearity: plan/goal columns of W_ih0 once per sequence, the embedding
        columns per token (one batched GEMM);
      * layer 1's input projection is fused into its recurrent GEMM: h1_t = relu([h0_t | h1_{t-1}] [W_ih1 | W_hh1]^T + b)
        (K = 2H), so layer 1 needs nothing but h0_t; the two layers can run as a wavefront on two HIP streams
        (HULC_WAVEFRONT=1) — measured neutral-to-slower on MI355X because each launch already fills the chip, so the
        default is one stream;
      * activations are time-major (S, B, .) so each step's rows are contiguous; zbuf[t+1] = [h0_t | h1_{t-1}].
    Backward mirrors it: delta0_t = ([delta1_t | delta0_{t+1}] [W_ih1^T | W_hh0^T]^T) * (h0_t > 0), again a wavefront.
    """

    @staticmethod
    def forward(ctx, plan, emb, goal, lo: int, hi: int, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1, b_ih1, b_hh1, time_major_out: bool = False,
                emb_tm: bool = False):
        """time_major_out: return h1 as the (S*B, H) column slice of the state buffer it already lies in (row = step * B + batch row, row
        stride 2H) instead of a (B, S, H) copy; the incoming gradient then has the same time-major row order.
        emb_tm: `emb` is the (S, B, hi - lo) time-major slice itself; its gradient comes back in the same layout."""
        zbuf, plan, emb_t, goal, z16, meta = _decoder_rnn_forward(plan, emb, goal, lo, hi, w_ih0, w_hh0, b_ih0, b_hh0, w_ih1, w_hh1,
                                                                 b_ih1, b_hh1, None, emb_tm)
        B, S, Hd = meta[0], meta[1], meta[2]
        ctx.time_major = bool(time_major_out)
        ctx.persistent = z16 is not None
        ctx.z16 = z16                                      # bf16 mirror of zbuf written by the persistent kernel (weight-gradient operand)
        ctx.save_for_backward(plan, emb_t, goal, zbuf, w_ih0, w_hh0, w_ih1, w_hh1)
        ctx.biases = (b_ih0, b_hh0, b_ih1, b_hh1)          # only their identity is needed (gradient sinks)
        ctx.meta = meta
        if time_major_out:
            return zbuf[2:S + 2].view(S * B, 2 * Hd)[:, Hd:]                       # (S*B, H) view, row stride 2H
        return zbuf[2:S + 2, :, Hd:].permute(1, 0, 2).contiguous()                 # (B, S, H) for the heads

    @staticmethod
    def backward(ctx, dH1):
        plan, emb_t, goal, zbuf, w_ih0, w_hh0, w_ih1, w_hh1 = ctx.saved_tensors
        B, S, Hd, P, G, E, lo, hi, Etot, emb_tm = ctx.meta
        dev = dH1.device
        Kin = w_ih0.shape[1]
        f32 = dict(dtype=torch.float32, device=dev)
        dH1_t = _c(dH1).view(S, B, Hd) if ctx.time_major else dH1.permute(1, 0, 2).contiguous()     # (S, B, H) time-major
        if ctx.persistent:                                                         # the kernel writes rows S..0; row S+1 is its zero start
            dbuf = torch.empty(S + 2, B, 2 * Hd, dtype=zbuf.dtype, device=dev)     # edges (row S+1, first half of row 0) cleared by the launcher
        else:
            dbuf = torch.zeros(S + 2, B, 2 * Hd, dtype=zbuf.dtype, device=dev)     # dbuf[t+1][:, :H] = delta1_t, dbuf[t][:, H:] = delta0_t
        if ctx.persistent:
            # reversed sweep: wave step s reads dbuf[S+1-s] (row S+1 = 0) and writes dbuf[S-s]; weights read transposed in place
            d16, d16t = kn.rnn_wavefront(dbuf[S + 1], -B * 2 * Hd, S, B, Hd, weight_operand(w_hh1), weight_operand(w_ih1), weight_operand(w_hh0), True,
                                   add1=dH1_t[S - 1], add1_step=-B * Hd, ld_add1=Hd,
                                   mask1=zbuf[S + 1][:, Hd:], mask1_step=-B * 2 * Hd, ld_mask1=2 * Hd,
                                   mask2=zbuf[S + 1][:, :Hd], mask2_step=-B * 2 * Hd, ld_mask2=2 * Hd,
                                   mirror_t=bool(os.environ.get("HULC_RNN_WGRAD_TMIRROR")), zero_edges=True)
        else:
            whh1_t = weight_operand(w_hh1, "t")
            wb0 = weight_operand(torch.cat([w_ih1.detach().t(), w_hh0.detach().t()], dim=1))   # (H, 2H) = [W_ih1^T | W_hh0^T]
            s0, s1 = torch.cuda.current_stream(dev), kn.side_stream(dev)
            s1.wait_stream(s0)
        for t in (range(S - 1, -1, -1) if not ctx.persistent else ()):
            h1_t, h0_t = zbuf[t + 2][:, Hd:], zbuf[t + 1][:, :Hd]
            # delta1_t = (dH1_t + delta1_{t+1} W_hh1) * (h1_t > 0)      (dbuf[S+1] does not exist: delta1_S = 0 -> zero rows of zbuf[0])
            prev = dbuf[t + 2][:, :Hd] if t + 2 <= S else zbuf[0][:, :Hd]
            kn.gemm(prev, whh1_t, dbuf[t + 1][:, :Hd], B, Hd, Hd, 2 * Hd, Hd, 2 * Hd, add=dH1_t[t], ld_add=Hd, mask=h1_t, ld_mask=2 * Hd)
            ev = torch.cuda.Event()
            ev.record(s0)
            with torch.cuda.stream(s1):
                s1.wait_event(ev)
                # delta0_t = ([delta1_t | delta0_{t+1}] [W_ih1^T | W_hh0^T]^T) * (h0_t > 0)
                kn.gemm(dbuf[t + 1], wb0, dbuf[t][:, Hd:], B, Hd, 2 * Hd, 2 * Hd, 2 * Hd, 2 * Hd, mask=h0_t, ld_mask=2 * Hd)
        if not ctx.persistent:
            s0.wait_stream(s1)
        d1 = dbuf[1:S + 1]            # rows (t, b): [delta1_t | delta0_{t+1}]
        d0 = dbuf[0:S][:, :, Hd:]     # rows (t, b): delta0_t   (strided view, ld 2H)
        # operands of the big weight-gradient GEMMs: the bf16 mirrors the persistent kernels left behind (half the bytes; the MFMA
        # rounds to bf16 while staging anyway), else the fp32 buffers
        use16 = ctx.persistent and not os.environ.get("HULC_RNN_WGRAD_FP32")
        z16, z16t = ctx.z16 if ctx.persistent else (None, None)
        zw = z16 if use16 else zbuf
        d1w = d16[1:S + 1] if use16 else d1
        d0w = d16[0:S][:, :, Hd:] if use16 else d0
        M = S * B
        fuse_b = kn.gemm_fuses_rowsum(Hd, False)
        # transposed mirrors (HULC_RNN_WGRAD_TMIRROR=1: feature-major, k-major GEMM operands) were the faster operands while row-major
        # tiles needed an in-register transpose (61 vs 99 us per GEMM); with the transpose-read tiles both layouts run 66-70 us and the
        # default is the row-major mirror alone (no extra stores in the recurrent kernel)
        use_t = use16 and z16t is not None and d16t is not None
        ldt = (S + 2) * B

        def wgrad_t(d_feat0, d_tok0, z_feat0, z_tok0, ncols, param, bias):
            """the same as wgrad() on the transposed mirrors: rows of dT = delta features, rows of zT = input features, k = S*B tokens"""
            sink, bsink = gradsink.get(param), gradsink.get(bias)
            out = sink if sink is not None else torch.empty(Hd, ncols, **f32)
            bout = bsink if bsink is not None else torch.empty(Hd, **f32)
            kn.gemm(d16t[d_feat0:, d_tok0 * B:], z16t[z_feat0:, z_tok0 * B:], out, Hd, ncols, M, ldt, ldt, ncols, a_kmajor=True, b_kmajor=True,
                    accumulate=sink is not None and not gradsink.first_write(param), rowsum=bout,
                    rowsum_accumulate=bsink is not None and not gradsink.first_write(bias))
            return (None if sink is not None else out), (None if bsink is not None else bout)

        def wgrad(dlt, inp_rows, ncols, param, bias):
            """param.grad (+)= dlt^T inp_rows, bias.grad (+)= column sums of dlt (the row sums of the GEMM's A operand, fused into
            the same launch); straight into the gradient arena when the trainer registered sinks"""
            sink, bsink = gradsink.get(param), gradsink.get(bias)
            out = sink if sink is not None else torch.empty(Hd, ncols, **f32)
            bout = bsink if bsink is not None else torch.empty(Hd, **f32)
            acc_b = bsink is not None and not gradsink.first_write(bias)
            kn.gemm(dlt, inp_rows, out, Hd, ncols, M, 2 * Hd, 2 * Hd, ncols, a_kmajor=False, b_kmajor=False,
                    accumulate=sink is not None and not gradsink.first_write(param), rowsum=bout if fuse_b else None, rowsum_accumulate=acc_b)
            if not fuse_b:
                kn.colsum(dlt, M, Hd, 2 * Hd, bout, accumulate=acc_b)
            return (None if sink is not None else out), (None if bsink is not None else bout)

        b_ih0, b_hh0, b_ih1, b_hh1 = ctx.biases
        # round 6, OPT-IN (HULC_WGRAD_FORK=1, native trainer only): the three 2048^3 recurrent weight gradients feed nothing downstream (they land in
        # the gradient arena) while the chain behind them — sequence sums, skinny data-gradient GEMMs, KL / sample backward: ~0.1 ms of launches that
        # leave the chip nearly idle — is the critical path: as a branch, joined by the end-of-pass weight-gradient launch.  Measured 3.177 -> 3.131 ms
        # on one box and nothing on another (3.13 either way); on a THIRD cached stream of its own the branch made a later step-node capture's
        # hipGraphLaunch segfault in every run of tests/test_trainer_gpu.py + tests/test_stepnode_gpu.py in that order (0 of 6 runs without it, 0 of 3
        # on the encoder stream): off by default.
        wg_fork = (kn.fork_branches() and kn.wgrad_branch_ok() and dev.type == "cuda" and ctx.persistent and not use_t
                   and all(gradsink.get(t_) is not None for t_ in (w_ih1, b_ih1, w_hh1, b_hh1, w_hh0, b_hh0)))
        if wg_fork:
            # (on the gripper camera's encoder stream: idle between the encoders' forward and their backward at the end of the pass)
            from .models.perceptual_encoders.concat_encoders import _encoder_side_stream
            wg_s = _encoder_side_stream(dev)
            cur_s = torch.cuda.current_stream(dev)
            wg_s.wait_stream(cur_s)
            with torch.cuda.stream(wg_s):
                wgrad(d1w, zw[1:S + 1], Hd, w_ih1, b_ih1)
                wgrad(d1w, zw[1:S + 1][:, :, Hd:], Hd, w_hh1, b_hh1)
                wgrad(d0w, zw[0:S], Hd, w_hh0, b_hh0)
            for t_ in (d1w, zw, d0w):
                t_.record_stream(wg_s)
            kn.note_producer_stream(dev, wg_s)
            dw_ih1 = db_ih1 = dw_hh1 = db_hh1 = dw_hh0 = db_hh0 = None
        # layer 1: dW_ih1 = delta1^T h0_t, dW_hh1 = delta1^T h1_{t-1}   (zbuf[t+1] = [h0_t | h1_{t-1}]); both biases see delta1
        elif use_t:
            dw_ih1, db_ih1 = wgrad_t(0, 1, 0, 1, Hd, w_ih1, b_ih1)                  # delta1 = d[1:S+1][:, :, :H], h0_t = z[1:S+1][:, :, :H]
            dw_hh1, db_hh1 = wgrad_t(0, 1, Hd, 1, Hd, w_hh1, b_hh1)                 # h1_{t-1} = z[1:S+1][:, :, H:]
            dw_hh0, db_hh0 = wgrad_t(Hd, 0, 0, 0, Hd, w_hh0, b_hh0)                 # delta0 = d[0:S][:, :, H:], h0_{t-1} = z[0:S][:, :, :H]
        else:
            dw_ih1, db_ih1 = wgrad(d1w, zw[1:S + 1], Hd, w_ih1, b_ih1)
            dw_hh1, db_hh1 = wgrad(d1w, zw[1:S + 1][:, :, Hd:], Hd, w_hh1, b_hh1)
            # layer 0 (b_ih0's gradient rides on the embedding-column GEMM of dW_ih0 below)
            dw_hh0, db_hh0 = wgrad(d0w, zw[0:S], Hd, w_hh0, b_hh0)                  # h0_{t-1} = zbuf[t][:, :H]
        dcs = torch.empty(B, 2 * Hd, **f32)
        dc = dcs[:, Hd:]                                                            # (B, H) strided view, ld 2H
        kn.strided_seq_sum(d0, dc, B, S, Hd, 2 * Hd, B * 2 * Hd, 2 * Hd)            # dc = sum_t delta0_t
        wih0 = weight_operand(w_ih0)
        s_ih0 = gradsink.get(w_ih0)
        dw_ih0 = s_ih0 if s_ih0 is not None else torch.empty(Hd, Kin, **f32)
        acc0 = s_ih0 is not None and not gradsink.first_write(w_ih0)     # three GEMMs, each the only writer of its column slice
        kn.wgrad(dc, plan, dw_ih0, Hd, P, B, 2 * Hd, P, Kin, accumulate=acc0, defer=s_ih0 is not None)
        sb_ih0 = gradsink.get(b_ih0)
        db_ih0 = sb_ih0 if sb_ih0 is not None else torch.empty(Hd, **f32)
        acc_b0 = sb_ih0 is not None and not gradsink.first_write(b_ih0)
        if kn.wgrad_group_ok(d0w, emb_t, dw_ih0[:, P:P + E], Hd, E, M, 2 * Hd, E, Kin):
            kn.wgrad(d0w, emb_t, dw_ih0[:, P:P + E], Hd, E, M, 2 * Hd, E, Kin, accumulate=acc0, rowsum=db_ih0, rowsum_accumulate=acc_b0,
                     defer=s_ih0 is not None and sb_ih0 is not None)
        else:
            kn.gemm(d0w, emb_t, dw_ih0[:, P:P + E], Hd, E, M, 2 * Hd, E, Kin, a_kmajor=False, b_kmajor=False, accumulate=acc0,
                    rowsum=db_ih0 if fuse_b else None, rowsum_accumulate=acc_b0)
            if not fuse_b:
                kn.colsum(d0, M, Hd, 2 * Hd, db_ih0, accumulate=acc_b0)
        if sb_ih0 is not None:
            db_ih0 = None
        kn.wgrad(dc, goal, dw_ih0[:, P + E:], Hd, G, B, 2 * Hd, G, Kin, accumulate=acc0, defer=s_ih0 is not None)
        if s_ih0 is not None:                 # written straight into the gradient arena
            dw_ih0 = None
        wih0_t = weight_operand(w_ih0, "t")                                         # (Kin, H): rows = input features, k-major
        dplan = torch.empty(B, P, **f32)
        kn.gemm(dc, wih0_t, dplan, B, P, Hd, 2 * Hd, Hd, P)
        dgoal = torch.empty(B, G, **f32)
        kn.gemm(dc, wih0_t[P + E:], dgoal, B, G, Hd, 2 * Hd, Hd, G)
        demb_t = torch.empty(S, B, E, **f32)
        kn.gemm(d0, wih0_t[P:P + E], demb_t, M, E, Hd, 2 * Hd, Hd, E)
        if emb_tm:
            demb = demb_t
        else:
            demb = torch.zeros(B, S, Etot, **f32)
            demb[:, :, lo:hi] = demb_t.permute(1, 0, 2)
        return (dplan, demb, dgoal, None, None, dw_ih0, dw_hh0, db_ih0, db_hh0, dw_ih1, dw_hh1, db_ih1, db_hh1, None, None)


@_scoped
class EmbFanoutFn(torch.autograd.Function):
    """emb (N, S, D) -> (emb[:, 0], emb[:n_last, -1], emb, emb[:, :, lo:hi] time-major (S, N, hi-lo)): the four views Hulc2.training_step
    hands to the prior, the visual goal encoder, the posterior and the action decoder (hulc2.py:380-387).  One gather launch forward,
    one merge launch backward (csrc/pointwise.hip: emb_fanout_fwd / emb_fanin_bwd)."""

    @staticmethod
    def forward(ctx, emb, n_last: int, lo: int, hi: int):
        emb = _c(emb)
        N, S, D = emb.shape
        e0, elast = _f32(N, D, like=emb), _f32(max(n_last, 1), D, like=emb)
        edec = _f32(S, N, hi - lo, like=emb)
        kn.emb_fanout_fwd(emb, N, S, D, n_last, lo, hi, e0, elast if n_last > 0 else None, edec)
        ctx.meta = (N, S, D, n_last, lo, hi)
        return e0, elast[:n_last], emb.view_as(emb), edec

    @staticmethod
    def backward(ctx, g0, glast, grec, gdec):
        N, S, D, n_last, lo, hi = ctx.meta
        like = next(g for g in (g0, glast, grec, gdec) if g is not None)
        demb = _f32(N, S, D, like=like)
        c = lambda t: None if t is None else _c(t)
        kn.emb_fanin_bwd(c(grec), c(g0), c(glast) if n_last > 0 else None, c(gdec), N, S, D, n_last, lo, hi, demb)
        return demb, None, None, None


@_scoped
class LossCombineFn(torch.autograd.Function):
    """(kl_loss[m], action_loss[m], clip) -> total (0-dim), logs (3 + n) = {kl mean, action mean, beta * clip, per-modality totals}; hulc2.py:400-430.
    Only `total` carries a gradient."""

    @staticmethod
    def forward(ctx, kls, acts, clip, beta: float):
        kls, acts = _c(kls), _c(acts)
        n = kls.numel()
        out = _f32(4 + n, like=kls)
        kn.loss_combine_fwd(kls, acts, None if clip is None else clip.reshape(1), n, beta, out)
        ctx.meta = (n, beta, clip is not None)
        ctx.set_materialize_grads(False)                   # (no zero tensor for the logged values' absent gradient)
        total, logs = out[0], out[1:]                      # two outputs (views of one buffer made here): no select-backward nodes later
        ctx.mark_non_differentiable(logs)
        return total, logs

    @staticmethod
    def backward(ctx, g, _glogs):
        n, beta, has_clip = ctx.meta
        g = _c(g.reshape(1))
        dk, da = _f32(n, like=g), _f32(n, like=g)
        dc = _f32(1, like=g) if has_clip else None
        kn.loss_combine_bwd(g, n, beta, dk, da, dc)          # g[0] = d total (the other outputs are logged values, detached by the caller)
        return dk, da, (dc.reshape(()) if has_clip else None), None


# ------------------------------------------------------------------------------------------------
# losses
# ------------------------------------------------------------------------------------------------
@_scoped
class MixLossFn(torch.autograd.Function):
    """y (T, 3*A*n_mix + 2 [+pad]) head outputs, actions (T, A+1) -> (nseg,) losses NLL + alpha * gripper CE, each the
    mean over its own T/nseg tokens (nseg = 1: the reference's single mean; nseg = 2: vis and lang batched together)."""

    @staticmethod
    def forward(ctx, y, act, act_min, act_max, n_mix: int, num_classes: int, log_scale_min: float, gripper_alpha: float, nseg: int = 1,
                time_major_B: int = 0):
        """time_major_B > 0: the rows of y / act are time-major (row = step * B + batch row, the order the recurrent kernel produces)"""
        y, act = _c(y), _c(act)
        T, A = act.shape[0], act.shape[1] - 1
        out = _f32(3, nseg, like=y)              # planar: totals | nll means | ce means
        cfg = (T, A, n_mix, num_classes, y.stride(0), log_scale_min, gripper_alpha)
        kn.mix_loss_fwd(y, act, out, *cfg, act_min, act_max, nseg=nseg, time_major_B=time_major_B)
        ctx.save_for_backward(y, act, act_min, act_max)
        ctx.cfg, ctx.nseg, ctx.tmb = cfg, nseg, time_major_B
        return out[0]

    @staticmethod
    def backward(ctx, g):
        y, act, act_min, act_max = ctx.saved_tensors
        dy = torch.empty_like(y)           # the kernel writes every column (pad columns beyond 3*A*n_mix + 2 as zeros)
        kn.mix_loss_bwd(y, act, _c(g.reshape(ctx.nseg)), dy, dy.stride(0), *ctx.cfg, act_min, act_max, nseg=ctx.nseg, time_major_B=ctx.tmb)
        return dy, None, None, None, None, None, None, None, None, None


@_scoped
class CatKLFn(torch.autograd.Function):
    """KL balancing of hulc2.py:444-466 on (B, G*32) logits of prior (pp) and posterior (pr)."""

    @staticmethod
    def forward(ctx, pp, pr, G: int, CLS: int, beta: float, mix: float, nseg: int = 1):
        """nseg > 1: the rows are nseg equal segments (modalities batched together) -> (nseg,) losses, each its own mean"""
        pp, pr = _c(pp), _c(pr)
        B = pp.shape[0]
        out = _f32(nseg, like=pp)
        klg = _f32(B * G, like=pp)
        kn.cat_kl_fwd(pp, pr, B, G, CLS, beta, out, klg, nseg)
        ctx.save_for_backward(pp, pr, klg)
        ctx.meta = (B, G, CLS, beta, mix, nseg)
        return out[0] if nseg == 1 else out

    @staticmethod
    def backward(ctx, g):
        pp, pr, klg = ctx.saved_tensors
        B, G, CLS, beta, mix, nseg = ctx.meta
        dpp, dpr = torch.empty_like(pp), torch.empty_like(pr)
        kn.cat_kl_bwd(pp, pr, klg, B, G, CLS, beta, mix, _c(g.reshape(nseg)), dpp, dpr, nseg)
        return dpp, dpr, None, None, None, None, None


@_scoped
class PlanSampleFn(torch.autograd.Function):
    """Straight-through one-hot sample; returns (plan (B, G*CLS), idx (B, G))."""

    @staticmethod
    def forward(ctx, logits, idx_in, G: int, CLS: int, seed: int):
        logits = _c(logits)
        B = logits.shape[0]
        plan = _f32(B, G * CLS, like=logits)
        idx = torch.empty(B, G, dtype=torch.long, device=logits.device)
        kn.plan_sample_fwd(logits, _c(idx_in) if idx_in is not None else None, seed, B * G, CLS, idx, plan)
        ctx.save_for_backward(logits)
        ctx.meta = (B, G, CLS)
        ctx.mark_non_differentiable(idx)
        ctx.set_materialize_grads(False)
        return plan, idx

    @staticmethod
    def backward(ctx, dplan, _didx=None):
        if dplan is None:
            return None, None, None, None, None
        (logits,) = ctx.saved_tensors
        B, G, CLS = ctx.meta
        dl = torch.empty_like(logits)
        kn.plan_sample_bwd(logits, _c(dplan), B * G, CLS, dl)
        return dl, None, None, None, None


@_scoped
class ClipLossFn(torch.autograd.Function):
    """hulc2.py:472-508 on projected features im / tx (M, 32): -> (loss, number of rows taking part).  row0: rows below it never take part and
    `use` describes rows row0 .. M-1 (the stacked modalities of a step: only the language rows enter the loss)."""

    @staticmethod
    def forward(ctx, im, tx, use, logit_scale, row0: int = 0):
        im, tx = _c(im), _c(tx)
        use_u8 = _c(use).view(torch.uint8) if use.dtype == torch.bool else use.to(torch.uint8)     # (a bool tensor IS bytes of 0 / 1: no copy)
        M, D = im.shape
        out = _f32(2, like=im)
        ls = logit_scale.reshape(1)
        kn.clip_loss_fwd(im, tx, use_u8, ls, M, D, out, row0)
        ctx.save_for_backward(im, tx, use_u8, ls)
        ctx.row0, ctx.scale_param = int(row0), logit_scale
        loss, count = out[0], out[1]                       # count = rows with use != 0 (1 when none): batch_size["aux_lang"], hulc2.py:391-394
        ctx.mark_non_differentiable(count)
        ctx.set_materialize_grads(False)
        return loss, count

    @staticmethod
    def backward(ctx, g, _gcount=None):
        if g is None:
            return None, None, None, None, None
        im, tx, use_u8, ls = ctx.saved_tensors
        M, D = im.shape
        dim, dtx = torch.empty_like(im), torch.empty_like(tx)
        sink = gradsink.get(ctx.scale_param)
        if sink is not None and gradsink.first_write(ctx.scale_param):
            dscale, ret = sink.reshape(1), None            # the step's only writer: straight into the gradient arena (no AccumulateGrad add)
        else:
            dscale = _f32(1, like=im)
            ret = dscale.reshape(())
        kn.clip_loss_bwd(im, tx, use_u8, ls, M, D, _c(g.reshape(1)), dim, dtx, dscale, ctx.row0)
        return dim, dtx, None, ret, None


@torch.no_grad()
def actions_time_major(acts, obss, to_tcp: bool) -> torch.Tensor:
    """per-modality actions (B_i, S, 7) [+ robot_obs (B_i, S, obs_dim)] of equal shapes -> (S * sum B_i, 7) in the recurrent kernel's time-major
    row order, world -> tcp frame applied on the way when to_tcp (gripper_control.py:16-36; no gradient: actions are data)"""
    acts = [_c(a.float()) for a in acts]
    obss = [_c(o.float()) for o in obss]
    Bm, S, _ = acts[0].shape
    if len(acts) > 4 or any(a.shape != acts[0].shape for a in acts):
        a, o = torch.cat(acts, 0), torch.cat(obss, 0)
        a = world_to_tcp_frame(a, o) if to_tcp else a
        return a.transpose(0, 1).contiguous().reshape(-1, a.shape[-1])
    out = torch.empty(S * Bm * len(acts), 7, dtype=torch.float32, device=acts[0].device)
    kn.actions_time_major(acts, obss, Bm, S, obss[0].shape[-1], to_tcp, out)
    return out


def world_to_tcp_frame(actions, robot_obs):
    """gripper_control.py:16-36 (no gradient: actions are data)."""
    B, S, _ = actions.shape
    out = torch.empty_like(actions)
    kn.world_to_tcp(_c(actions.float()), _c(robot_obs.float()), B * S, robot_obs.shape[-1], out)
    return out


@torch.no_grad()
def tcp_to_world_frame(actions: torch.Tensor, robot_obs: torch.Tensor) -> torch.Tensor:
    """gripper_control.py:39-63 (sampled actions back to the world frame; no gradient, fp32 forced like the reference)."""
    B, S, _ = actions.shape
    out = torch.empty(B, S, 7, dtype=torch.float32, device=actions.device)
    kn.tcp_to_world(_c(actions.float()), _c(robot_obs.float()), B * S, robot_obs.shape[-1], out)
    return out


@torch.no_grad()
def mix_sample(y: torch.Tensor, A: int, n_mix: int, log_scale_min: float, gripper_bounds: torch.Tensor, seed: int,
               u_mix: Optional[torch.Tensor] = None, u_inv: Optional[torch.Tensor] = None, return_idx: bool = False):
    """LogisticDecoderRNN._sample on the fused head output y (T, >= 3*A*n_mix + 2) -> actions (T, A+1) [, mixture idx (T, A)]."""
    y = _c(y)
    T = y.shape[0]
    act = torch.empty(T, A + 1, dtype=torch.float32, device=y.device)
    idx = torch.empty(T, A, dtype=torch.int64, device=y.device) if return_idx else None
    kn.mix_sample(y, y.stride(0), T, A, n_mix, log_scale_min, gripper_bounds, act, seed,
                  None if u_mix is None else _c(u_mix.float()), None if u_inv is None else _c(u_inv.float()), idx)
    return (act, idx) if return_idx else act
