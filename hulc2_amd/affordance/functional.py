"""The language-fused U-Net decoder, segmentation head and pixel cross-entropy of the affordance model as ONE autograd Function over
padded-grid kernels (csrc/gridconv.hip, csrc/affordance.hip; weight gradients through the grouped launch of csrc/wgrad_group.hip).

Reference: UnetLangFusionDecoder.forward / DecoderBlock.forward (hulc2/affordance/models/core/unet_decoder.py:60-80,131-146), R3M.forward
(models/visual_lang_encoders/r3m_rn18.py:78-94), AffDepthLangFusionPixel.forward (models/lang_fusion/aff_lang_depth_pixel.py:98-129),
cross_entropy_with_logits as PixelAffLangDetector.criterion applies it (pixel_aff_lang_detector.py:122-145, utils/losses.py:6-13)."""
from typing import List, Sequence

import torch

from .. import gradsink, kernels as kn
from ..shadow import weight_operand

DECODER_CHANNELS = (512, 256, 128, 64, 32)                 # r3m_rn18.py:54
ENCODER_CHANNELS = (3, 64, 64, 128, 256, 512)              # r3m_rn18.py:59


def block_channels():
    """(in, skip, out) channels of the five DecoderBlocks (unet_decoder.py:104-116)"""
    enc = list(ENCODER_CHANNELS[1:])[::-1]
    return list(zip([enc[0]] + list(DECODER_CHANNELS[:-1]), enc[1:] + [0], DECODER_CHANNELS))


def _fwd_w(w):          # (Cout, Cin, 3, 3) fp32 -> bf16 [Cout][9 Cin], k = tap * Cin + ci: the trainer's per-step layout shadow when there is one
    return weight_operand(w, "ohwi")


def _dgrad_w(w, cin_keep=None):      # -> bf16 [Cin'][9 Cout] = the UNflipped (ci, kh, kw, co) shadow (gridconv flips the taps); first cin_keep rows
    m = weight_operand(w, "ihwo").reshape(w.shape[1], -1)
    return m if cin_keep is None else m[:cin_keep]


def _plain_strides(t):   # (N, H, W, C) NHWC map, dense or a grid tensor's pixel view (channels contiguous) -> (tensor, (sn, sy, sx))
    if t.stride(3) != 1:
        raise ValueError("feature maps must be channels-last with contiguous channels")
    return t, tuple(int(v) for v in t.stride()[:3])


def _conv_wgrad(dz: "kn.Grid", x: "kn.Grid", w: torch.Tensor, cin: int):
    """dW (Cout, cin, 3, 3) (+)= nine products dZ^T X[. + off_t] over the grid rows, written straight into the OIHW layout (col_mul = 9):
    into the trainer's gradient arena at the end of the pass when a sink is registered, else into a fresh tensor handed to autograd"""
    cout = dz.C
    sink = gradsink.get(w)
    out = sink.view(cout, cin * 9) if sink is not None else torch.empty(cout, cin * 9, dtype=torch.float32, device=dz.rows.device)
    acc = sink is not None and not gradsink.first_write(w)
    A = dz.rows[dz.guard:dz.guard + dz.Rpad]
    B = x.rows[x.guard:x.guard + x.Rpad]                               # tap u reads it shifted by (dy (W + 2) + dx) rows: inside the zero guards
    kn.wgrad(A, B, out, cout, cin, x.Rpad, cout, x.C, cin * 9, accumulate=acc, defer=sink is not None, col_mul=9, conv_taps_wp=x.W + 2)
    return None if sink is not None else out.view(cout, cin, 3, 3)


def _vec_grad(param, like):
    """(destination, accumulate flag, tensor to return to autograd or None) of a 1-D parameter gradient"""
    sink = gradsink.get(param)
    if sink is None:
        t = torch.empty_like(param, dtype=torch.float32)                  # (same shape as the parameter: (1, D) weights included)
        return t, False, t
    return sink, not gradsink.first_write(param), None


def _eval_bn(gamma, beta, run_mean, run_var, eps: float = 1e-5):
    """the (mean, rstd, scale, shift) table of hulc_grid_bn_relu_fwd from an nn.BatchNorm2d's running statistics (inference mode)"""
    rstd = torch.rsqrt(run_var + eps)
    scale = gamma.detach() * rstd
    return torch.stack([run_mean, rstd, scale, beta.detach() - run_mean * scale]).contiguous()


class AffDecoderLossFn(torch.autograd.Function):
    """(p0, out_hw, running-stat buffers, g0, g1, g2, stem, l1, l2, l3, l4, 30 block parameters, head weight, head bias) -> (aff_loss, logits)

    g_i (N, C_in_i) fp32 = lang_proj_i(l_enc) for the three language-fused blocks (computed outside: its Linear is an ordinary MLP layer);
    the trunk maps are NHWC bf16; they receive a gradient only when they ask for one (trunk_mode "reference": the trainable stem behind them,
    affordance/trunk.py) — the skip connections' share is the data gradient of a block's first convolution over its skip channels, the
    last map's the first block's upsampled input.  BatchNorm runs on batch statistics and updates the running buffers."""

    @staticmethod
    def forward(ctx, p0, out_hw, buffers, g0, g1, g2, f_stem, f1, f2, f3, f4, *params):
        """buffers: the ten BatchNorms' [running_mean, running_var] * 10 — updated from the batch statistics; ("eval", buffers): inference mode,
        normalisation BY the running statistics (no update)"""
        if kn.get_compute() != "bf16":
            raise NotImplementedError("the affordance decoder is built for the bf16 compute mode")
        blocks, head_w, head_b = params[:30], params[30], params[31]
        evalm = isinstance(buffers, tuple) and buffers[0] == "eval"
        if evalm:
            buffers = buffers[1]
        N = f4.shape[0]
        dev = f4.device
        gs = (g0, g1, g2)
        skips = (f3, f2, f1, f_stem, None)
        chans = block_channels()
        saved = []
        x_map, x_str = _plain_strides(f4)
        hi = f4.shape[1]
        for i, (cin, cs, cout) in enumerate(chans):
            w1, ga1, be1, w2, ga2, be2 = blocks[6 * i:6 * i + 6]
            skip = skips[i]
            ho = skip.shape[1] if skip is not None else out_hw
            s = ho // hi
            g = gs[i].contiguous() if i < 3 else None
            sk, ss = _plain_strides(skip) if skip is not None else (None, None)
            X = kn.grid_upcat_fwd(x_map, x_str, g, sk, ss, N, ho, ho, s, cin, cs)
            Y1, st = kn.gridconv3x3(X, _fwd_w(w1), cout, want_stats=not evalm)
            bn1 = _eval_bn(ga1, be1, *buffers[4 * i:4 * i + 2]) if evalm else \
                kn.grid_bn_finalize(st, N, ho, ho, cout, ga1, be1, *(buffers[4 * i:4 * i + 2] if buffers else (None, None)))
            O1 = kn.grid_bn_relu_fwd(Y1, bn1)
            Y2, st = kn.gridconv3x3(O1, _fwd_w(w2), cout, want_stats=not evalm)
            bn2 = _eval_bn(ga2, be2, *buffers[4 * i + 2:4 * i + 4]) if evalm else \
                kn.grid_bn_finalize(st, N, ho, ho, cout, ga2, be2, *(buffers[4 * i + 2:4 * i + 4] if buffers else (None, None)))
            O2 = kn.grid_bn_relu_fwd(Y2, bn2)
            saved.append((X, Y1, bn1, O1, Y2, bn2, O2, x_map, x_str, g, hi, s))
            x_map, sn, sy, sx = O2.pixel_strides()
            x_str = (sn, sy, sx)
            hi = ho
        last = saved[-1][6]
        logit0 = kn.head_conv_fwd(last, head_w.detach().contiguous(), head_b.detach())      # the one-channel head: a streaming kernel, no matrix cores
        p0i = p0.to(device=dev, dtype=torch.int32).contiguous()
        lse, picked = kn.pixel_ce_fwd(logit0, p0i, N, out_hw, out_hw)
        loss = -(picked - lse).sum() / float(N * out_hw * out_hw)
        ctx.saved = (saved, logit0, lse, p0i, blocks, head_w, head_b, N, out_hw)
        logits = logit0.view(N, out_hw + 2, out_hw + 2)[:, 1:-1, 1:-1].reshape(N, -1)
        ctx.mark_non_differentiable(logits)
        return loss, logits

    @staticmethod
    def backward(ctx, dloss, _dlogits):
        saved, logit0, lse, p0i, blocks, head_w, head_b, N, out_hw = ctx.saved
        dev = logit0.device
        chans = block_channels()
        grads_blocks: List = [None] * 30
        up = dloss.reshape(1).to(torch.float32).contiguous()
        g = kn.pixel_ce_bwd_rows(logit0, p0i, lse, up, N, out_hw, out_hw)              # (softmax - onehot) / (N H W) per grid row, fp32
        last = saved[-1][6]
        # head: dW[ci][t] = sum_r g[r] x[r + off_t][ci]; the bias gradient is the sum of (softmax - onehot) = 0
        sink_h = gradsink.get(head_w)
        ci = head_w.shape[1]
        dwh = sink_h.view(-1) if sink_h is not None else torch.empty(ci * 9, dtype=torch.float32, device=dev)
        kn.head_conv_wgrad(last, g, dwh, accumulate=sink_h is not None and not gradsink.first_write(head_w))
        d_head_w = None if sink_h is not None else dwh.view(1, ci, 3, 3)
        d_head_b = torch.zeros_like(head_b, dtype=torch.float32)
        sink_hb = gradsink.get(head_b)
        if sink_hb is not None:
            if gradsink.first_write(head_b):
                sink_hb.zero_()
            d_head_b = None
        dO2 = kn.head_conv_dgrad(g, head_w.detach().contiguous(), N, out_hw, out_hw, last.C)
        dgs = [None, None, None]
        want_maps = ctx.needs_input_grad[6:11]                          # (f_stem, f1, f2, f3, f4)
        dmaps = [None] * 5
        skip_of = (3, 2, 1, 0, None)                                    # block i's skip connection is trunk map skip_of[i]
        for i in range(4, -1, -1):
            cin, cs, cout = chans[i]
            X, Y1, bn1, O1, Y2, bn2, O2, x_map, x_str, g, hi, s = saved[i]
            w1, ga1, be1, w2, ga2, be2 = blocks[6 * i:6 * i + 6]
            d_ga2, a1, r1 = _vec_grad(ga2, bn2)
            d_be2, a2, r2 = _vec_grad(be2, bn2)
            DZ2 = kn.grid_bn_relu_bwd(dO2, O2, Y2, bn2, d_ga2, d_be2, accumulate=a1 or a2)
            grads_blocks[6 * i + 4], grads_blocks[6 * i + 5] = r1, r2
            grads_blocks[6 * i + 3] = _conv_wgrad(DZ2, O1, w2, cout)
            dO1, _ = kn.gridconv3x3(DZ2, _dgrad_w(w2), cout, flip=True)
            d_ga1, a1, r1 = _vec_grad(ga1, bn1)
            d_be1, a2, r2 = _vec_grad(be1, bn1)
            DZ1 = kn.grid_bn_relu_bwd(dO1, O1, Y1, bn1, d_ga1, d_be1, accumulate=a1 or a2)
            grads_blocks[6 * i + 1], grads_blocks[6 * i + 2] = r1, r2
            grads_blocks[6 * i] = _conv_wgrad(DZ1, X, w1, cin + cs)
            need_small = i > 0 or want_maps[4]                          # block 0's input is the trunk's last map
            need_dg = i < 3
            if skip_of[i] is not None and want_maps[skip_of[i]]:        # the skip channels' data gradient: the trunk map's share
                dS, _ = kn.gridconv3x3(DZ1, _dgrad_w(w1)[cin:], cs, flip=True)
                dmaps[skip_of[i]] = dS.interior().contiguous()
            if need_small or need_dg:
                dX, _ = kn.gridconv3x3(DZ1, _dgrad_w(w1, cin), cin, flip=True)
                dsmall, dg = kn.grid_upcat_bwd(dX, x_map, x_str, g, N, hi, hi, s, cin, want_dsmall=need_small, want_dg=need_dg)
                if need_dg:
                    dgs[i] = dg
                dO2 = dsmall
                if i == 0 and want_maps[4]:
                    dmaps[4] = dsmall.interior().contiguous()
        return (None, None, None, dgs[0], dgs[1], dgs[2], *dmaps, *grads_blocks, d_head_w, d_head_b)


class DepthNllFn(torch.autograd.Function):
    """(x (B, 256) fp32, depth_mu.weight, depth_mu.bias, depth_sigma.weight, depth_sigma.bias, target (B,)) -> (depth_loss, mu (B, 1), sigma (B, 1)):
    the two one-output heads of DepthEstimationGaussian.forward, sigma = exp(clamp(log_sigma, -20, 2)) (depth_gaussian.py:94-102) and
    nn.GaussianNLLLoss with sigma passed as the variance (depth_gaussian.py:67-69) as one launch per direction (csrc/affordance.hip)"""

    @staticmethod
    def forward(ctx, x, w_mu, b_mu, w_s, b_s, target):
        x = x.float().contiguous()
        t = target.reshape(-1).float().contiguous()
        mu, sigma, ls, loss = kn.depth_nll_fwd(x, w_mu.detach().contiguous(), b_mu.detach(), w_s.detach().contiguous(), b_s.detach(), t)
        ctx.save_for_backward(x, mu, sigma, ls, t)
        ctx.params = (w_mu, b_mu, w_s, b_s)
        ctx.mark_non_differentiable(mu, sigma)
        return loss, mu, sigma

    @staticmethod
    def backward(ctx, dloss, _dmu, _dsigma):
        x, mu, sigma, ls, t = ctx.saved_tensors
        w_mu, b_mu, w_s, b_s = ctx.params
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dst, rets, mask = [], [], 0
        for bit, p in zip((1, 2, 4, 8), ctx.params):
            d, acc, r = _vec_grad(p, None)
            dst.append(d); rets.append(r)
            mask |= bit if acc else 0
        kn.depth_nll_bwd(x, w_mu.detach().contiguous(), w_s.detach().contiguous(), mu, sigma, ls, t, dloss.reshape(1).float().contiguous(), dx,
                         dst[0], dst[1], dst[2], dst[3], mask)
        return (dx, *rets, None)
