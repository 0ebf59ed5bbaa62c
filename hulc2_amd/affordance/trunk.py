"""The affordance model's ResNet-18 trunk with its TRAINABLE STEM, as the reference trains it.

reference: hulc2/affordance/models/visual_lang_encoders/r3m_rn18.py:27-43 — `_load_vision` freezes the parameters of `layer1 .. layer4` only, so
the stem's `conv1.weight`, `bn1.weight`, `bn1.bias` stay trainable, and pixel_aff_lang_detector.py:51-53 leaves Lightning's train mode on: every
BatchNorm2d of the trunk normalises with the statistics of the batch.  The stem's gradient therefore needs the DATA gradient through all sixteen
frozen convolutions and twenty BatchNorms of the ResNet, the max pool's backward and a 7 x 7 weight gradient: this file.

`TrunkStemFn` is ONE autograd node: forward = vision_r3m._trunk_maps_batch_stats with what the backward needs kept (per layer: the fp32
convolution output z, the batch (mean, rstd), the layer's output); backward, given the gradients of the five maps the decoder and the depth
head consumed:
    BasicBlock   g   = dy * (out > 0)                                  hulc_nhwc_bn_train_bwd (also leaves g: the shortcut's share)
                 dz2 = bn2'(g),  do = conv2^T dz2,  dz1 = bn1'(do * (o > 0)),  dx = conv1^T dz1 + shortcut'(g)
    conv^T       a stride-1 convolution with the taps flipped and the channel roles swapped (hulc_conv2d_padded_fwd on a weight repacked once:
                 the layers are frozen), behind a zero insertion for the stride-2 layers (hulc_nhwc_scatter)
    stem         dp = maxpool'(da0)  (first maximum of a window takes the gradient, as nn.MaxPool2d's indices),  dz = bn1'(dp * (p > 0)) with
                 d bn1.weight / d bn1.bias,  d conv1.weight = the 7 x 7 stride-2 weight gradient on the zero-padded NHWC-8 input
                 (hulc_conv2d_bwd_weight).
Gradient maps are stored in the activation type of the arithmetic mode (bf16 / fp32), every reduction has a fixed order."""
from typing import List

import torch

from .. import gradsink, kernels as kn


def _act_dtype():
    return torch.bfloat16 if kn.get_compute() == "bf16" else torch.float32


def _dgrad_weight(conv: torch.nn.Conv2d, wdtype) -> torch.Tensor:
    """(Cout, Cin, KH, KW) -> the OHWI weight of the data-gradient convolution: [Cin][KH * KW * Cout], taps flipped"""
    w = conv.weight.detach().float()
    return w.flip(2, 3).permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous().to(wdtype)


class _Layer:
    __slots__ = ("bn", "z", "saved", "y", "hin", "win", "cin", "cout", "k", "stride", "pad", "oh", "ow", "wd")


class TrunkStemFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, w_stem, g_stem, b_stem, trunk):
        """img (N, 3, H, W) fp32 (transforms applied); the three stem parameters are inputs only so that autograd routes their gradients here;
        trunk: the vision_r3m._TrunkOnly wrapper of the R3M module.  -> the five NHWC maps (stem + max pool, layer1 .. layer4)"""
        f = trunk._raw_trunk()
        net = trunk.r3m.convnet
        adt = _act_dtype()
        wdt = adt
        n, _, h, w = img.shape
        dev = img.device
        frozen = [c.weight for b in net.blocks() for c in (b.conv1, b.conv2) + ((b.downsample[0],) if b.downsample is not None else ())]
        dkey = (wdt, tuple(p._version for p in frozen), tuple(p.data_ptr() for p in frozen))
        if getattr(trunk, "_dgrad_key", None) != dkey:          # the FROZEN layers' data-gradient weights: repacked once (the trainable
                                                                # stem has no data gradient; its forward operand is live, vision_r3m._raw_trunk)
            trunk._dgrad_w = [(_dgrad_weight(b.conv1, wdt), _dgrad_weight(b.conv2, wdt),
                               _dgrad_weight(b.downsample[0], wdt) if b.downsample is not None else None) for b in net.blocks()]
            trunk._dgrad_key = dkey
        layers: List[_Layer] = []

        def conv_bn(a, wt, bn, hh, ww, cin, k, stride, pad, relu, add=None, wd=None):
            cout = wt.shape[0]
            oh, ow = (hh + 2 * pad - k) // stride + 1, (ww + 2 * pad - k) // stride + 1
            L = _Layer()
            L.z = torch.empty((n, oh, ow, cout), dtype=torch.float32, device=dev)
            kn.conv2d_padded_fwd(a, wt, f["zero"][:cout], L.z, n, hh, ww, cin, cout, k, k, stride, pad, relu=False)
            L.y = torch.empty((n, oh, ow, cout), dtype=adt, device=dev)
            L.saved = torch.empty((2, cout), dtype=torch.float32, device=dev)
            kn.nhwc_bn_train_fwd(L.z, n * oh * ow, cout, bn.weight, bn.bias, bn.eps, 0.1 if bn.momentum is None else bn.momentum,
                                 bn.running_mean, bn.running_var, L.y, add=add, relu=relu, saved=L.saved)
            bn.num_batches_tracked += 1
            L.bn, L.hin, L.win, L.cin, L.cout, L.k, L.stride, L.pad, L.oh, L.ow, L.wd = bn, hh, ww, cin, cout, k, stride, pad, oh, ow, wd
            layers.append(L)
            return L.y, oh, ow

        a_in = kn.r3m_normalize(img.contiguous(), torch.empty((n, h, w, 8), dtype=adt, device=dev), (0.0, 0.0, 0.0), (1.0 / 255.0,) * 3)
        p, ph_, pw_ = conv_bn(a_in, f["stem"], net.bn1, h, w, 8, 7, 2, 3, True)
        oh, ow = (ph_ + 2 - 3) // 2 + 1, (pw_ + 2 - 3) // 2 + 1
        a = kn.maxpool_nhwc(p, torch.empty((n, oh, ow, 64), dtype=adt, device=dev), n, ph_, pw_, 64, 3, 2, 1)
        hh, ww, c = oh, ow, 64
        maps = [a]
        blocks = []
        for bi, (blk, (w1, w2, wd, stride), dws) in enumerate(zip(net.blocks(), f["blocks"], trunk._dgrad_w)):
            rec_d = None
            idn = a
            if wd is not None:
                idn = conv_bn(a, wd, blk.downsample[1], hh, ww, c, 1, stride, 0, False, wd=dws[2])[0]
                rec_d = layers[-1]
            o, oh, ow = conv_bn(a, w1, blk.bn1, hh, ww, c, 3, stride, 1, True, wd=dws[0])
            rec1 = layers[-1]
            c = w1.shape[0]
            a, hh, ww = conv_bn(o, w2, blk.bn2, oh, ow, c, 3, 1, 1, True, add=idn, wd=dws[1])
            blocks.append((rec1, layers[-1], rec_d))
            if bi % 2 == 1:
                maps.append(a)
        ctx.stem, ctx.blocks, ctx.a_in, ctx.geom = layers[0], blocks, a_in, (n, h, w)
        ctx.params = (w_stem, g_stem, b_stem)
        ctx.zero = f["zero"]
        return tuple(maps)

    @staticmethod
    def backward(ctx, *dmaps):
        n, h, w = ctx.geom
        adt = ctx.stem.y.dtype
        dev = ctx.stem.y.device
        zero = ctx.zero

        def as_act(g, like):
            if g is None:
                return None
            g = g if g.is_contiguous() else g.contiguous()
            return g if g.dtype == like.dtype else g.to(like.dtype)

        def bn_bwd(dy, L, relu=True, want_g=False, dgamma=None, dbeta=None, acc=False):
            dz = torch.empty(L.y.shape, dtype=adt, device=dev)
            g = torch.empty(L.y.shape, dtype=adt, device=dev) if want_g else None
            kn.nhwc_bn_train_bwd(dy, L.y if relu else None, L.z, L.y.numel() // L.cout, L.cout, L.bn.weight, L.saved, dz, g_out=g,
                                 dgamma=dgamma, dbeta=dbeta, accumulate_params=acc)
            return dz, g

        def conv_dgrad(dz, L, add=None):
            src, hs, ws_ = dz, L.oh, L.ow
            if L.stride == 2:                                   # zero insertion: the stride-2 layer's data gradient as a stride-1 convolution
                src = kn.nhwc_scatter(dz, torch.empty((n, L.hin, L.win, L.cout), dtype=adt, device=dev), 2, 0)
                hs, ws_ = L.hin, L.win
            dx = torch.empty((n, L.hin, L.win, L.cin), dtype=adt, device=dev)
            kn.conv2d_padded_fwd(src, L.wd, zero[:L.cin], dx, n, hs, ws_, L.cout, L.cin, L.k, L.k, 1, L.k - 1 - L.pad, relu=False, add=add)
            return dx

        # stage outputs are maps[1..4] = the outputs of blocks 1, 3, 5, 7; maps[0] feeds block 0
        dy = None
        for bi in range(len(ctx.blocks) - 1, -1, -1):
            rec1, rec2, rec_d = ctx.blocks[bi]
            if bi % 2 == 1:
                dm = as_act(dmaps[(bi + 1) // 2], rec2.y)
                dy = dm if dy is None else (dy if dm is None else dy.add_(dm))
            if dy is None:                                      # nothing downstream of this block consumed its output
                continue
            dz2, g2 = bn_bwd(dy, rec2, relu=True, want_g=True)
            do = conv_dgrad(dz2, rec2)
            dz1, _ = bn_bwd(do, rec1, relu=True)
            if rec_d is None:
                dy = conv_dgrad(dz1, rec1, add=g2)
            else:
                dzd, _ = bn_bwd(g2, rec_d, relu=False)
                dy = conv_dgrad(dz1, rec1, add=conv_dgrad(dzd, rec_d))
        d0 = as_act(dmaps[0], ctx.stem.y)
        da0 = d0 if dy is None else (dy if d0 is None else dy.add_(d0))
        w_stem, g_stem, b_stem = ctx.params
        if da0 is None:
            return None, None, None, None, None
        S = ctx.stem
        dp = kn.maxpool_nhwc_bwd(S.y, da0, torch.empty(S.y.shape, dtype=adt, device=dev), n, S.oh, S.ow, 64, 3, 2, 1)

        def vec(param):                                        # (destination, accumulate, what autograd gets)
            sink = gradsink.get(param)
            if sink is None:
                t = torch.empty(param.shape, dtype=torch.float32, device=dev)
                return t, False, t
            return sink, not gradsink.first_write(param), None
        dga, acc_g, ret_g = vec(g_stem)
        dbe, acc_b, ret_b = vec(b_stem)
        if acc_g != acc_b:                                     # one flag for both: make the one that must not accumulate zero first
            (dga if not acc_g else dbe).zero_()
            acc_g = acc_b = True
        dz, _ = bn_bwd(dp, S, relu=True, dgamma=dga, dbeta=dbe, acc=acc_g)
        # d conv1.weight: 7 x 7 stride 2 over the zero-padded NHWC-8 input (channels 3..7 are zero: their gradient columns are dropped)
        xp = kn.nhwc_scatter(ctx.a_in, torch.empty((n, h + 6, w + 6, 8), dtype=ctx.a_in.dtype, device=dev), 1, 3)
        dw8 = torch.empty((64, 8 * 49), dtype=torch.float32, device=dev)
        kn.conv2d_bwd_weight(xp, dz, dw8, None, n, h + 6, w + 6, 8, 64, 7, 7, 2, False, dw_oihw=True)
        dw = dw8.view(64, 8, 7, 7)[:, :3]
        sink = gradsink.get(w_stem)
        if sink is not None:
            if gradsink.first_write(w_stem):
                sink.copy_(dw)
            else:
                sink.add_(dw)
            ret_w = None
        else:
            ret_w = dw.contiguous()
        return None, ret_w, ret_g, ret_b, None
