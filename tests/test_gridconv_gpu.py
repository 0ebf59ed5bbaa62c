"""Kernels of the affordance model's trainable part (csrc/gridconv.hip, affordance.hip, wgrad_group's col_mul) against
torch fp64 on the bf16-rounded operands: the padded-grid 3 x 3 convolution, its data gradient (same kernel, flipped / transposed weights) and
weight gradient (nine grouped products), BatchNorm(batch statistics) + ReLU forward / backward, the DecoderBlock input (language fusion,
nearest up-sampling, skip concatenation) forward / backward, and the pixel cross-entropy."""
import sys
from pathlib import Path

import pytest
import torch
import torch.nn.functional as F

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import kernels as kn  # noqa: E402


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    kn.set_compute("bf16")
    return torch.device("cuda", 0)


def to_grid(x_nchw, dev) -> "kn.Grid":
    N, C, H, W = x_nchw.shape
    g = kn.Grid(N, H, W, C, dev)
    g.t.zero_()
    g.interior().copy_(x_nchw.permute(0, 2, 3, 1).to(dev))
    return g


def from_grid(g) -> torch.Tensor:
    return g.interior().float().permute(0, 3, 1, 2).cpu()


def borders_zero(g) -> bool:
    v = g.t.view(g.N, g.H + 2, g.W + 2, g.C).float()
    return bool((v[:, 0] == 0).all() and (v[:, -1] == 0).all() and (v[:, :, 0] == 0).all() and (v[:, :, -1] == 0).all())


def bf(x):
    return x.to(torch.bfloat16).double()


def fwd_weights(w):           # (Cout, Cin, 3, 3) -> bf16 [Cout][9 Cin], k = tap * Cin + ci
    return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1).contiguous().to(torch.bfloat16)


def dgrad_weights(w):         # -> bf16 [Cin][9 Cout]: (ci, kh, kw, co), taps NOT flipped (the kernel's flip flag walks them backwards)
    return w.permute(1, 2, 3, 0).reshape(w.shape[1], -1).contiguous().to(torch.bfloat16)


# the thin-layer kernel (Cin 32 / 64, filter in registers, persistent workgroups) takes (64, 32), (32, 64), (32, 32): one tile, a dozen tiles
# over the XCDs, and 545 tiles (two trips for some workgroups)
@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 6, 5, 64, 32), (3, 14, 14, 96, 128), (2, 9, 9, 32, 64), (1, 30, 17, 160, 256), (2, 11, 7, 32, 32),
                                            (3, 20, 20, 64, 32), (4, 130, 130, 32, 32), (2, 40, 33, 32, 64), (2, 23, 23, 64, 64)])
def test_gridconv_forward_dgrad_wgrad(N, H, W, Cin, Cout):
    dev = _dev()
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    dy = torch.randn(N, Cout, H, W, generator=g)
    xr = bf(x).requires_grad_(True)
    wr = bf(w).requires_grad_(True)
    want = F.conv2d(xr, wr, padding=1)
    # forward + statistics
    X = to_grid(x, dev)
    Y, stats = kn.gridconv3x3(X, fwd_weights(w).to(dev), Cout, want_stats=True)
    torch.cuda.synchronize()
    got = from_grid(Y)
    assert borders_zero(Y)
    assert (got.double() - want.detach()).abs().max().item() <= 2e-2 * want.abs().max().item()            # bf16 output rounding
    nb = (X.R + 127) // 128
    st = stats[:nb * 2 * Cout].view(nb, 2, Cout).double().sum(0).cpu()
    assert (st[0] - want.detach().sum((0, 2, 3))).abs().max().item() < 1e-3 * max(1.0, want.detach().abs().sum((0, 2, 3)).max().item())
    assert (st[1] - (want.detach() ** 2).sum((0, 2, 3))).abs().max().item() < 1e-3 * (want.detach() ** 2).sum((0, 2, 3)).max().item()
    # data gradient: the same kernel on the gradient map
    want.backward(bf(dy))
    DY = to_grid(dy, dev)
    DX, _ = kn.gridconv3x3(DY, dgrad_weights(w).to(dev), Cin, flip=True)
    torch.cuda.synchronize()
    assert borders_zero(DX)
    assert (from_grid(DX).double() - xr.grad).abs().max().item() <= 2e-2 * xr.grad.abs().max().item()
    # weight gradient: nine items of the grouped launch, straight into the OIHW layout
    dw = torch.full((Cout, Cin * 9), float("nan"), device=dev)
    Wp = W + 2
    for t in range(9):
        off = (t // 3 - 1) * Wp + (t % 3 - 1)
        B = X.rows[X.guard + off: X.guard + off + X.Rpad]
        kn._wg_pending.setdefault(dev, []).append((DY.rows[DY.guard: DY.guard + DY.Rpad], B, dw[:, t:], None, Cout, Cin, X.Rpad, Cout, Cin, Cin * 9, False, False, 0, 9))
    kn.wgrad_flush(dev)
    torch.cuda.synchronize()
    got_w = dw.view(Cout, Cin, 3, 3).cpu().double()
    assert torch.isfinite(got_w).all()
    assert (got_w - wr.grad).abs().max().item() <= 1e-3 * wr.grad.abs().max().item() + 1e-4


# (N, H, W, Cin, Cout): every wave arrangement of csrc/wgrad_taps.hip (64 x 64 blocks, 32 output / input channels, both), ragged last k-step
# (rows % 64 == 32), one slice and split tiles (> 64 k-steps; > 256 for the 32 x 32 arrangement), several tiles per item
TAP_SHAPES = [(2, 6, 5, 64, 64), (1, 30, 17, 192, 128), (4, 40, 40, 64, 128), (3, 9, 9, 64, 32), (2, 30, 30, 32, 64), (5, 62, 62, 64, 32),
              (8, 46, 46, 32, 32), (2, 11, 7, 32, 32), (2, 14, 14, 768, 512),
              (1, 7, 7, 32, 32), (1, 7, 7, 64, 32), (1, 7, 7, 32, 64), (1, 7, 7, 128, 64)]        # 96 rows: the last k-step is half empty in every wave arrangement


@pytest.mark.parametrize("N,H,W,Cin,Cout", TAP_SHAPES)
def test_nine_tap_weight_gradient(N, H, W, Cin, Cout):
    """conv_taps_wp items (csrc/wgrad_taps.hip: all nine taps of a tile from one pass over dZ and X) against fp64 conv2d weight gradients of
    the bf16-rounded operands; written twice (store, then accumulate on top) and repeated bit for bit"""
    dev = _dev()
    g = torch.Generator().manual_seed(N * 1000 + H * 10 + Cin)
    x = torch.randn(N, Cin, H, W, generator=g)
    dy = torch.randn(N, Cout, H, W, generator=g)
    xr = bf(x)
    wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
    F.conv2d(xr, wr, padding=1).backward(bf(dy))
    X, DY = to_grid(x, dev), to_grid(dy, dev)
    outs = []
    for rep in range(2):
        dw = torch.full((Cout, Cin * 9), float("nan"), device=dev)
        A = DY.rows[DY.guard:DY.guard + DY.Rpad]
        B = X.rows[X.guard:X.guard + X.Rpad]
        kn.wgrad(A, B, dw, Cout, Cin, X.Rpad, Cout, Cin, Cin * 9, col_mul=9, conv_taps_wp=W + 2)
        kn.wgrad(A, B, dw, Cout, Cin, X.Rpad, Cout, Cin, Cin * 9, accumulate=True, col_mul=9, conv_taps_wp=W + 2)
        torch.cuda.synchronize()
        outs.append(dw.clone())
    got = outs[0].view(Cout, Cin, 3, 3).cpu().double()
    assert torch.isfinite(got).all()
    err = (got - 2 * wr.grad).abs().max().item()
    assert err <= 2e-5 * (N * H * W) ** 0.5 * 8 + 1e-4, err          # fp32 accumulation of N H W unit-variance products (x 2)
    assert torch.equal(outs[0], outs[1])
    # store_rows: only the first output channel leaves the kernel (the one-channel segmentation head padded to 32)
    if Cout == 32:
        one = torch.full((1, Cin * 9), float("nan"), device=dev)
        kn.wgrad(A, B, one, Cout, Cin, X.Rpad, Cout, Cin, Cin * 9, col_mul=9, store_rows=1, conv_taps_wp=W + 2)
        torch.cuda.synchronize()
        assert (one.cpu().double() - wr.grad.view(Cout, -1)[:1]).abs().max().item() <= 2e-5 * (N * H * W) ** 0.5 * 4 + 1e-4


def test_nine_tap_items_beyond_one_launch_table():
    """twenty nine-tap items in one hulc_wgrad_group call (the kernel's item table holds 16: two launches, slab regions one after the other)
    next to three ordinary products"""
    dev = _dev()
    g = torch.Generator().manual_seed(77)
    q = kn._wg_pending.setdefault(dev, [])
    want, outs = [], []
    for i in range(20):
        N, H, W = 2, 9 + i % 3, 40 + i                                  # > 64 k-steps for some: split tiles among unsplit ones
        Cin, Cout = (64, 32) if i % 2 else (32, 64)
        x = torch.randn(N, Cin, H, W, generator=g)
        dy = torch.randn(N, Cout, H, W, generator=g)
        wr = torch.zeros(Cout, Cin, 3, 3, dtype=torch.float64, requires_grad=True)
        F.conv2d(bf(x), wr, padding=1).backward(bf(dy))
        X, DY = to_grid(x, dev), to_grid(dy, dev)
        dw = torch.full((Cout, Cin * 9), float("nan"), device=dev)
        q.append((DY.rows[DY.guard:DY.guard + DY.Rpad], X.rows[X.guard:X.guard + X.Rpad], dw, None, Cout, Cin, X.Rpad, Cout, Cin, Cin * 9, False, False, 0, 9, 0, W + 2))
        want.append(wr.grad)
        outs.append((dw, X, DY))
    plain = []
    for M, Nn, K in [(64, 128, 256), (8, 24, 96), (200, 72, 64)]:
        A, B = torch.randn(K, M, generator=g).to(dev), torch.randn(K, Nn, generator=g).to(dev)
        C = torch.empty(M, Nn, device=dev)
        q.append((A, B, C, None, M, Nn, K, M, Nn, Nn, False, False, 0, 1, 0, 0))
        plain.append((A, B, C))
    kn.wgrad_flush(dev)
    torch.cuda.synchronize()
    for (dw, X, DY), w in zip(outs, want):
        got = dw.view(w.shape).cpu().double()
        assert torch.isfinite(got).all() and (got - w).abs().max().item() <= 2e-5 * X.R ** 0.5 * 4 + 1e-4
    for A, B, C in plain:
        ref = A.to(torch.bfloat16).double().t() @ B.to(torch.bfloat16).double()
        assert (C.double() - ref).abs().max().item() <= 1e-3


@pytest.mark.parametrize("N,H,W,C", [(2, 7, 5, 64), (3, 12, 12, 32), (2, 5, 9, 128)])
def test_batchnorm_relu_forward_backward(N, H, W, C):
    dev = _dev()
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(N, C, H, W, generator=g) * 1.5 + 0.3
    gamma, beta = torch.rand(C, generator=g) + 0.5, torch.randn(C, generator=g) * 0.2
    dout = torch.randn(N, C, H, W, generator=g)
    # through an identity-like convolution so that the statistics come from gridconv's epilogue: 32-channel-aligned one-tap weights
    w = torch.zeros(C, C, 3, 3)
    w[torch.arange(C), torch.arange(C), 1, 1] = 1.0
    X = to_grid(x, dev)
    Y, stats = kn.gridconv3x3(X, fwd_weights(w).to(dev), C, want_stats=True)
    rm, rv = torch.zeros(C, device=dev), torch.ones(C, device=dev)
    bn = kn.grid_bn_finalize(stats, N, H, W, C, gamma.to(dev), beta.to(dev), rm, rv)
    O = kn.grid_bn_relu_fwd(Y, bn)
    torch.cuda.synchronize()
    yr = bf(x).requires_grad_(True)                                        # Y holds bf16(x) exactly
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    m = torch.nn.BatchNorm2d(C).double().train()
    with torch.no_grad():
        m.weight.copy_(gr); m.bias.copy_(br)
    want = F.relu(m(yr))
    assert borders_zero(O)
    assert (from_grid(O).double() - want.detach()).abs().max().item() <= 1.5e-2 * want.detach().abs().max().item()
    assert (rm.cpu().double() - m.running_mean).abs().max().item() < 1e-4 and (rv.cpu().double() - m.running_var).abs().max().item() < 1e-3
    want.backward(bf(dout))
    dg_, db_ = torch.empty(C, device=dev), torch.empty(C, device=dev)
    DZ = kn.grid_bn_relu_bwd(to_grid(dout, dev), O, Y, bn, dg_, db_)
    torch.cuda.synchronize()
    assert borders_zero(DZ)
    # the ReLU mask comes from the bf16 output: elements within rounding of zero may flip; compare in the L2 sense
    e = (from_grid(DZ).double() - yr.grad).norm() / yr.grad.norm()
    assert e < 2e-2, e
    assert (dg_.cpu().double() - m.weight.grad).abs().max().item() <= 2e-2 * m.weight.grad.abs().max().item()
    assert (db_.cpu().double() - m.bias.grad).abs().max().item() <= 2e-2 * m.bias.grad.abs().max().item()


@pytest.mark.parametrize("N,Hi,Wi,s,Cx,Cs,fused,skip_plain", [(2, 3, 4, 2, 64, 32, True, True), (2, 5, 5, 1, 128, 64, False, True), (3, 4, 4, 4, 64, 0, False, False)])
def test_decoder_block_input_forward_backward(N, Hi, Wi, s, Cx, Cs, fused, skip_plain):
    dev = _dev()
    g = torch.Generator().manual_seed(Cx + s)
    x = torch.randn(N, Cx, Hi, Wi, generator=g)
    gv = torch.randn(N, Cx, generator=g) if fused else None
    Ho, Wo = Hi * s, Wi * s
    skip = torch.randn(N, Cs, Ho, Wo, generator=g) if Cs else None
    dX = torch.randn(N, Cx + Cs, Ho, Wo, generator=g)
    XS = to_grid(x, dev)                                                   # the small map as a grid tensor (a previous block's output)
    xt, xsn, xsy, xsx = XS.pixel_strides()
    sk = skip.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev) if Cs else None
    out = kn.grid_upcat_fwd(xt, (xsn, xsy, xsx), gv.to(dev) if fused else None, sk, (Ho * Wo * Cs, Wo * Cs, Cs), N, Ho, Wo, s, Cx, Cs)
    torch.cuda.synchronize()
    xr = bf(x).requires_grad_(True)
    gr = gv.double().requires_grad_(True) if fused else None
    up = F.interpolate(xr * gr[:, :, None, None] if fused else xr, scale_factor=s, mode="nearest")
    want = torch.cat([up, bf(skip)], 1) if Cs else up
    assert borders_zero(out)
    assert (from_grid(out).double() - want.detach()).abs().max().item() <= 1e-2 * want.detach().abs().max().item()
    want.backward(bf(dX))
    D = to_grid(dX, dev)
    dsmall, dgv = kn.grid_upcat_bwd(D, xt, (xsn, xsy, xsx), gv.to(dev) if fused else None, N, Hi, Wi, s, Cx, want_dsmall=True, want_dg=fused)
    torch.cuda.synchronize()
    assert (from_grid(dsmall).double() - xr.grad).abs().max().item() <= 1e-2 * xr.grad.abs().max().item()
    if fused:
        assert (dgv.cpu().double() - gr.grad).abs().max().item() <= 1e-3 * gr.grad.abs().max().item()


def test_pixel_cross_entropy_with_head():
    """the one-channel head as a 32-channel gridconv with fp32 channel 0, then the cross-entropy over the pixels and its backward"""
    dev = _dev()
    N, H, W, C = 3, 11, 13, 32
    g = torch.Generator().manual_seed(5)
    x = torch.randn(N, C, H, W, generator=g)
    w = torch.randn(1, C, 3, 3, generator=g) * 0.3
    b = torch.randn(1, generator=g)
    p0 = torch.stack([torch.randint(0, H, (N,), generator=g), torch.randint(0, W, (N,), generator=g)], 1).to(torch.int32)
    w32 = torch.zeros(32, C, 3, 3)
    w32[0] = w[0]
    X = to_grid(x, dev)
    logit0 = torch.empty(X.R, dtype=torch.float32, device=dev)
    kn.gridconv3x3(X, fwd_weights(w32).to(dev), 32, out0=logit0, bias0=b.to(dev))
    lse, picked = kn.pixel_ce_fwd(logit0, p0.to(dev), N, H, W)
    up = torch.tensor([0.7], device=dev)
    DZ = kn.pixel_ce_bwd(logit0, p0.to(dev), lse, up, N, H, W, 32)
    torch.cuda.synchronize()
    xr = bf(x)
    lg = (F.conv2d(xr, bf(w), b.double(), padding=1)).permute(0, 2, 3, 1).reshape(N, -1).requires_grad_(True)
    label = torch.zeros(N, H, W, dtype=torch.double)
    label[torch.arange(N), p0[:, 0].long(), p0[:, 1].long()] = 1
    loss = (-label.reshape(N, -1) * F.log_softmax(lg, -1)).mean()
    got_loss = -(picked - lse).sum().item() / (N * H * W)
    assert abs(got_loss - loss.item()) <= 1e-5 * abs(loss.item()) + 1e-7
    (loss * 0.7).backward()
    got = from_grid(DZ)[:, 0].reshape(N, -1).double()
    assert (got - lg.grad).abs().max().item() <= 1e-2 * lg.grad.abs().max().item()
    assert float(from_grid(DZ)[:, 1:].abs().max()) == 0.0 and borders_zero(DZ)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 9, 9, 64, 64), (2, 7, 7, 128, 128), (3, 20, 20, 64, 32), (1, 14, 14, 256, 256), (2, 12, 10, 32, 32)])
@pytest.mark.parametrize("with_add,relu", [(True, True), (False, True), (True, False)])
def test_gridconv_basic_block_epilogue(N, H, W, Cin, Cout, with_add, relu):
    """hulc_gridconv3x3_fused (the frozen ResNet trunk's BasicBlock convolutions: folded-BatchNorm bias, residual branch, ReLU) and
    hulc_grid_from_nhwc against fp64 conv2d on the bf16-rounded operands"""
    dev = _dev()
    g = torch.Generator().manual_seed(N * 100 + Cin + Cout)
    x = torch.randn(N, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / (3 * Cin ** 0.5)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(N, Cout, H, W, generator=g)
    X = kn.grid_from_nhwc(x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev))
    assert borders_zero(X) and torch.equal(from_grid(X), x.to(torch.bfloat16).float())
    A = kn.grid_from_nhwc(res.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16).to(dev)) if with_add else None
    Y = kn.gridconv3x3_fused(X, fwd_weights(w).to(dev), Cout, bias=b.to(dev), add=A, relu=relu)
    torch.cuda.synchronize()
    want = F.conv2d(bf(x), bf(w), b.double(), padding=1) + (bf(res) if with_add else 0)
    if relu:
        want = want.relu()
    assert borders_zero(Y)
    assert (from_grid(Y).double() - want).abs().max().item() <= 1e-2 * want.abs().max().item()


@pytest.mark.parametrize("N,H,W,C", [(3, 11, 13, 32), (2, 40, 37, 64), (5, 60, 64, 8), (2, 200, 190, 32)])
def test_one_channel_head_streaming_kernels(N, H, W, C):
    """hulc_head_conv_fwd / _dgrad / _wgrad + hulc_pixel_ce_bwd_rows (the head without matrix cores) against fp64 conv2d autograd on the
    bf16-rounded activations; the last shape has more rows than one weight-gradient workgroup pass (several blocks + the fixed-order sum)"""
    dev = _dev()
    gen = torch.Generator().manual_seed(N * 7 + C)
    x = torch.randn(N, C, H, W, generator=gen)
    w = torch.randn(1, C, 3, 3, generator=gen) * 0.3
    b = torch.randn(1, generator=gen)
    p0 = torch.stack([torch.randint(0, H, (N,), generator=gen), torch.randint(0, W, (N,), generator=gen)], 1).to(torch.int32)
    X = to_grid(x, dev)
    wd, bd = w.to(dev), b.to(dev)
    logit0 = kn.head_conv_fwd(X, wd, bd)
    lse, picked = kn.pixel_ce_fwd(logit0, p0.to(dev), N, H, W)
    up = torch.tensor([0.7], device=dev)
    g = kn.pixel_ce_bwd_rows(logit0, p0.to(dev), lse, up, N, H, W)
    dX = kn.head_conv_dgrad(g, wd, N, H, W, C)
    dw = torch.full((C * 9,), 2.0, device=dev)
    kn.head_conv_wgrad(X, g, dw, accumulate=True)
    dw2 = torch.full((C * 9,), float("nan"), device=dev)
    kn.head_conv_wgrad(X, g, dw2)
    torch.cuda.synchronize()
    xr = bf(x).requires_grad_(True)
    wr = w.double().requires_grad_(True)
    lg = F.conv2d(xr, wr, b.double(), padding=1)
    got_lg = logit0.view(N, H + 2, W + 2)[:, 1:-1, 1:-1].cpu().double()
    assert (got_lg - lg.detach()[:, 0]).abs().max().item() <= 1e-5 * lg.detach().abs().max().item() + 1e-5
    v = logit0.view(N, H + 2, W + 2)
    assert float(v[:, 0].abs().max()) == 0.0 and float(v[:, :, 0].abs().max()) == 0.0 and float(v[:, -1].abs().max()) == 0.0
    label = torch.zeros(N, H, W, dtype=torch.double)
    label[torch.arange(N), p0[:, 0].long(), p0[:, 1].long()] = 1
    loss = (-label.reshape(N, -1) * F.log_softmax(lg.permute(0, 2, 3, 1).reshape(N, -1), -1)).mean()
    (loss * 0.7).backward()
    assert borders_zero(dX)
    assert (from_grid(dX).double() - xr.grad).abs().max().item() <= 1e-2 * xr.grad.abs().max().item()          # bf16 output
    want_w = wr.grad.reshape(-1)
    assert (dw2.cpu().double() - want_w).abs().max().item() <= 2e-4 * want_w.abs().max().item() + 1e-9
    assert (dw.cpu().double() - 2.0 - want_w).abs().max().item() <= 2e-4 * want_w.abs().max().item() + 1e-6


@pytest.mark.parametrize("B,D", [(32, 256), (3, 256), (70, 96)])
def test_depth_head_gaussian_nll(B, D):
    """hulc_depth_nll_fwd / _bwd against torch autograd of the reference's expressions (depth_gaussian.py:67-69,94-102): both clamp ranges
    are entered (log_sigma beyond +2 and below -20, sigma under the 1e-6 floor), store and accumulate"""
    dev = _dev()
    g = torch.Generator().manual_seed(B + D)
    x = torch.randn(B, D, generator=g)
    w_mu, b_mu = torch.randn(1, D, generator=g) / D ** 0.5, torch.randn(1, generator=g)
    w_s, b_s = torch.randn(1, D, generator=g) * (10.0 / D ** 0.5), torch.tensor([-6.0])
    t = torch.randn(B, generator=g)
    ref = [v.double().requires_grad_(True) for v in (x, w_mu, b_mu, w_s, b_s)]
    mu_r = F.linear(ref[0], ref[1], ref[2])
    ls_r = F.linear(ref[0], ref[3], ref[4])
    sig_r = torch.clamp(ls_r, -20, 2).exp()
    var = torch.clamp(sig_r, min=1e-6)
    loss_r = (0.5 * (torch.log(var) + (mu_r - t.double().reshape(-1, 1)) ** 2 / var)).mean()
    (loss_r * 0.9).backward()
    assert B < 32 or ((ls_r > 2).any() and (ls_r < -13.9).any() and (ls_r < -20).any() or B < 32)      # the ranges the test is about
    xd, wm, bm, ws, bs, td = (v.to(dev) for v in (x, w_mu, b_mu, w_s, b_s, t))
    mu, sigma, ls, loss = kn.depth_nll_fwd(xd, wm, bm, ws, bs, td)
    torch.cuda.synchronize()
    assert abs(loss.item() - loss_r.item()) <= 1e-4 * abs(loss_r.item()) + 1e-5
    assert (mu.cpu().double() - mu_r.detach()).abs().max().item() < 1e-4 and (sigma.cpu().double() / sig_r.detach() - 1).abs().max().item() < 1e-3
    dx = torch.empty_like(xd)
    outs = [torch.full_like(v, 1.0) for v in (wm, bm, ws, bs)]
    gout = torch.tensor([0.9], device=dev)
    kn.depth_nll_bwd(xd, wm, ws, mu, sigma, ls, td, gout, dx, *outs, accumulate_mask=15)
    torch.cuda.synchronize()
    for got, want, base in zip([dx] + outs, [r.grad for r in ref], [0.0, 1.0, 1.0, 1.0, 1.0]):
        w = want + base
        assert (got.cpu().double() - w).abs().max().item() <= 2e-4 * w.abs().max().item() + 1e-6
