"""GPU parity of the conv kernels (forward / data gradient / weight gradient) against torch's
conv2d autograd in float64 on the same (bf16-rounded, for bf16 compute) operands."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

# (name, N, H, W, Cin, Cout, KH, stride, x_nchw) — the six conv layers of the two camera encoders
LAYERS = [
    ("static1", 3, 200, 200, 3, 32, 8, 4, True),
    ("static2", 3, 49, 49, 32, 64, 4, 2, False),
    ("static3", 3, 23, 23, 64, 64, 3, 1, False),
    ("grip1", 5, 84, 84, 3, 32, 8, 4, True),
    ("grip2", 5, 20, 20, 32, 64, 4, 2, False),
    ("grip3", 5, 9, 9, 64, 64, 3, 1, False),
    # 36 x 36 frames: 8 output columns = one 8-pixel block per row — the conv1 weight-gradient band kernel leaves the shape to the generic one
    ("tiny1", 4, 36, 36, 3, 32, 8, 4, True),
]


def _round(t, compute):
    return t.to(torch.bfloat16).double() if compute == "bf16" else t.double()


def _setup(dev, N, H, W, Cin, Cout, K, seed):
    g = torch.Generator().manual_seed(seed)
    x = (torch.rand(N, Cin, H, W, generator=g) * 2 - 1)
    w = (torch.rand(Cout, Cin, K, K, generator=g) * 2 - 1) / (Cin * K * K) ** 0.5
    b = (torch.rand(Cout, generator=g) * 2 - 1) * 0.1
    return x.to(dev), w.to(dev), b.to(dev), g


@pytest.mark.parametrize("compute", ["bf16", "fp32"])
@pytest.mark.parametrize("name,N,H,W,Cin,Cout,K,stride,nchw", LAYERS)
def test_conv_forward(dev, compute, name, N, H, W, Cin, Cout, K, stride, nchw):
    from hulc2_amd import kernels as kn

    x, w, b, _ = _setup(dev, N, H, W, Cin, Cout, K, 11)
    OH, OW = kn.conv_out_hw(H, W, K, K, stride)
    xin = x.contiguous() if nchw else x.permute(0, 2, 3, 1).contiguous()
    w2d = w.reshape(Cout, -1).contiguous() if nchw else w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous()
    y = torch.full((N, OH, OW, Cout), float("nan"), device=dev)
    kn.conv2d_fwd(xin, w2d, b, y, N, H, W, Cin, Cout, K, K, stride, nchw, relu=True, compute=kn._COMPUTE[compute])
    torch.cuda.synchronize()
    ref = torch.relu(F.conv2d(_round(x, compute), _round(w, compute), b.double(), stride=stride)).permute(0, 2, 3, 1)
    err = (y.double() - ref).abs().max().item()
    assert torch.isfinite(y).all()
    assert err < 2e-4, f"{name}: max err {err:.3e} (ref max {ref.abs().max().item():.3e})"


@pytest.mark.parametrize("compute", ["bf16", "fp32"])
@pytest.mark.parametrize("name,N,H,W,Cin,Cout,K,stride,nchw", [l for l in LAYERS if not l[-1]])
def test_conv_bwd_data(dev, compute, name, N, H, W, Cin, Cout, K, stride, nchw):
    from hulc2_amd import kernels as kn

    x, w, b, g = _setup(dev, N, H, W, Cin, Cout, K, 12)
    OH, OW = kn.conv_out_hw(H, W, K, K, stride)
    dy = torch.randn(N, Cout, OH, OW, generator=g).to(dev)
    relu_src = torch.randn(N, H, W, Cin, generator=g).to(dev)
    wt = w.permute(1, 2, 3, 0).contiguous()            # [Cin][KH][KW][Cout]
    dx = torch.full((N, H, W, Cin), float("nan"), device=dev)
    kn.conv2d_bwd_data(dy.permute(0, 2, 3, 1).contiguous(), wt, dx, relu_src, N, H, W, Cin, Cout, K, K, stride,
                       compute=kn._COMPUTE[compute])
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_input((N, Cin, H, W), _round(w, compute), _round(dy, compute), stride=stride)
    ref = ref.permute(0, 2, 3, 1) * (relu_src > 0)
    err = (dx.double() - ref).abs().max().item()
    assert torch.isfinite(dx).all(), f"{name}: non-finite (uncovered output pixels?)"
    assert err < 5e-4, f"{name}: max err {err:.3e} (ref max {ref.abs().max().item():.3e})"


@pytest.mark.parametrize("compute", ["bf16", "fp32"])
@pytest.mark.parametrize("name,N,H,W,Cin,Cout,K,stride,nchw", LAYERS)
def test_conv_bwd_weight(dev, compute, name, N, H, W, Cin, Cout, K, stride, nchw):
    from hulc2_amd import kernels as kn

    x, w, b, g = _setup(dev, N, H, W, Cin, Cout, K, 13)
    OH, OW = kn.conv_out_hw(H, W, K, K, stride)
    dy = torch.randn(N, Cout, OH, OW, generator=g).to(dev)
    xin = x.contiguous() if nchw else x.permute(0, 2, 3, 1).contiguous()
    Kd = Cin * K * K
    dw = torch.full((Cout, Kd), float("nan"), device=dev)
    db = torch.full((Cout,), float("nan"), device=dev)
    kn.conv2d_bwd_weight(xin, dy.permute(0, 2, 3, 1).contiguous(), dw, db, N, H, W, Cin, Cout, K, K, stride, nchw,
                         compute=kn._COMPUTE[compute])
    torch.cuda.synchronize()
    ref = torch.nn.grad.conv2d_weight(_round(x, compute), (Cout, Cin, K, K), _round(dy, compute), stride=stride)
    ref2d = ref.reshape(Cout, -1) if nchw else ref.permute(0, 2, 3, 1).reshape(Cout, -1)
    scale = ref2d.abs().max().item()
    err = (dw.double() - ref2d).abs().max().item()
    assert err < 1e-4 * scale + 1e-4, f"{name}: dW max err {err:.3e} (scale {scale:.3e})"
    refb = dy.double().sum(dim=(0, 2, 3))
    errb = (db.double() - refb).abs().max().item()
    assert errb < 1e-4 * refb.abs().max().item() + 1e-3, f"{name}: db max err {errb:.3e}"
    # gradient-arena form: the parameter's OIHW order, accumulated into the destination
    base_w, base_b = torch.randn(Cout, Kd, generator=g).to(dev), torch.randn(Cout, generator=g).to(dev)
    dw2, db2 = base_w.clone(), base_b.clone()
    kn.conv2d_bwd_weight(xin, dy.permute(0, 2, 3, 1).contiguous(), dw2, db2, N, H, W, Cin, Cout, K, K, stride, nchw,
                         compute=kn._COMPUTE[compute], dw_oihw=True, accumulate=True)
    torch.cuda.synchronize()
    want = base_w.double() + ref.reshape(Cout, -1)
    assert (dw2.double() - want).abs().max().item() < 1e-4 * scale + 1e-4, f"{name}: OIHW accumulated dW"
    assert (db2.double() - (base_b.double() + refb)).abs().max().item() < 1e-4 * refb.abs().max().item() + 1e-3, f"{name}: accumulated db"


def _pack_bits(y_nhwc):
    """reference sign planes: (C / 32, npix) int32 flattened, bit c % 32 of plane c / 32 = (y > 0)"""
    pos = (y_nhwc.float() > 0).reshape(-1, y_nhwc.shape[-1] // 32, 32).to(torch.int64)
    w = (pos << torch.arange(32, device=y_nhwc.device)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).t().contiguous().reshape(-1)


@pytest.mark.parametrize("name,N,H,W,Cin,Cout,K,stride,nchw", [l for l in LAYERS if l[0] in ("static1", "static2", "grip1", "grip2")] +
                         [("odd1", 2, 62, 62, 3, 32, 8, 4, True), ("odd2", 2, 17, 21, 32, 64, 4, 2, False)])
def test_relu_sign_planes_written_by_the_forward(dev, name, N, H, W, Cin, Cout, K, stride, nchw):
    """hulc_conv_desc.relu_bits: the planes next to a bf16 ReLU output equal (y > 0) bit for bit — from the band kernels' epilogues and, for
    geometries they do not take (odd1 / odd2), from the second pass behind the gather kernel; y itself is unchanged by asking for them"""
    from hulc2_amd import kernels as kn

    x, w, b, _ = _setup(dev, N, H, W, Cin, Cout, K, 21)
    OH, OW = kn.conv_out_hw(H, W, K, K, stride)
    xin = x.contiguous() if nchw else x.permute(0, 2, 3, 1).contiguous().to(torch.bfloat16)
    w2d = (w.reshape(Cout, -1) if nchw else w.permute(0, 2, 3, 1).reshape(Cout, -1)).contiguous().to(torch.bfloat16)
    y0 = torch.empty(N, OH, OW, Cout, device=dev, dtype=torch.bfloat16)
    y1 = torch.empty_like(y0)
    bits = torch.full((N * OH * OW * (Cout // 32),), 0x5A5A5A5A, dtype=torch.int32, device=dev)
    kn.conv2d_fwd(xin, w2d, b, y0, N, H, W, Cin, Cout, K, K, stride, nchw, relu=True, compute=kn.BF16)
    kn.conv2d_fwd(xin, w2d, b, y1, N, H, W, Cin, Cout, K, K, stride, nchw, relu=True, compute=kn.BF16, relu_bits=bits)
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    assert torch.equal(bits, _pack_bits(y1)), f"{name}: {(bits != _pack_bits(y1)).sum().item()} of {bits.numel()} words differ"
    assert 0.2 < (y1 > 0).float().mean().item() < 0.8            # (a meaningful mix of signs)


@pytest.mark.parametrize("name,N,H,W,Cin,Cout,K,stride,nchw", [l for l in LAYERS if not l[-1]] + [("odd2", 2, 17, 21, 32, 64, 4, 2, False)])
def test_data_gradient_masked_by_sign_planes_equals_masked_by_the_activation(dev, name, N, H, W, Cin, Cout, K, stride, nchw):
    from hulc2_amd import kernels as kn

    x, w, b, g = _setup(dev, N, H, W, Cin, Cout, K, 22)
    OH, OW = kn.conv_out_hw(H, W, K, K, stride)
    dy = torch.randn(N, OH, OW, Cout, generator=g).to(dev).to(torch.bfloat16)
    act = torch.relu(torch.randn(N, H, W, Cin, generator=g)).to(dev).to(torch.bfloat16)       # the layer input: a stored ReLU output
    wt = w.permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    a = torch.full((N, H, W, Cin), float("nan"), device=dev, dtype=torch.bfloat16)
    b2 = torch.full_like(a, float("nan"))
    kn.conv2d_bwd_data(dy, wt, a, act, N, H, W, Cin, Cout, K, K, stride, compute=kn.BF16)
    kn.conv2d_bwd_data(dy, wt, b2, act, N, H, W, Cin, Cout, K, K, stride, compute=kn.BF16, relu_bits=_pack_bits(act))
    torch.cuda.synchronize()
    assert torch.isfinite(a.float()).all() and torch.equal(a, b2), f"{name}: {(a != b2).sum().item()} elements differ"
    assert (a.float().abs() > 0).float().mean().item() > 0.2


@pytest.mark.parametrize("name,N,H,W", [("static1", 5, 200, 200), ("grip1", 7, 84, 84)])
def test_conv1_split_operands_are_fp32_class(dev, name, N, H, W):
    """hulc_conv_desc.w_lo (selective-precision site "conv1"): fp32 frames and weights as hi + lo bf16 splits, three MFMAs per product —
    the output agrees with the UNROUNDED float64 convolution to 2e-5 of its range (plain bf16 operands: ~3e-3), fp32 and bf16 outputs"""
    from hulc2_amd import kernels as kn

    x, w, b, _ = _setup(dev, N, H, W, 3, 32, 8, 13)
    OH, OW = kn.conv_out_hw(H, W, 8, 8, 4)
    w2d = w.reshape(32, -1).contiguous()
    whi = w2d.to(torch.bfloat16)
    wlo = (w2d - whi.float()).to(torch.bfloat16)
    ref = torch.relu(F.conv2d(x.double(), w.double(), b.double(), stride=4)).permute(0, 2, 3, 1)
    y = torch.full((N, OH, OW, 32), float("nan"), device=dev)
    kn.conv2d_fwd(x.contiguous(), whi, b, y, N, H, W, 3, 32, 8, 8, 4, True, relu=True, compute=kn.BF16, w_lo=wlo)
    y0 = torch.empty_like(y)
    kn.conv2d_fwd(x.contiguous(), whi, b, y0, N, H, W, 3, 32, 8, 8, 4, True, relu=True, compute=kn.BF16)
    torch.cuda.synchronize()
    scale = ref.abs().max().item()
    e, e0 = (y.double() - ref).abs().max().item() / scale, (y0.double() - ref).abs().max().item() / scale
    assert torch.isfinite(y).all()
    assert e < 2e-5 and e < 0.02 * e0, (name, e, e0)
    yb = torch.empty(N, OH, OW, 32, dtype=torch.bfloat16, device=dev)          # bf16 storage of the fp32-class values: one rounding
    kn.conv2d_fwd(x.contiguous(), whi, b, yb, N, H, W, 3, 32, 8, 8, 4, True, relu=True, compute=kn.BF16, w_lo=wlo)
    torch.cuda.synchronize()
    assert torch.equal(yb, y.to(torch.bfloat16)) or ((yb.double() - ref).abs().max().item() / scale) < 4e-3


@pytest.mark.parametrize("name,Na,Nb,H,W", [("static1", 5, 3, 200, 200), ("grip1", 2, 7, 84, 84), ("static1-empty-second", 4, 0, 200, 200)])
def test_conv1_two_frame_tensors_in_one_launch(dev, name, Na, Nb, H, W):
    """hulc_conv_desc.x2 / n_split (round 4): conv1's forward (with its ReLU sign plane) over two fp32 NCHW frame tensors as ONE launch is
    bit-identical to the two launches it replaces; the weight gradient (one set of slabs instead of two accumulating launches: another
    summation order) agrees to fp32 rounding and with torch's conv2d autograd in float64 on the bf16-rounded operands."""
    from hulc2_amd import kernels as kn
    g = torch.Generator().manual_seed(5)
    xa = (torch.rand(Na, 3, H, W, generator=g) * 2 - 1).to(dev)
    xb = (torch.rand(Nb, 3, H, W, generator=g) * 2 - 1).to(dev)
    w = ((torch.rand(32, 3, 8, 8, generator=g) * 2 - 1) / 192 ** 0.5).to(dev)
    b = ((torch.rand(32, generator=g) * 2 - 1) * 0.1).to(dev)
    w2d = w.reshape(32, -1).contiguous().to(torch.bfloat16)
    N = Na + Nb
    OH, OW = kn.conv_out_hw(H, W, 8, 8, 4)
    y1 = torch.zeros(N, OH, OW, 32, dtype=torch.bfloat16, device=dev)
    y2 = torch.zeros_like(y1)
    b1 = torch.zeros(N * OH * OW, dtype=torch.int32, device=dev)
    b2 = torch.zeros_like(b1)
    kn.conv2d_fwd(xa, w2d, b, y1[:Na], Na, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=b1[:Na * OH * OW])
    if Nb:
        kn.conv2d_fwd(xb, w2d, b, y1[Na:], Nb, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=b1[Na * OH * OW:])
    kn.conv2d_fwd(xa, w2d, b, y2, N, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=b2, x2=xb)
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)) and torch.equal(b1, b2)
    dy = torch.randn(N, OH, OW, 32, generator=g).to(dev).to(torch.bfloat16)
    dw1, db1 = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
    dw2, db2 = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
    kn.conv2d_bwd_weight(xa, dy[:Na].contiguous(), dw1, db1, Na, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16)
    if Nb:
        kn.conv2d_bwd_weight(xb, dy[Na:].contiguous(), dw1, db1, Nb, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, accumulate=True)
    kn.conv2d_bwd_weight(xa, dy, dw2, db2, N, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, x2=xb)
    torch.cuda.synchronize()
    assert float((dw1 - dw2).abs().max()) <= 1e-5 * float(dw1.abs().max()) and float((db1 - db2).abs().max()) <= 1e-5 * float(db1.abs().max())
    xall = torch.cat([xa, xb]).to(torch.bfloat16).double().requires_grad_(False)
    wd = torch.zeros(32, 3, 8, 8, dtype=torch.float64, device=dev, requires_grad=True)
    F.conv2d(xall, wd, None, stride=4).backward(dy.double().permute(0, 3, 1, 2))
    assert float((dw2.double() - wd.grad.reshape(32, -1)).abs().max()) <= 2e-3 * float(wd.grad.abs().max())


@pytest.mark.parametrize("name,Na,Nb,H,W,shift", [("static1", 5, 3, 200, 200, True), ("grip1", 2, 7, 84, 84, True), ("static1-no-aug", 3, 4, 200, 200, False)])
def test_conv1_two_uint8_frame_tensors_in_one_launch(dev, name, Na, Nb, H, W, shift):
    """x2 with uint8 NHWC frames: the two modalities' frame tensors, their per-frame augmentation shifts as one (N, 2)
    tensor — forward and sign plane bit-identical to the two launches, weight gradient to fp32 rounding (one set of slabs instead of two
    accumulating launches)."""
    from hulc2_amd import kernels as kn
    g = torch.Generator().manual_seed(6)
    xa = torch.randint(0, 256, (Na, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    xb = torch.randint(0, 256, (Nb, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    pad = 10 if H == 200 else 4
    sa = torch.randint(0, 2 * pad + 1, (Na, 2), generator=g, dtype=torch.int32).to(dev) if shift else None
    sb = torch.randint(0, 2 * pad + 1, (Nb, 2), generator=g, dtype=torch.int32).to(dev) if shift else None
    sab = torch.cat([sa, sb]) if shift else None
    w2d = ((torch.rand(32, 192, generator=g) * 2 - 1) / 192 ** 0.5).to(dev).to(torch.bfloat16)
    b = ((torch.rand(32, generator=g) * 2 - 1) * 0.1).to(dev)
    N = Na + Nb
    OH, OW = kn.conv_out_hw(H, W, 8, 8, 4)
    y1 = torch.zeros(N, OH, OW, 32, dtype=torch.bfloat16, device=dev)
    y2 = torch.zeros_like(y1)
    b1 = torch.zeros(N * OH * OW, dtype=torch.int32, device=dev)
    b2 = torch.zeros_like(b1)
    kn.conv2d_fwd(xa, w2d, b, y1[:Na], Na, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=b1[:Na * OH * OW], aug_shift=sa, aug_pad=pad)
    kn.conv2d_fwd(xb, w2d, b, y1[Na:], Nb, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=b1[Na * OH * OW:], aug_shift=sb, aug_pad=pad)
    kn.conv2d_fwd(xa, w2d, b, y2, N, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, relu_bits=b2, x2=xb, aug_shift=sab, aug_pad=pad)
    torch.cuda.synchronize()
    assert torch.equal(y1.view(torch.int16), y2.view(torch.int16)) and torch.equal(b1, b2)
    dy = torch.randn(N, OH, OW, 32, generator=g).to(dev).to(torch.bfloat16)
    dw1, db1 = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
    dw2, db2 = torch.zeros(32, 192, device=dev), torch.zeros(32, device=dev)
    kn.conv2d_bwd_weight(xa, dy[:Na].contiguous(), dw1, db1, Na, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, aug_shift=sa, aug_pad=pad)
    kn.conv2d_bwd_weight(xb, dy[Na:].contiguous(), dw1, db1, Nb, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, accumulate=True, aug_shift=sb, aug_pad=pad)
    kn.conv2d_bwd_weight(xa, dy, dw2, db2, N, H, W, 3, 32, 8, 8, 4, True, compute=kn.BF16, x2=xb, aug_shift=sab, aug_pad=pad)
    torch.cuda.synchronize()
    assert float((dw1 - dw2).abs().max()) <= 1e-5 * float(dw1.abs().max()) and float((db1 - db2).abs().max()) <= 1e-5 * float(db1.abs().max())


@pytest.mark.parametrize("name,N,H,W", [("static3", 300, 23, 23), ("static3-few", 3, 23, 23), ("grip3", 700, 9, 9), ("grip3-few", 5, 9, 9),
                                        ("odd3", 2, 11, 13)])
def test_conv3_stores_the_exact_map_and_the_bf16_map_from_one_launch(dev, name, N, H, W):
    """hulc_conv_desc.y_bf16 (ABI 7; precision site "a3"): an fp32-output forward that also leaves the bf16 map.  The fp32 map equals the plain
    fp32-output launch bit for bit, the bf16 map equals its rounding (= what a bf16-output launch of the same kernel family stores: ReLU and
    round-to-nearest commute); covered on the LDS-band kernels (frame-sized and packed-frame units) and on the path that follows the generic
    kernel with a cast launch (a map too small for a band unit)."""
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    Cin, Cout, K, s = 64, 64, 3, 1
    g = torch.Generator().manual_seed(17)
    x = torch.relu(torch.randn(N, H, W, Cin, generator=g)).to(torch.bfloat16).to(dev)
    w = (torch.randn(Cout, Cin, K, K, generator=g) / (Cin * K * K) ** 0.5)
    w2d = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(torch.bfloat16).to(dev)
    b = (torch.rand(Cout, generator=g) * 0.2 - 0.1).to(dev)
    OH, OW = kn.conv_out_hw(H, W, K, K, s)
    y32 = torch.empty(N, OH, OW, Cout, device=dev)
    kn.conv2d_fwd(x, w2d, b, y32, N, H, W, Cin, Cout, K, K, s, False)
    y32b = torch.full_like(y32, float("nan"))
    y16 = torch.full((N, OH, OW, Cout), float("nan"), device=dev, dtype=torch.bfloat16)
    kn.conv2d_fwd(x, w2d, b, y32b, N, H, W, Cin, Cout, K, K, s, False, y_bf16=y16)
    torch.cuda.synchronize()
    assert torch.equal(y32b, y32)
    assert torch.equal(y16, y32.to(torch.bfloat16))
    ref = F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), w.to(torch.bfloat16).double().to(dev), b.double(), stride=s)).permute(0, 2, 3, 1)
    assert (y32.double() - ref).abs().max().item() <= 2e-5 * max(1.0, ref.abs().max().item())
    with pytest.raises(Exception):                                           # the copy goes with an fp32 / fp16 output only
        kn.conv2d_fwd(x, w2d, b, y16, N, H, W, Cin, Cout, K, K, s, False, y_bf16=y16)
    # the fp16 twin (HULC_F16): served by the direct-to-LDS kernel for the static camera's conv3 only — the same accumulators rounded to half
    # precision next to the same bf16 map; refused (loudly) everywhere else
    yh = torch.full((N, OH, OW, Cout), float("nan"), device=dev, dtype=torch.float16)
    y16b = torch.full_like(y16, float("nan"))
    if (H, W) == (23, 23):
        kn.conv2d_fwd(x, w2d, b, yh, N, H, W, Cin, Cout, K, K, s, False, y_bf16=y16b)
        torch.cuda.synchronize()
        assert torch.equal(yh, y32.to(torch.float16)) and torch.equal(y16b, y16)
    else:
        with pytest.raises(Exception):
            kn.conv2d_fwd(x, w2d, b, yh, N, H, W, Cin, Cout, K, K, s, False, y_bf16=y16b)


def test_site_a3_keeps_the_stack_output_bf16_and_hands_out_its_exact_twin(dev, monkeypatch):
    """site "a3": the conv stack's differentiable output is the bf16 map (its gradient is never cast), the consumers of the VALUES get the exact
    map; without the site there is no twin"""
    from hulc2_amd import functional as HF, kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(3)
    x = (torch.rand(4, 3, 200, 200, generator=g) * 2 - 1).to(dev)
    ps = []
    for co, ci, k in ((32, 3, 8), (64, 32, 4), (64, 64, 3)):
        ps += [(torch.randn(co, ci, k, k, generator=g) / (ci * k * k) ** 0.5).to(dev).requires_grad_(), torch.zeros(co, device=dev).requires_grad_()]
    monkeypatch.setenv("HULC_FP32_SITES", "head,goal,encfc,txl,a3")
    a = HF.conv_stack(x, ps, grad_premasked=True)
    tw = HF.exact_map(a)
    assert a.dtype == torch.bfloat16 and tw is not None and tw.dtype == torch.float16 and not tw.requires_grad      # (the static camera's map: fp16 twin)
    assert float((a.float() - tw.float()).abs().max()) <= 2.0 ** -8 * float(tw.float().abs().max())                # both are roundings of one fp32 map
    a.backward(torch.ones_like(a))                                           # a bf16 gradient goes in as it is
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in ps)
    monkeypatch.setenv("HULC_FP32_SITES", "head,goal,encfc,txl")
    assert HF.exact_map(HF.conv_stack(x, ps, grad_premasked=True)) is None
    # a consumer that asks for the twin gets one whatever the sites say (the gripper camera's flatten-linear under "encfc"): fp32 for a small map
    xg = (torch.rand(6, 3, 84, 84, generator=g) * 2 - 1).to(dev)
    ag = HF.conv_stack(xg, ps, grad_premasked=True, exact_out=True)
    twg = HF.exact_map(ag)
    assert ag.dtype == torch.bfloat16 and twg is not None and twg.dtype == torch.float32 and torch.equal(ag, twg.to(torch.bfloat16))


@pytest.mark.parametrize("name,N,H,W,Cin,Cout,K,stride", [("static2", 300, 49, 49, 32, 64, 4, 2), ("static3", 300, 23, 23, 64, 64, 3, 1),
                                                          ("grip2", 700, 20, 20, 32, 64, 4, 2), ("grip3", 700, 9, 9, 64, 64, 3, 1)])
def test_weight_gradient_load_arrangements_give_the_same_bits(dev, name, N, H, W, Cin, Cout, K, stride, monkeypatch):
    """round 6: the next unit's loads leave wave by wave over the MFMA loop (default), as one burst in front of it (HULC_WB_BURST=1, round 5), or
    over a part of the loop (HULC_WB_STAG): WHEN a load is issued changes nothing about what is multiplied — dW and db bit for bit"""
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(23)
    x = torch.relu(torch.randn(N, H, W, Cin, generator=g)).to(torch.bfloat16).to(dev)
    OH, OW = kn.conv_out_hw(H, W, K, K, stride)
    dy = torch.randn(N, OH, OW, Cout, generator=g).to(torch.bfloat16).to(dev)
    outs = []
    for env in ({}, {"HULC_WB_BURST": "1"}, {"HULC_WB_STAG": "3"}):
        for k_ in ("HULC_WB_BURST", "HULC_WB_STAG"):
            monkeypatch.delenv(k_, raising=False)
        for k_, v_ in env.items():
            monkeypatch.setenv(k_, v_)
        dw = torch.full((Cout, Cin * K * K), float("nan"), device=dev)
        db = torch.full((Cout,), float("nan"), device=dev)
        kn.conv2d_bwd_weight(x, dy, dw, db, N, H, W, Cin, Cout, K, K, stride, False)
        torch.cuda.synchronize()
        outs.append((dw, db))
    for dw, db in outs[1:]:
        assert torch.equal(dw, outs[0][0]) and torch.equal(db, outs[0][1])
    assert torch.isfinite(outs[0][0]).all()
