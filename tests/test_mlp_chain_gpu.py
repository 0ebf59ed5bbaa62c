"""The persistent MLP-chain kernel (csrc/mlp_chain.hip) against (a) fp64 torch on bf16-rounded operands' own arithmetic bound and (b) the
per-layer GEMM path of the same Function (HULC_NO_MLP_CHAIN=1) — forward values, input gradient, every weight / bias gradient — on the
shapes the policy uses: the prior (160 -> 4 x 2048 -> 1024, 64 rows), the goal encoders (128 / 384 -> 2048 -> 2048 -> 32, 32 rows), the
contrastive projections (4096 -> 128 -> 32, 32 -> 128 -> 32) and ragged row counts."""
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import functional as HF, kernels as kn  # noqa: E402

SHAPES = [(64, (160, 2048, 2048, 2048, 2048, 1024)), (32, (128, 2048, 2048, 32)), (32, (384, 2048, 2048, 32)), (32, (4096, 128, 32)),
          (32, (32, 128, 32)), (5, (160, 2048, 1024)), (2, (128, 2048, 2048, 32)), (17, (384, 64, 48))]


def _run(M, dims, seed, chain, need_x=True):
    torch.manual_seed(seed)
    ws = [torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])]
    x, r = torch.randn(M, dims[0]), torch.randn(M, dims[-1])
    dev = torch.device("cuda", 0)
    import copy
    wsd = [copy.deepcopy(l).to(dev) for l in ws]
    xd = x.to(dev).requires_grad_(need_x)
    if chain:
        os.environ.pop("HULC_NO_MLP_CHAIN", None)
    else:
        os.environ["HULC_NO_MLP_CHAIN"] = "1"
    try:
        y = HF.mlp(xd, [(l.weight, l.bias, i < len(wsd) - 1) for i, l in enumerate(wsd)])
        (y * r.to(dev)).sum().backward()
    finally:
        os.environ.pop("HULC_NO_MLP_CHAIN", None)
    torch.cuda.synchronize()
    return y.detach(), (xd.grad if need_x else None), [l.weight.grad for l in wsd], [l.bias.grad for l in wsd], (ws, x, r)


def _rel(a, b):
    return ((a.double().cpu() - b.double().cpu()).norm() / (b.double().cpu().norm() + 1e-30)).item()


@pytest.mark.parametrize("M,dims", SHAPES)
def test_chain_matches_per_layer_gemms(dev, M, dims):
    kn.set_compute("bf16")
    if not kn.mlp_chain_ok(M, dims[0], list(dims[1:]), dev):
        pytest.skip("shape not taken by the chain kernel on this device")
    yc, dxc, dWc, dbc, _ = _run(M, dims, 7, chain=True)
    yg, dxg, dWg, dbg, (ws, x, r) = _run(M, dims, 7, chain=False)
    # forward: both paths round operands to bf16 and accumulate in fp32 — they differ by summation order and by the bf16 rounding of the
    # exchanged hidden activations (the GEMM path rounds them while staging, the chain when it stores them): same values, 1e-3 class
    assert _rel(yc, yg) < 4e-3, _rel(yc, yg)
    # against fp32 torch: the bf16 error of the stack (3e-3 per layer, grows with depth; ReLU sign flips dominate the gradients)
    h = x.clone().requires_grad_(True)
    t = h
    for i, l in enumerate(ws):
        t = l(t)
        if i < len(ws) - 1:
            t = torch.relu(t)
    (t * r).sum().backward()
    e_c, e_g = _rel(yc, t.detach()), _rel(yg, t.detach())
    assert e_c < max(1.5 * e_g, 6e-3), (e_c, e_g)
    e_c, e_g = _rel(dxc, h.grad), _rel(dxg, h.grad)
    assert e_c < max(1.6 * e_g, 2e-2), ("dx", e_c, e_g)
    for i, l in enumerate(ws):
        e_c, e_g = _rel(dWc[i], l.weight.grad), _rel(dWg[i], l.weight.grad)
        assert e_c < max(1.6 * e_g, 2e-2), ("dW", i, e_c, e_g)
        e_c, e_g = _rel(dbc[i], l.bias.grad), _rel(dbg[i], l.bias.grad)
        assert e_c < max(1.6 * e_g, 2e-2), ("db", i, e_c, e_g)


def test_chain_is_deterministic_and_skips_dx_when_not_needed(dev):
    kn.set_compute("bf16")
    M, dims = 64, (160, 2048, 2048, 1024)
    if not kn.mlp_chain_ok(M, dims[0], list(dims[1:]), dev):
        pytest.skip("chain kernel not available")
    a = _run(M, dims, 3, chain=True)
    b = _run(M, dims, 3, chain=True)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and all(torch.equal(p, q) for p, q in zip(a[2], b[2]))
    c = _run(M, dims, 3, chain=True, need_x=False)
    assert torch.equal(a[0], c[0]) and all(torch.equal(p, q) for p, q in zip(a[2], c[2])) and c[1] is None


def test_chain_rows_do_not_mix(dev):
    """rows are independent samples: changing one input row changes only that output row (the 64-row exchange layout is per row)"""
    kn.set_compute("bf16")
    dims = (128, 2048, 2048, 32)
    if not kn.mlp_chain_ok(33, dims[0], list(dims[1:]), dev):
        pytest.skip("chain kernel not available")
    torch.manual_seed(0)
    ws = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(33, dims[0], device=dev)
    layers = [(l.weight, l.bias, i < len(ws) - 1) for i, l in enumerate(ws)]
    with torch.no_grad():
        y1 = HF.mlp(x, layers)
        x2 = x.clone()
        x2[20] += 1.0
        y2 = HF.mlp(x2, layers)
    same = [i for i in range(33) if torch.equal(y1[i], y2[i])]
    assert same == [i for i in range(33) if i != 20]


@pytest.mark.parametrize("Ma,Mb,need_b", [(32, 32, False), (32, 32, True), (5, 17, False), (32, 1, False)])
def test_paired_chains_match_the_single_chain(dev, Ma, Mb, need_b):
    """hulc_mlp_chain2 (the visual + the language goal encoder as one launch, and their data-gradient chains as one) against the same two
    stacks through hulc_mlp_chain one after the other: every output, input gradient and parameter gradient BIT-identical (same tiles, same
    k split, same reduction order — only the launch is shared)."""
    import copy
    kn.set_compute("bf16")
    da, db_ = (128, 2048, 2048, 32), (384, 2048, 2048, 32)
    if not kn.mlp_chain2_ok(Ma, da[0], list(da[1:]), Mb, db_[0], list(db_[1:]), dev):
        pytest.skip("shape not taken by the paired chain kernel on this device")
    torch.manual_seed(3)
    la = [torch.nn.Linear(a, b).to(dev) for a, b in zip(da[:-1], da[1:])]
    lb = [torch.nn.Linear(a, b).to(dev) for a, b in zip(db_[:-1], db_[1:])]
    xa0, xb0 = torch.randn(Ma, da[0], device=dev), torch.randn(Mb, db_[0], device=dev)
    ra, rb = torch.randn(Ma, 32, device=dev), torch.randn(Mb, 32, device=dev)

    def layers(ls):
        return [(l.weight, l.bias, i < len(ls) - 1) for i, l in enumerate(ls)]

    res = []
    for paired in (True, False):
        A, Bm = copy.deepcopy(la), copy.deepcopy(lb)
        xa, xb = xa0.clone().requires_grad_(True), xb0.clone().requires_grad_(need_b)
        if paired:
            ya, yb = HF.dual_mlp(xa, layers(A), xb, layers(Bm))
            assert type(ya.grad_fn).__name__.startswith("DualMLPFn"), type(ya.grad_fn).__name__
        else:
            ya, yb = HF.mlp(xa, layers(A)), HF.mlp(xb, layers(Bm))
        ((ya * ra).sum() + (yb * rb).sum()).backward()
        torch.cuda.synchronize()
        res.append([ya.detach(), yb.detach(), xa.grad, xb.grad if need_b else None] + [p.grad for l in A + Bm for p in (l.weight, l.bias)])
    for i, (p, q) in enumerate(zip(*res)):
        if p is None:
            assert q is None
            continue
        assert torch.equal(p, q), (i, (p - q).abs().max().item())
    kn.check_faults(dev)


def test_paired_chain_falls_back_when_rows_do_not_fit(dev):
    kn.set_compute("bf16")
    la = [torch.nn.Linear(128, 64).to(dev), torch.nn.Linear(64, 32).to(dev)]
    lb = [torch.nn.Linear(384, 64).to(dev), torch.nn.Linear(64, 32).to(dev)]
    xa, xb = torch.randn(40, 128, device=dev, requires_grad=True), torch.randn(8, 384, device=dev)
    ya, yb = HF.dual_mlp(xa, [(la[0].weight, la[0].bias, True), (la[1].weight, la[1].bias, False)],
                         xb, [(lb[0].weight, lb[0].bias, True), (lb[1].weight, lb[1].bias, False)])
    assert type(ya.grad_fn).__name__.startswith("MLPFn")
    (ya.sum() + yb.sum()).backward()
    assert xa.grad is not None and lb[0].weight.grad is not None


@pytest.mark.parametrize("Ma,Mb", [(32, 32), (7, 19)])
def test_paired_chain_with_an_exact_second_stack(dev, Ma, Mb):
    """hulc_mlp_chain_layer.W_lo (precision site "goal"): the SECOND chain of the paired launch forms its products from hi / lo splits of
    both operands (three MFMAs) — its output agrees with fp32 torch to 2e-5 (plain bf16: ~3e-3) while the first chain's output is the
    plain launch's, bit for bit; gradients come from the bf16 data-gradient chains as before"""
    import copy
    kn.set_compute("bf16")
    da, db_ = (128, 2048, 2048, 32), (384, 2048, 2048, 32)
    torch.manual_seed(9)
    la = [torch.nn.Linear(a, b) for a, b in zip(da[:-1], da[1:])]
    lb = [torch.nn.Linear(a, b) for a, b in zip(db_[:-1], db_[1:])]
    xa0, xb0 = torch.randn(Ma, da[0]), torch.randn(Mb, db_[0])
    h = xb0
    for i, l in enumerate(lb):
        h = l(h)
        if i < len(lb) - 1:
            h = torch.relu(h)
    want_b = h.detach()
    la, lb = [l.to(dev) for l in la], [l.to(dev) for l in lb]
    layers = lambda ls: [(l.weight, l.bias, i < len(ls) - 1) for i, l in enumerate(ls)]
    xa, xb = xa0.to(dev).requires_grad_(True), xb0.to(dev)
    ya0, yb0 = HF.dual_mlp(xa, layers(la), xb, layers(lb))
    pair = HF.dual_mlp(xa, layers(la), xb, layers(lb), exact_b=True)
    assert pair is not None
    ya, yb = pair
    assert torch.equal(ya, ya0)
    e0, e = _rel(yb0, want_b), _rel(yb, want_b)
    print(f"second stack vs fp32: bf16 {e0:.2e}, split operands {e:.2e}")
    assert e < 2e-5 and e < 0.02 * e0, (e, e0)
    (ya.sum() + yb.sum()).backward()
    assert xa.grad is not None and lb[0].weight.grad is not None and torch.isfinite(lb[0].weight.grad).all()
    kn.check_faults(dev)
