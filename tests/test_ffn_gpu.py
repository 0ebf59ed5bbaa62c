"""Fused transformer feed-forward block (csrc/ffn_fused.hip) against the unfused GEMM chain of the same operator: same bf16
operands, same dropout masks (one counter-RNG stream), different summation order only."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _run(dev, T, drop_p, fused, seed=0):
    from hulc2_amd import functional as HF, kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev)
    x = r(T, 128).requires_grad_(True)
    W1, b1 = r(2048, 128, k=128 ** -0.5).requires_grad_(True), r(2048, k=0.1).requires_grad_(True)
    W2, b2 = r(128, 2048, k=2048 ** -0.5).requires_grad_(True), r(128, k=0.1).requires_grad_(True)
    w = r(T, 128)
    if fused:
        f = HF.FFNFn.apply(x, W1, b1, W2, b2, drop_p, 1234)
    else:
        f = HF.mlp(x, [(W1, b1, True), (W2, b2, False)], drops=[drop_p, 0.0], seed=1234)
    (f * w).sum().backward()
    torch.cuda.synchronize()
    return f.detach(), [t.grad for t in (x, W1, b1, W2, b2)]


@pytest.mark.parametrize("T,drop_p", [(2048, 0.1), (2048, 0.0), (100, 0.1), (64 * 40, 0.1)])
def test_fused_ffn_matches_gemm_chain(dev, T, drop_p):
    f_ref, g_ref = _run(dev, T, drop_p, fused=False)
    f, g = _run(dev, T, drop_p, fused=True)
    assert torch.isfinite(f).all()
    err = (f - f_ref).abs().max().item() / f_ref.abs().max().item()
    assert err < 2e-3, f"f rel max err {err:.3e}"
    for n, a, b in zip(["x", "W1", "b1", "W2", "b2"], g, g_ref):
        e = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
        assert e < 1e-2, f"grad {n}: rel L2 err {e:.3e}"


def test_fused_ffn_against_fp64(dev):
    """and against a float64 evaluation on the bf16-rounded operands (no dropout)"""
    T = 512
    f, g = _run(dev, T, 0.0, fused=True, seed=5)
    gen = torch.Generator().manual_seed(5)
    r = lambda *s, k=1.0: ((torch.rand(*s, generator=gen) * 2 - 1) * k)
    x = r(T, 128)
    W1, b1 = r(2048, 128, k=128 ** -0.5), r(2048, k=0.1)
    W2, b2 = r(128, 2048, k=2048 ** -0.5), r(128, k=0.1)
    rb = lambda t: t.to(torch.bfloat16).double()
    h = torch.relu(rb(x) @ rb(W1).t() + b1.double())
    want = rb(h.float()) @ rb(W2).t() + b2.double()
    err = (f.double().cpu() - want).abs().max().item() / want.abs().max().item()
    assert err < 3e-3, f"f vs fp64 on bf16-rounded operands: {err:.3e}"


def test_fused_ffn_deterministic(dev):
    f1, g1 = _run(dev, 2048, 0.1, fused=True, seed=2)
    f2, g2 = _run(dev, 2048, 0.1, fused=True, seed=2)
    assert torch.equal(f1, f2) and all(torch.equal(a, b) for a, b in zip(g1, g2))
