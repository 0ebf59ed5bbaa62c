"""The persistent wavefront RNN kernel (csrc/rnn_wavefront.hip) against the per-step GEMM path of the same operator.

Both run bf16 MFMA with fp32 accumulation on the same bf16-rounded operands; they differ only in summation order and in
rounding the recurrent state to bf16 once (wavefront) instead of at every operand load (per-step) — the same values.
"""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


def _run(dev, B, S, persistent, seed=0):
    from hulc2_amd import functional as HF, kernels as kn

    kn.set_compute("bf16")
    H, P, E, G = 2048, 64, 40, 32
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, k=1.0: ((torch.rand(*s, generator=g) * 2 - 1) * k).to(dev)
    plan, emb, goal = r(B, P), r(B, S, E + 8), r(B, G)
    k0, k1 = (P + E + G) ** -0.5, H ** -0.5
    params = [r(H, P + E + G, k=k0), r(H, H, k=k1), r(H, k=k1), r(H, k=k1), r(H, H, k=k1), r(H, H, k=k1), r(H, k=k1), r(H, k=k1)]
    leaves = [t.requires_grad_(True) for t in (plan, emb, goal, *params)]
    if persistent:
        os.environ.pop("HULC_NO_RNN_WAVEFRONT", None)
    else:
        os.environ["HULC_NO_RNN_WAVEFRONT"] = "1"
    try:
        h1 = HF.DecoderRNNFn.apply(leaves[0], leaves[1], leaves[2], 4, 4 + E, *leaves[3:])
        w = r(B, S, H)
        (h1 * w).sum().backward()
        torch.cuda.synchronize()
    finally:
        os.environ.pop("HULC_NO_RNN_WAVEFRONT", None)
    return h1.detach(), [t.grad for t in leaves]


@pytest.mark.parametrize("B,S", [(3, 1), (5, 4), (64, 6), (33, 3), (16, 5)])
def test_wavefront_matches_per_step(dev, B, S):
    h_ref, g_ref = _run(dev, B, S, persistent=False)
    h, g = _run(dev, B, S, persistent=True)
    assert torch.isfinite(h).all(), "wavefront kernel: non-finite state (barrier timeout poisons with NaN)"
    err = (h - h_ref).abs().max().item() / h_ref.abs().max().item()
    assert err < 2e-3, f"h1 rel max err {err:.3e}"
    names = ["plan", "emb", "goal", "w_ih0", "w_hh0", "b_ih0", "b_hh0", "w_ih1", "w_hh1", "b_ih1", "b_hh1"]
    for n, a, b in zip(names, g, g_ref):
        e = ((a - b).norm() / b.norm().clamp_min(1e-12)).item()
        assert e < 2e-2, f"grad {n}: rel L2 err {e:.3e}"     # ReLU-mask flips on bf16-level state differences


def test_transposed_mirror_operands_give_the_same_gradients(dev, monkeypatch):
    """HULC_RNN_WGRAD_TMIRROR=1: the weight gradients from the transposed mirrors (k-major GEMMs) against the default row-major ones"""
    _, g0 = _run(dev, 64, 5, persistent=True, seed=2)
    monkeypatch.setenv("HULC_RNN_WGRAD_TMIRROR", "1")
    _, g1 = _run(dev, 64, 5, persistent=True, seed=2)
    for a, b in zip(g0, g1):
        assert ((a - b).norm() / b.norm().clamp_min(1e-12)).item() < 1e-5


def test_wavefront_deterministic(dev):
    h1, g1 = _run(dev, 64, 8, persistent=True, seed=3)
    h2, g2 = _run(dev, 64, 8, persistent=True, seed=3)
    assert torch.equal(h1, h2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)


@pytest.mark.parametrize("B,transposed", [(64, False), (24, True)])
def test_mirrors_agree_with_the_state_rows(dev, B, transposed):
    """the row-major and the transposed bf16 mirror the kernel leaves behind are the bf16 rounding of its fp32 state rows"""
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    S, H = 5, 2048
    g = torch.Generator().manual_seed(B)
    w = [((torch.rand(H, H, generator=g) * 2 - 1) * H ** -0.5).to(dev).to(torch.bfloat16) for _ in range(3)]
    pre0 = torch.randn(S, B, H, generator=g).to(dev)
    zbuf = torch.zeros(S + 2, B, 2 * H, device=dev)
    if transposed:                                                   # the backward sweep: rows S+1 -> 0
        z16, z16t = kn.rnn_wavefront(zbuf[S + 1], -B * 2 * H, S, B, H, w[0], w[1], w[2], True, add1=pre0[S - 1], add1_step=-B * H, ld_add1=H, mirror_t=True)
    else:
        z16, z16t = kn.rnn_wavefront(zbuf[0], B * 2 * H, S, B, H, w[0], w[1], w[2], False, add1=pre0, add1_step=B * H, ld_add1=H, relu=True, mirror_t=True)
    torch.cuda.synchronize()
    assert z16t is not None and tuple(z16t.shape) == (2 * H, (S + 2) * B)
    lo, hi = (0, S) if transposed else (1, S + 1)                    # rows the sweep wrote completely
    want = zbuf[lo:hi + 1].to(torch.bfloat16)
    got, got_t = z16[lo:hi + 1].clone(), z16t.view(2 * H, S + 2, B).permute(1, 2, 0)[lo:hi + 1].clone()
    skip = 0 if transposed else -1                                   # the half of the sweep's last row that is never produced (never read either)
    for t in (want, got, got_t):
        t[skip, :, :H] = 0
    assert torch.equal(got, want)
    assert torch.equal(got_t, got)
    start = S + 1 if transposed else 0
    assert not z16t.view(2 * H, S + 2, B)[:, start].any(), "initial state row of the transposed mirror is zero"
