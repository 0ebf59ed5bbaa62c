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


@pytest.mark.parametrize("B,S", [(3, 1), (5, 4), (64, 6), (33, 3)])
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


def test_wavefront_deterministic(dev):
    h1, g1 = _run(dev, 64, 8, persistent=True, seed=3)
    h2, g2 = _run(dev, 64, 8, persistent=True, seed=3)
    assert torch.equal(h1, h2)
    for a, b in zip(g1, g2):
        assert torch.equal(a, b)
