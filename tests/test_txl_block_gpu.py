"""The posterior's transformer trunk as one launch per direction (csrc/txl_block.hip: position embedding + dropout -> L post-norm layers ->
mean over the sequence, one workgroup per sequence) against
  (a) plain PyTorch fp32: nn.Embedding + nn.TransformerEncoder + mean — the reference's own composition
      (plan_recognition_net.py:125-146) — dropout off,
  (b) the per-layer HIP launches (AddPosFn, TxlLayerFn, SeqMeanFn) on the same counter-RNG streams, dropout on: identical masks by construction,
      so outputs and every gradient agree to bf16 summation-order level.
Tolerances are bf16's and stated per assertion."""
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import functional as HF, kernels as kn  # noqa: E402

KEYS = HF._TXL_KEYS


@pytest.fixture(autouse=True)
def _plain_bf16_forward(monkeypatch):
    """the comparisons below are between launches of the SAME arithmetic (bf16 operands): the default selective-precision sites include
    "txl" (split-operand forward of the block launch); the test of that forward selects it itself"""
    monkeypatch.setenv("HULC_FP32_SITES", "head")


def _trunk(seed, L, p_drop, ff=2048):
    torch.manual_seed(seed)
    layer = torch.nn.TransformerEncoderLayer(128, 8, dim_feedforward=ff, dropout=p_drop)
    enc = torch.nn.TransformerEncoder(layer, num_layers=L, norm=None, enable_nested_tensor=False)
    pos = torch.nn.Embedding(40, 128)
    with torch.no_grad():
        for q in enc.parameters():                     # independent layers, non-trivial biases and LayerNorm parameters
            q.copy_(torch.randn_like(q) * (0.1 if q.dim() == 1 else 0.06))
            if q.dim() == 1 and q.numel() == 128:
                q.add_(0.5)
    return enc, pos


def _layer_params(enc):
    out = []
    for m in enc.layers:
        sd = dict(m.named_parameters())
        out.append({"in_proj_weight": sd["self_attn.in_proj_weight"], "in_proj_bias": sd["self_attn.in_proj_bias"],
                    "out_proj.weight": sd["self_attn.out_proj.weight"], "out_proj.bias": sd["self_attn.out_proj.bias"],
                    "linear1.weight": sd["linear1.weight"], "linear1.bias": sd["linear1.bias"], "linear2.weight": sd["linear2.weight"],
                    "linear2.bias": sd["linear2.bias"], "norm1.weight": sd["norm1.weight"], "norm1.bias": sd["norm1.bias"],
                    "norm2.weight": sd["norm2.weight"], "norm2.bias": sd["norm2.bias"]})
    return out


def _run(enc, pos, emb, r, p_drop, seed, block):
    for q in list(enc.parameters()) + list(pos.parameters()):
        q.grad = None
    B, S, E = emb.shape
    x = emb.clone().requires_grad_(True)
    ids = torch.arange(S, device=emb.device)
    layers = _layer_params(enc)
    if block:
        assert HF.txl_block_ok(x, layers, S, 8)
        pooled = HF.transformer_trunk_pooled(x, pos.weight, ids, layers, 8, p_drop, seed)
        assert type(pooled.grad_fn).__name__.startswith("TxlBlockFn")
    else:
        h = HF.AddPosFn.apply(x, pos.weight, ids, p_drop, seed, False).reshape(B * S, E)
        for li, p in enumerate(layers):
            h = HF.transformer_encoder_layer(h, p, B, S, 8, p_drop, seed + 100 * (li + 1))
        pooled = HF.SeqMeanFn.apply(h.reshape(B, S, E))
    (pooled * r).sum().backward()
    torch.cuda.synchronize()
    grads = {f"{li}.{k}": layers[li][k].grad.detach().clone() for li in range(len(layers)) for k in KEYS}
    grads["pos"] = pos.weight.grad.detach().clone()
    return pooled.detach(), x.grad.detach(), grads


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


@pytest.mark.parametrize("B,S,L", [(64, 32, 2), (3, 17, 2), (5, 32, 1), (2, 7, 3)])
def test_block_matches_torch_fp32(dev, B, S, L):
    kn.set_compute("bf16")
    enc, pos = _trunk(1, L, 0.0)
    g = torch.Generator().manual_seed(2)
    emb, r = torch.randn(B, S, 128, generator=g), torch.randn(B, 128, generator=g)
    xr = emb.clone().requires_grad_(True)
    h = xr + pos(torch.arange(S)).unsqueeze(0)
    yr = enc(h.transpose(0, 1)).transpose(0, 1).mean(dim=1)          # the reference runs (S, B, E), batch_first=False
    (yr * r).sum().backward()
    want = {f"{li}.{k}": p[k].grad.clone() for li, p in enumerate(_layer_params(enc)) for k in KEYS}
    want["pos"] = pos.weight.grad.clone()
    enc_d, pos_d = _trunk(1, L, 0.0)
    enc_d, pos_d = enc_d.to(dev), pos_d.to(dev)
    y, dx, got = _run(enc_d, pos_d, emb.to(dev), r.to(dev), 0.0, 77, block=True)
    yu, dxu, unf = _run(enc_d, pos_d, emb.to(dev), r.to(dev), 0.0, 77, block=False)
    e, eu = _rel(y, yr.detach()), _rel(yu, yr.detach())
    print(f"B={B} S={S} L={L} pooled: block {e:.2e} per-layer {eu:.2e}")
    assert e < 1.5e-2, e                                              # relative L2, bf16 operands (the per-layer launches: the same class)
    e, eu = _rel(dx, xr.grad), _rel(dxu, xr.grad)
    print(f"  demb: block {e:.2e} per-layer {eu:.2e}")
    assert e < max(4e-2, 1.5 * eu), (e, eu)                        # (two stacked bf16 layers with O(1) pre-activations: the per-layer launches sit at the same level)
    for k in want:
        e, eu = _rel(got[k], want[k]), _rel(unf[k], want[k])
        print(f"  {k}: block {e:.2e} per-layer {eu:.2e}")
        assert e < max(4e-2, 2.0 * eu), (k, e, eu)
    kn.check_faults(dev)


@pytest.mark.parametrize("B,S", [(64, 32), (4, 19)])
def test_block_matches_per_layer_launches_with_dropout(dev, B, S):
    """dropout 0.1: the block draws the per-layer kernels' streams (position add, attention probabilities, both residual branches, the
    hidden activation), so both paths see the same masks and differ by bf16 summation order only"""
    kn.set_compute("bf16")
    enc, pos = _trunk(3, 2, 0.1)
    enc, pos = enc.to(dev), pos.to(dev)
    g = torch.Generator().manual_seed(4)
    emb, r = torch.randn(B, S, 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
    y, dx, got = _run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    yu, dxu, unf = _run(enc, pos, emb, r, 0.1, 0x5EED0001, block=False)
    assert _rel(y, yu) < 6e-3, _rel(y, yu)
    assert _rel(dx, dxu) < 3e-2, _rel(dx, dxu)
    for k in got:
        assert _rel(got[k], unf[k]) < 3e-2, (k, _rel(got[k], unf[k]))
    # and the launch is deterministic
    y2, dx2, got2 = _run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    assert torch.equal(y, y2) and torch.equal(dx, dx2)
    kn.check_faults(dev)


def test_block_inference_keeps_nothing(dev):
    kn.set_compute("bf16")
    enc, pos = _trunk(5, 2, 0.0)
    enc, pos = enc.to(dev), pos.to(dev)
    emb = torch.randn(6, 32, 128, device=dev)
    with torch.no_grad():
        a = HF.transformer_trunk_pooled(emb, pos.weight, torch.arange(32, device=dev), _layer_params(enc), 8, 0.0, 1)
    b = HF.transformer_trunk_pooled(emb.clone().requires_grad_(True), pos.weight, torch.arange(32, device=dev), _layer_params(enc), 8, 0.0, 1)
    assert torch.equal(a, b.detach())


@pytest.mark.parametrize("B,S", [(64, 32), (11, 32)])
def test_shared_sequences_match_one_workgroup_per_sequence(dev, B, S, monkeypatch):
    """while B Q <= 256 the launch shares a sequence between Q = 4 workgroups (feed-forward hidden units split, partial tiles exchanged
    through memory): against one workgroup per sequence (HULC_TXL_NO_SHARE=1) the results differ by the summation order of four partial
    tiles only"""
    kn.set_compute("bf16")
    enc, pos = _trunk(7, 2, 0.1)
    enc, pos = enc.to(dev), pos.to(dev)
    g = torch.Generator().manual_seed(8)
    emb, r = torch.randn(B, S, 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
    y, dx, got = _run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    monkeypatch.setenv("HULC_TXL_NO_SHARE", "1")
    y1, dx1, got1 = _run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    assert _rel(y, y1) < 2e-3, _rel(y, y1)
    assert _rel(dx, dx1) < 2e-2, _rel(dx, dx1)
    for k in got:
        assert _rel(got[k], got1[k]) < 2e-2, (k, _rel(got[k], got1[k]))
    kn.check_faults(dev)


def test_one_workspace_serves_launches_of_any_batch_size(dev):
    """the shared-sequence launches of different B reuse one cached workspace: its counter area has a fixed size, so the partial tiles of a
    small-B launch never land where a large-B launch keeps its arrival counters (a B-dependent layout did: wrong gradients for the last
    sequences, one launch in forty)"""
    kn.set_compute("bf16")
    enc, pos = _trunk(3, 2, 0.1)
    enc, pos = enc.to(dev), pos.to(dev)
    want = {}
    for it in range(24):
        B, S = [(4, 19), (64, 32), (11, 32), (5, 7)][it % 4]
        g = torch.Generator().manual_seed(4)
        emb, r = torch.randn(B, S, 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
        y, dx, got = _run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
        if (B, S) not in want:
            want[(B, S)] = (y, dx, got)
            continue
        y0, dx0, got0 = want[(B, S)]
        assert torch.equal(y, y0) and torch.equal(dx, dx0), (it, B, S)
        for k in got:
            assert torch.equal(got[k], got0[k]), (it, B, S, k)
    kn.check_faults(dev)


@pytest.mark.parametrize("B,S", [(64, 32), (3, 17)])
def test_split_operand_forward_is_fp32_class(dev, B, S, monkeypatch):
    """selective-precision site "txl": the forward launch forms every product from hi / lo splits of both operands (three bf16 MFMAs).  Its
    pooled output agrees with fp32 torch to 1e-4 (plain bf16: 1.7e-3); the backward stays on the bf16 products (it recomputes the
    projections and the hidden tile in bf16: a backward-direction rounding), so the gradients keep their bf16 tolerances."""
    kn.set_compute("bf16")
    enc, pos = _trunk(1, 2, 0.0)
    g = torch.Generator().manual_seed(2)
    emb, r = torch.randn(B, S, 128, generator=g), torch.randn(B, 128, generator=g)
    xr = emb.clone().requires_grad_(True)
    h = xr + pos(torch.arange(S)).unsqueeze(0)
    yr = enc(h.transpose(0, 1)).transpose(0, 1).mean(dim=1)
    (yr * r).sum().backward()
    want = {f"{li}.{k}": p[k].grad.clone() for li, p in enumerate(_layer_params(enc)) for k in KEYS}
    enc_d, pos_d = _trunk(1, 2, 0.0)
    enc_d, pos_d = enc_d.to(dev), pos_d.to(dev)
    y0, dx0, got0 = _run(enc_d, pos_d, emb.to(dev), r.to(dev), 0.0, 77, block=True)
    monkeypatch.setenv("HULC_FP32_SITES", "head,txl")
    y, dx, got = _run(enc_d, pos_d, emb.to(dev), r.to(dev), 0.0, 77, block=True)
    e0, e = _rel(y0, yr.detach()), _rel(y, yr.detach())
    print(f"pooled vs fp32: bf16 {e0:.2e}, split operands {e:.2e}")
    assert e < 1e-4 and e < 0.1 * e0, (e, e0)
    assert _rel(dx, xr.grad) < max(4e-2, 1.5 * _rel(dx0, xr.grad))
    for k in want:
        assert _rel(got[k], want[k]) < max(4e-2, 2.0 * _rel(got0[k], want[k])), k
    kn.check_faults(dev)
