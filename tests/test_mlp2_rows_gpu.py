"""The camera encoders' head fc2(relu(fc1(x))) as one launch per direction (csrc/mlp2_rows.hip) against (a) plain PyTorch fp32 and (b) the
two-GEMM path of HF.mlp (HULC_NO_MLP2_ROWS=1): outputs, input gradient and all four parameter gradients.  Tolerances are bf16's (operands
rounded to bf16, fp32 accumulation) and stated per assertion."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import functional as HF, kernels as kn  # noqa: E402


def _rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / (b.norm() + 1e-30)).item()


def _run(fc1, fc2, x, r, need_x=True):
    for q in list(fc1.parameters()) + list(fc2.parameters()):
        q.grad = None
    xd = x.clone().requires_grad_(need_x)
    y = HF.mlp2_rows(xd, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
    (y * r).sum().backward()
    torch.cuda.synchronize()
    return y, xd.grad, [fc1.weight.grad.clone(), fc1.bias.grad.clone(), fc2.weight.grad.clone(), fc2.bias.grad.clone()]


@pytest.mark.parametrize("T,H,OUT", [(2048, 512, 64), (77, 512, 64), (1, 128, 32), (300, 256, 128), (64, 384, 96)])
def test_head_matches_torch_and_the_gemm_path(dev, T, H, OUT, monkeypatch):
    kn.set_compute("bf16")
    torch.manual_seed(T + H)
    fc1, fc2 = torch.nn.Linear(128, H), torch.nn.Linear(H, OUT)
    x, r = torch.randn(T, 128), torch.randn(T, OUT)
    xr = x.clone().requires_grad_(True)
    yr = fc2(torch.relu(fc1(xr)))
    (yr * r).sum().backward()
    want = [fc1.weight.grad.clone(), fc1.bias.grad.clone(), fc2.weight.grad.clone(), fc2.bias.grad.clone()]
    fc1, fc2 = fc1.to(dev), fc2.to(dev)
    assert kn.mlp2_rows_ok(x.to(dev), fc1.weight, fc2.weight)
    y, dx, got = _run(fc1, fc2, x.to(dev), r.to(dev))
    assert type(y.grad_fn).__name__.startswith("Mlp2RowsFn")
    monkeypatch.setenv("HULC_NO_MLP2_ROWS", "1")
    yg, dxg, gem = _run(fc1, fc2, x.to(dev), r.to(dev))
    assert type(yg.grad_fn).__name__.startswith("MLPFn")
    # both HIP paths round the same operands to bf16 and accumulate in fp32: they differ by summation order and by the bf16 rounding of
    # the hidden activation as a weight-gradient operand
    assert _rel(y, yg) < 3e-3, _rel(y, yg)
    assert _rel(dx, dxg) < 6e-3, _rel(dx, dxg)
    for a, b, name in zip(got, gem, ("dW1", "db1", "dW2", "db2")):
        assert _rel(a, b) < 8e-3, (name, _rel(a, b))
    # against fp32 torch: the bf16 error of two stacked products
    assert _rel(y, yr.detach()) < 8e-3, _rel(y, yr.detach())
    # (gradients: ReLU sign flips of pre-activations near zero dominate — the GEMM path sits at the same level)
    assert _rel(dx, xr.grad) < max(1.5e-2, 1.3 * _rel(dxg, xr.grad)), (_rel(dx, xr.grad), _rel(dxg, xr.grad))
    for a, b, g, name in zip(got, want, gem, ("dW1", "db1", "dW2", "db2")):
        assert _rel(a, b) < max(1.5e-2, 1.3 * _rel(g, b)), (name, _rel(a, b), _rel(g, b))


def test_head_without_input_gradient_and_determinism(dev):
    kn.set_compute("bf16")
    torch.manual_seed(0)
    fc1, fc2 = torch.nn.Linear(128, 512).to(dev), torch.nn.Linear(512, 64).to(dev)
    x, r = torch.randn(500, 128, device=dev), torch.randn(500, 64, device=dev)
    y0, dx0, g0 = _run(fc1, fc2, x, r, need_x=False)
    assert dx0 is None
    y1, dx1, g1 = _run(fc1, fc2, x, r)
    y2, dx2, g2 = _run(fc1, fc2, x, r)
    assert torch.equal(y1, y2) and torch.equal(dx1, dx2) and torch.equal(y0, y1)
    for a, b, c in zip(g0, g1, g2):
        assert torch.equal(a, b) and torch.equal(b, c)


def test_split_operand_forward_inside_an_exact_site(dev, monkeypatch):
    """inside kernels.site_scope("encfc") of a bf16 step the head launch forms its products from hi / lo splits of both operands: output
    within 1e-5 of fp32 torch (plain bf16: ~3e-3), gradients at their bf16 level (the backward recomputes the hidden tile in bf16)"""
    kn.set_compute("bf16")
    monkeypatch.setenv("HULC_FP32_SITES", "head,encfc")
    torch.manual_seed(5)
    fc1, fc2 = torch.nn.Linear(128, 512), torch.nn.Linear(512, 64)
    x, r = torch.randn(777, 128), torch.randn(777, 64)
    xr = x.clone().requires_grad_(True)
    yr = fc2(torch.relu(fc1(xr)))
    (yr * r).sum().backward()
    fc1, fc2 = fc1.to(dev), fc2.to(dev)
    y0, dx0, g0 = _run(fc1, fc2, x.to(dev), r.to(dev))
    with kn.site_scope("encfc"):
        y, dx, g = _run(fc1, fc2, x.to(dev), r.to(dev))
    assert type(y.grad_fn).__name__.startswith("Mlp2RowsFn")
    e0, e = _rel(y0, yr.detach()), _rel(y, yr.detach())
    assert e < 1e-5 and e < 0.01 * e0, (e, e0)
    assert _rel(dx, xr.grad) < max(1.5e-2, 1.3 * _rel(dx0, xr.grad))
