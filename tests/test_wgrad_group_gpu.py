"""The grouped weight-gradient launch (csrc/wgrad_group.hip, kernels.wgrad / wgrad_flush) against fp64 torch on the bf16-rounded operands
(what the MFMA multiplies; the bias sums against the unrounded values): every product shape of a training step, ragged tiles, K = 32 and a
K that is not a multiple of the 64-deep step, fp32 / bf16 / strided operands, accumulation, more items than one launch holds, a second
writer of the same destination, bit-identical repeats — and the deferred path through autograd with the trainer's gradient sinks."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import functional as HF, gradsink, kernels as kn  # noqa: E402

# (M, N, K) of the weight gradients of one step of the benchmark configuration (bench.py --breakdown), then edge cases
STEP_SHAPES = [(2048, 2048, 64), (512, 128, 2048), (384, 128, 2048), (64, 512, 2048), (128, 128, 2048), (2048, 2048, 32), (32, 128, 32),
               (184, 2048, 2048), (128, 3136, 2048), (2048, 64, 2048), (1024, 2048, 64), (32, 2048, 32), (128, 32, 32), (1024, 4096, 64),
               (2048, 160, 64), (128, 4096, 32), (2048, 384, 32), (2048, 128, 32), (4096, 128, 64), (2048, 1024, 64), (2048, 32, 64)]
EDGE_SHAPES = [(8, 8, 32), (72, 200, 96), (64, 64, 1056), (200, 8, 4128), (16, 24, 160)]


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    kn.set_compute("bf16")
    return torch.device("cuda", 0)


def _ref(A, B):
    a, b = A.to(torch.bfloat16).double(), B.to(torch.bfloat16).double()
    return a.t() @ b, A.double().sum(0)


def _problem(M, N, K, dev, a_bf16, b_bf16, pad=0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed * 7919 + M * 31 + N * 17 + K)
    A = torch.randn(K, M + pad, generator=g).to(dev)
    B = torch.randn(K, N + pad, generator=g).to(dev)
    if a_bf16:
        A = A.to(torch.bfloat16)
    if b_bf16:
        B = B.to(torch.bfloat16)
    return A[:, :M], B[:, :N]


def _check(C, rs, A, B, K, base_C=None, base_rs=None):
    want_C, want_rs = _ref(A, B)
    if base_C is not None:
        want_C, want_rs = want_C + base_C.double(), want_rs + base_rs.double()
    tol = 2e-5 * (K ** 0.5) + 1e-5                     # fp32 accumulation of K unit-variance products
    assert (C.double() - want_C).abs().max().item() <= tol * 4, (C.double() - want_C).abs().max().item()
    assert (rs.double() - want_rs).abs().max().item() <= tol * 4


@pytest.mark.parametrize("a_bf16,b_bf16", [(False, False), (True, False), (False, True), (True, True)])
def test_step_shapes_in_one_call(a_bf16, b_bf16):
    dev = _dev()
    probs = []
    for i, (M, N, K) in enumerate(STEP_SHAPES + EDGE_SHAPES):
        A, B = _problem(M, N, K, dev, a_bf16, b_bf16, pad=8 if i % 3 == 0 else 0)
        C = torch.full((M, N), float("nan"), device=dev)
        rs = torch.full((M,), float("nan"), device=dev)
        probs.append((A, B, C, rs, M, N, K))
    for A, B, C, rs, M, N, K in probs:                 # queued by hand: defer outside autograd flushes at once, so fill the queue directly
        kn._wg_pending.setdefault(dev, []).append((A, B, C, rs, M, N, K, A.stride(0), B.stride(0), N, False, False))
    kn.wgrad_flush(dev)                                # 26 items: two launches
    torch.cuda.synchronize()
    for A, B, C, rs, M, N, K in probs:
        _check(C, rs, A, B, K)


def test_accumulate_and_repeat_bitwise():
    dev = _dev()
    out = []
    for rep in range(3):
        res = []
        for (M, N, K) in [(184, 2048, 2048), (128, 3136, 2048), (512, 128, 2048), (2048, 2048, 64)]:
            A, B = _problem(M, N, K, dev, False, True)
            C0 = torch.randn(M, N, generator=torch.Generator().manual_seed(1)).to(dev)
            r0 = torch.randn(M, generator=torch.Generator().manual_seed(2)).to(dev)
            C, rs = C0.clone(), r0.clone()
            kn.wgrad(A, B, C, M, N, K, A.stride(0), B.stride(0), N, accumulate=True, rowsum=rs, rowsum_accumulate=True)
            torch.cuda.synchronize()
            if rep == 0:
                _check(C, rs, A, B, K, C0, r0)
            res.append((C, rs))
        out.append(res)
    for res in out[1:]:
        for (C, rs), (C1, r1) in zip(res, out[0]):
            assert torch.equal(C, C1) and torch.equal(rs, r1)          # slabs are summed in slice order whoever arrives last


def test_second_writer_of_a_destination_keeps_order():
    dev = _dev()
    M, N, K = 128, 256, 512
    A1, B1 = _problem(M, N, K, dev, False, False, seed=1)
    A2, B2 = _problem(M, N, K, dev, False, False, seed=2)
    C = torch.full((M, N), float("nan"), device=dev)
    rs = torch.full((M,), float("nan"), device=dev)
    q = kn._wg_pending.setdefault(dev, [])
    q.append((A1, B1, C, rs, M, N, K, M, N, N, False, False))          # first writer overwrites ...
    kn.wgrad(A2, B2, C, M, N, K, M, N, N, accumulate=True, rowsum=rs, rowsum_accumulate=True)     # ... the second adds: flushed in between
    torch.cuda.synchronize()
    w1, r1 = _ref(A1, B1)
    w2, r2 = _ref(A2, B2)
    assert (C.double() - (w1 + w2)).abs().max().item() < 2e-3 and (rs.double() - (r1 + r2)).abs().max().item() < 2e-3


def test_column_permutation_of_a_flattened_map():
    """col_perm = C: the product's (h*w, c) columns land in the parameter's (c, h*w) order (vision_network_gripper.py:16-17)"""
    dev = _dev()
    M, HW, C, K = 128, 49, 64, 2048
    N = HW * C
    A, B = _problem(M, N, K, dev, False, False)
    A, B = A.contiguous(), B.contiguous()
    out = torch.full((M, N), float("nan"), device=dev)
    rs = torch.empty(M, device=dev)
    kn.wgrad(A, B, out, M, N, K, M, N, N, rowsum=rs, col_perm=C)
    torch.cuda.synchronize()
    want, _ = _ref(A, B)
    want = want.view(M, HW, C).transpose(1, 2).reshape(M, N)
    assert (out.double() - want).abs().max().item() < 4e-3


def test_rejected_shapes_take_the_gemm_path():
    dev = _dev()
    M, N, K = 60, 100, 48                              # not multiples of 8 / 32
    A, B = _problem(M, N, K, dev, False, False)
    A, B = A.contiguous(), B.contiguous()
    C, rs = torch.empty(M, N, device=dev), torch.empty(M, device=dev)
    assert not kn.wgrad_group_ok(A, B, C, M, N, K, M, N, N)
    kn.wgrad(A, B, C, M, N, K, M, N, N, rowsum=rs)
    torch.cuda.synchronize()
    want_C, want_rs = _ref(A, B)
    assert (C.double() - want_C).abs().max().item() < 1e-3 and (rs.double() - want_rs).abs().max().item() < 1e-3


def test_deferred_through_autograd_with_sinks():
    """MLP weight / bias gradients land in registered arena slices at the end of backward() — and equal the immediate path's."""
    import copy
    dev = _dev()
    torch.manual_seed(3)
    dims = (160, 2048, 512, 64)
    ls = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    ref = copy.deepcopy(ls)
    x = torch.randn(64, dims[0], device=dev)
    r = torch.randn(64, dims[-1], device=dev)

    def run(layers):
        y = HF.mlp(x, [(l.weight, l.bias, i < len(layers) - 1) for i, l in enumerate(layers)])
        (y * r).sum().backward()
        torch.cuda.synchronize()

    run(ref)                                           # no sinks: gradients come back through autograd
    total = sum(p.numel() for l in ls for p in (l.weight, l.bias))
    arena = torch.full((total,), float("nan"), device=dev)
    off = 0
    try:
        for l in ls:
            for p in (l.weight, l.bias):
                gradsink.register(p, arena[off:off + p.numel()].view_as(p))
                off += p.numel()
        gradsink.begin_step(True)                      # first write of the step overwrites
        before = len(kn._wg_pending.get(dev, []))
        run(ls)
        assert before == 0 and not kn._wg_pending.get(dev)             # the end-of-pass callback issued the launch
        off = 0
        for l, lr in zip(ls, ref):
            for p, pr in ((l.weight, lr.weight), (l.bias, lr.bias)):
                got = arena[off:off + p.numel()].view_as(p)
                off += p.numel()
                assert p.grad is None
                assert torch.isfinite(got).all()
                assert (got - pr.grad).abs().max().item() <= 1e-4 * max(pr.grad.abs().max().item(), 1.0)
    finally:
        gradsink.clear()
        gradsink.begin_step(False)


def test_shared_layer_at_two_row_counts_under_sinks():
    """one Linear applied twice in a pass with different row counts (lang B = 32 rows through the grouped launch, vis B = 20 rows:
    K % 32 != 0, the immediate GEMM path).  Backward reaches the 20-row use... whichever comes first, the deferred overwrite of the other must
    not swallow the accumulated contribution: arena slice == sum of both contributions."""
    dev = _dev()
    torch.manual_seed(11)
    for first_rows, second_rows in ((32, 20), (20, 32)):
        lin = torch.nn.Linear(160, 2048).to(dev)
        ref = torch.nn.Linear(160, 2048).to(dev)
        ref.load_state_dict(lin.state_dict())
        xa = torch.randn(first_rows, 160, device=dev)
        xb = torch.randn(second_rows, 160, device=dev)
        ra = torch.randn(first_rows, 2048, device=dev)
        rb = torch.randn(second_rows, 2048, device=dev)

        def run(l):
            ya = HF.mlp(xa, [(l.weight, l.bias, False)])
            yb = HF.mlp(xb, [(l.weight, l.bias, False)])
            ((ya * ra).sum() + (yb * rb).sum()).backward()
            torch.cuda.synchronize()

        run(ref)
        arena = torch.full((lin.weight.numel() + lin.bias.numel(),), float("nan"), device=dev)
        try:
            gradsink.register(lin.weight, arena[:lin.weight.numel()].view_as(lin.weight))
            gradsink.register(lin.bias, arena[lin.weight.numel():].view_as(lin.bias))
            gradsink.begin_step(True)
            run(lin)
            gw, gb = arena[:lin.weight.numel()].view_as(lin.weight), arena[lin.weight.numel():]
            assert lin.weight.grad is None and torch.isfinite(arena).all()
            for got, want in ((gw, ref.weight.grad), (gb, ref.bias.grad)):
                assert (got - want).abs().max().item() <= 2e-2 * want.abs().max().item(), (first_rows, second_rows)
                assert (got - want).norm().item() <= 5e-3 * want.norm().item(), (first_rows, second_rows)
        finally:
            gradsink.clear()
            gradsink.begin_step(False)
