"""GPU parity of the small HBM-bound kernels of the C ABI (column sums = bias gradients) against float64 torch."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


COLSUM = [  # M, N, ld, dtype
    (2048, 2048, 2048, torch.float32),      # RNN pre-activation gradient block
    (2112, 4096, 4096, torch.float32),      # time-major dbuf (33 x 64 rows)
    (2048, 182, 184, torch.float32),        # padded mixture heads (N not a multiple of 4, ld > N)
    (65536, 32, 32, torch.bfloat16),        # conv-like tall and narrow
    (1024, 512, 640, torch.bfloat16),       # strided bf16 view
    (33, 7, 7, torch.float32),              # ragged: scalar path
    (1, 128, 128, torch.float32),           # single row
]


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("M,N,ld,dtype", COLSUM)
def test_colsum(dev, M, N, ld, dtype, accumulate):
    from hulc2_amd import kernels as kn

    g = torch.Generator().manual_seed(M * 31 + N)
    x = torch.randn(M, ld, generator=g).to(dev).to(dtype)
    out0 = torch.randn(N, generator=g).to(dev)
    out = out0.clone()
    kn.colsum(x, M, N, ld, out, accumulate=accumulate)
    torch.cuda.synchronize()
    ref = x[:, :N].double().sum(0) + (out0.double() if accumulate else 0)
    err = (out.double() - ref).abs().max().item()
    assert err < 1e-5 * (M ** 0.5) + 1e-5, f"colsum {M}x{N}: max err {err:.3e}"
    out2 = out0.clone()
    kn.colsum(x, M, N, ld, out2, accumulate=accumulate)
    torch.cuda.synchronize()
    assert torch.equal(out, out2), "colsum must be deterministic (fixed summation order)"
