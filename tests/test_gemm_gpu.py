"""GPU parity of hulc_gemm (C ABI) against a float64 torch reference of the same op.

bf16 compute: operands are rounded to bf16 (as the kernel does while staging) before the reference
product, so the only remaining difference is fp32 accumulation order -> tight tolerance.
fp32 compute: exact-fp32 MFMA, compared against the float64 product.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(A, B, a_kmajor, b_kmajor, compute):
    A2 = A if a_kmajor else A.t()
    B2 = B if b_kmajor else B.t()
    if compute == "bf16":
        A2 = A2.to(torch.bfloat16).to(torch.float64)
        B2 = B2.to(torch.bfloat16).to(torch.float64)
    else:
        A2, B2 = A2.double(), B2.double()
    return A2 @ B2.t()


SHAPES = [
    (32, 2048, 2048),   # RNN step / plan proposal (skinny M)
    (1024, 512, 128),   # vision fc1
    (1024, 64, 512),    # vision fc2
    (1024, 384, 128),   # transformer in_proj
    (2048, 2048, 1024), # big wgrad-like
    (33, 70, 40),       # ragged everything (K multiple of 8)
    (200, 182, 2048),   # decoder heads (N not multiple of 32)
    (64, 2048, 160),
]


@pytest.mark.parametrize("compute", ["bf16", "fp32"])
@pytest.mark.parametrize("a_kmajor,b_kmajor", [(True, True), (True, False), (False, False), (False, True)])
@pytest.mark.parametrize("M,N,K", SHAPES)
def test_gemm_layouts(dev, compute, a_kmajor, b_kmajor, M, N, K):
    from hulc2_amd import kernels as kn

    g = torch.Generator(device="cpu").manual_seed(1234 + M + N + K)
    A = torch.randn((M, K) if a_kmajor else (K, M), generator=g).to(dev)
    B = torch.randn((N, K) if b_kmajor else (K, N), generator=g).to(dev)
    C = torch.full((M, N), float("nan"), device=dev)
    kn.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), C.stride(0), a_kmajor=a_kmajor, b_kmajor=b_kmajor,
            compute=kn._COMPUTE[compute])
    torch.cuda.synchronize()
    ref = _ref(A, B, a_kmajor, b_kmajor, compute)
    err = (C.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    tol = 2e-5 * scale * (K ** 0.5) / 8 + 1e-4
    assert torch.isfinite(C).all(), "non-finite output"
    assert err <= tol, f"max err {err:.3e} > tol {tol:.3e} (scale {scale:.3e})"


@pytest.mark.parametrize("compute", ["bf16", "fp32"])
def test_gemm_asymmetric_identity(dev, compute):
    """A = I with an asymmetric B catches a transposed C-write (MFMA row/col mix-up)."""
    from hulc2_amd import kernels as kn

    M = N = K = 64
    A = torch.eye(M, device=dev)
    B = (torch.arange(N * K, device=dev, dtype=torch.float32).reshape(N, K) % 251) / 8.0  # exact in bf16? no: keep small ints
    B = torch.round(B)
    C = torch.zeros(M, N, device=dev)
    kn.gemm(A, B, C, M, N, K, K, K, N, compute=kn._COMPUTE[compute])
    torch.cuda.synchronize()
    assert torch.equal(C, B.t().contiguous()), f"C != B^T; max diff {(C - B.t()).abs().max().item()}"


@pytest.mark.parametrize("M", [96, 48, 20])      # LDS-tiled path, skinny path with two / one row tiles (split-K epilogue)
@pytest.mark.parametrize("compute", ["bf16", "fp32"])
def test_gemm_epilogue(dev, compute, M):
    from hulc2_amd import kernels as kn

    N, K = 200, 2048
    g = torch.Generator().manual_seed(7)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(N, K, generator=g).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    add = torch.randn(M, N, generator=g).to(dev)
    mask = torch.randn(M, N, generator=g).to(dev)
    C0 = torch.randn(M, N, generator=g).to(dev)
    C = C0.clone()
    kn.gemm(A, B, C, M, N, K, K, K, N, bias=bias, add=add, ld_add=N, mask=mask, ld_mask=N, mask_scale=1.5,
            relu=True, accumulate=True, alpha=0.5, compute=kn._COMPUTE[compute])
    torch.cuda.synchronize()
    ref = 0.5 * _ref(A, B, True, True, compute) + bias.double() + add.double()
    ref = torch.relu(ref)
    ref = torch.where(mask > 0, ref * 1.5, torch.zeros_like(ref)) + C0.double()
    err = (C.double() - ref).abs().max().item()
    assert err < 5e-3, f"epilogue max err {err:.3e}"


def test_gemm_strided_views_and_bf16_storage(dev):
    """Row strides larger than the logical width (time-step slices of (B,S,H)) and bf16 in/out."""
    from hulc2_amd import kernels as kn

    Bsz, S, H = 32, 4, 256
    g = torch.Generator().manual_seed(3)
    h = torch.randn(Bsz, S, H, generator=g).to(dev)
    W = torch.randn(H, H, generator=g).to(dev).to(torch.bfloat16)
    out = torch.zeros(Bsz, S, H, device=dev, dtype=torch.bfloat16)
    for t in range(S):
        kn.gemm(h[:, t], W, out[:, t], Bsz, H, H, S * H, H, S * H, compute=kn.BF16)
    torch.cuda.synchronize()
    ref = h.to(torch.bfloat16).double() @ W.double().t()
    err = (out.double() - ref).abs().max().item()
    assert err < 0.3, f"max err {err}"   # bf16 output rounding of values up to ~60
    rel = ((out.double() - ref).norm() / ref.norm()).item()
    assert rel < 5e-3, f"rel err {rel}"


def test_gemm_dropout_statistics(dev):
    from hulc2_amd import kernels as kn

    M, N, K = 256, 256, 64
    A = torch.ones(M, K, device=dev)
    B = torch.ones(N, K, device=dev) / K
    C = torch.zeros(M, N, device=dev)
    kn.gemm(A, B, C, M, N, K, K, K, N, drop_p=0.1, drop_seed=42, compute=kn.F32)
    C2 = torch.zeros(M, N, device=dev)
    kn.gemm(A, B, C2, M, N, K, K, K, N, drop_p=0.1, drop_seed=42, compute=kn.F32)
    torch.cuda.synchronize()
    assert torch.equal(C, C2), "dropout mask must be a pure function of (seed, index)"
    keep = (C > 0).float().mean().item()
    assert abs(keep - 0.9) < 0.01, f"keep fraction {keep}"
    assert torch.allclose(C[C > 0], torch.tensor(1 / 0.9, device=dev), atol=1e-5)


def test_gemm_rejects_bad_calls(dev):
    from hulc2_amd import kernels as kn
    from hulc2_amd.lib import HulcKernelError

    A = torch.zeros(8, 12, device=dev)
    with pytest.raises(HulcKernelError):
        kn.gemm(A, A, torch.zeros(8, 8, device=dev), 8, 8, 12, 12, 12, 8)   # K % 8 != 0
    with pytest.raises(HulcKernelError):
        kn.gemm(torch.zeros(8, 16), torch.zeros(8, 16), torch.zeros(8, 8), 8, 8, 16, 16, 16, 8)  # CPU tensors


@pytest.mark.parametrize("accumulate", [False, True])
@pytest.mark.parametrize("M,N,K", [(2048, 128, 2048), (128, 3136, 64), (200, 72, 520), (2048, 2048, 256), (68, 40, 33 * 8)])
def test_gemm_fused_rowsum(dev, M, N, K, accumulate):
    """rowsum_a: the bias gradient of a weight-gradient GEMM (row sums of the row-major A operand) from the same launch —
    split-K and single-slice launches, aligned micro-tile staging and ragged edge tiles"""
    from hulc2_amd import kernels as kn

    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g).to(dev)            # [tokens][out features]  (dY)
    B = torch.randn(K, N, generator=g).to(dev)            # [tokens][in features]   (X)
    C = torch.zeros(M, N, device=dev)
    r0 = torch.randn(M, generator=g).to(dev)
    rs = r0.clone()
    kn.gemm(A, B, C, M, N, K, M, N, N, a_kmajor=False, b_kmajor=False, rowsum=rs, rowsum_accumulate=accumulate)
    torch.cuda.synchronize()
    want = A.double().sum(0) + (r0.double() if accumulate else 0)
    err = (rs.double() - want).abs().max().item()
    assert err < 1e-5 * K ** 0.5 + 1e-5, f"rowsum max err {err:.3e}"
    ref = A.to(torch.bfloat16).double().t() @ B.to(torch.bfloat16).double()
    assert (C.double() - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3, "the product itself is unchanged"
    rs2 = r0.clone()
    kn.gemm(A, B, C, M, N, K, M, N, N, a_kmajor=False, b_kmajor=False, rowsum=rs2, rowsum_accumulate=accumulate)
    torch.cuda.synchronize()
    assert torch.equal(rs, rs2), "fixed summation order"


@pytest.mark.parametrize("a16", [False, True])
@pytest.mark.parametrize("M,N,K", [(2048, 128, 2048), (200, 72, 520), (128, 2048, 64)])
def test_gemm_fused_rowsum_kmajor(dev, M, N, K, a16):
    """rowsum_a with a k-major A (the transposed bf16 mirrors of the recurrent decoder's weight gradients)"""
    from hulc2_amd import kernels as kn

    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn(N, K, generator=g).to(dev).to(torch.bfloat16)
    if a16:
        A = A.to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    r0 = torch.randn(M, generator=g).to(dev)
    rs = r0.clone()
    kn.gemm(A, B, C, M, N, K, K, K, N, a_kmajor=True, b_kmajor=True, rowsum=rs, rowsum_accumulate=True)
    torch.cuda.synchronize()
    want = A.double().sum(1) + r0.double()
    assert (rs.double() - want).abs().max().item() < 1e-5 * K ** 0.5 + 1e-5
    ref = A.to(torch.bfloat16).double() @ B.double().t()
    assert (C.double() - ref).abs().max().item() < 2e-3 * ref.abs().max().item() + 1e-3


@pytest.mark.parametrize("M,N,K,a16,b16", [(2048, 2048, 2048, True, True), (256, 128, 520, True, True), (128, 384, 96, True, False),
                                           (256, 128, 64, False, True), (200, 136, 72, True, True)])
def test_gemm_row_major_bf16_operands(dev, M, N, K, a16, b16):
    """dW = dY^T X with bf16 row-major operands (the recurrent decoder reads the bf16 state copies): 8x8 micro-tile staging with
    an in-register 16-bit transpose; ragged shapes take the scalar path.  Exact against float64 on the same bf16 values."""
    from hulc2_amd import kernels as kn

    g = torch.Generator().manual_seed(M * 7 + N + K)
    A = torch.randn(K, M, generator=g).to(dev)
    B = torch.randn(K, N, generator=g).to(dev)
    A = A.to(torch.bfloat16) if a16 else A
    B = B.to(torch.bfloat16) if b16 else B
    C = torch.zeros(M, N, device=dev)
    rs = torch.zeros(M, device=dev)
    kn.gemm(A, B, C, M, N, K, M, N, N, a_kmajor=False, b_kmajor=False, rowsum=rs if M > 64 else None)
    torch.cuda.synchronize()
    ref = A.to(torch.bfloat16).double().t() @ B.to(torch.bfloat16).double()
    err = (C.double() - ref).abs().max().item()
    assert err < 1e-5 * K ** 0.5 * ref.abs().max().item() / max(K ** 0.5, 1) + 2e-3, f"max err {err:.3e}"
    if M > 64:
        want = A.double().sum(0)
        assert (rs.double() - want).abs().max().item() < 1e-4 * K ** 0.5 + 1e-4


@pytest.mark.parametrize("M,N,K,accumulate", [(1024, 512, 640, False), (512, 2048, 1024, True)])
def test_gemm_tn128_strided_views_and_accumulate(dev, M, N, K, accumulate):
    """the 128-deep k-step kernel (csrc/gemm_tn128.hip) on what the recurrent decoder hands it: column slices of wider bf16 row-major
    buffers (lda / ldb = twice the width), output overwritten or accumulated, fused bias gradient; an asymmetric product so that a
    transposed result cannot pass; identical to the generic kernel (HULC_NO_GEMM_TN128) up to fp32 summation order"""
    import os
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(M + 3 * N + K)
    Aw = torch.randn(K, 2 * M, generator=g).to(dev).to(torch.bfloat16)
    Bw = torch.randn(K, 2 * N, generator=g).to(dev).to(torch.bfloat16)
    A, B = Aw[:, M:], Bw[:, :N]
    C0 = torch.randn(M, N, generator=g).to(dev)
    rs0 = torch.randn(M, generator=g).to(dev)

    def run():
        C, rs = C0.clone(), rs0.clone()
        kn.gemm(A, B, C, M, N, K, 2 * M, 2 * N, N, a_kmajor=False, b_kmajor=False, accumulate=accumulate, rowsum=rs, rowsum_accumulate=accumulate)
        torch.cuda.synchronize()
        return C, rs
    C, rs = run()
    ref = A.double().t() @ B.double() + (C0.double() if accumulate else 0)
    assert (C.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item() + 1e-4
    want = A.double().sum(0) + (rs0.double() if accumulate else 0)
    assert (rs.double() - want).abs().max().item() < 1e-4 * K ** 0.5 + 1e-4
    os.environ["HULC_NO_GEMM_TN128"] = "1"
    try:
        C2, rs2 = run()
    finally:
        del os.environ["HULC_NO_GEMM_TN128"]
    assert (C - C2).abs().max().item() < 1e-4 * ref.abs().max().item()


@pytest.mark.parametrize("M,N,K,accumulate", [(512, 640, 1024, False), (2048, 2048, 2048, True), (768, 512, 576, False)])
def test_gemm_nt128_strided_views_and_accumulate(dev, M, N, K, accumulate):
    """the k-major x k-major 128 x 128 kernel (csrc/gemm_nt128.hip) on what the recurrent decoder's transposed mirrors look like: row slices
    of wider bf16 buffers (lda / ldb beyond K, a column offset), output overwritten or accumulated, fused bias gradient (row sums of A); an
    asymmetric product; identical to the generic kernel (HULC_NO_GEMM_NT128) up to fp32 summation order"""
    import os
    from hulc2_amd import kernels as kn

    kn.set_compute("bf16")
    g = torch.Generator().manual_seed(M + 3 * N + K)
    Aw = torch.randn(M, K + 64, generator=g).to(dev).to(torch.bfloat16)
    Bw = torch.randn(N, K + 128, generator=g).to(dev).to(torch.bfloat16)
    A, B = Aw[:, 64:], Bw[:, :K]
    C0 = torch.randn(M, N, generator=g).to(dev)
    rs0 = torch.randn(M, generator=g).to(dev)

    def run():
        C, rs = C0.clone(), rs0.clone()
        kn.gemm(A, B, C, M, N, K, K + 64, K + 128, N, a_kmajor=True, b_kmajor=True, accumulate=accumulate, rowsum=rs, rowsum_accumulate=accumulate)
        torch.cuda.synchronize()
        return C, rs
    C, rs = run()
    ref = A.double() @ B.double().t() + (C0.double() if accumulate else 0)
    assert (C.double() - ref).abs().max().item() < 1e-5 * ref.abs().max().item() + 1e-4
    want = A.double().sum(1) + (rs0.double() if accumulate else 0)
    assert (rs.double() - want).abs().max().item() < 1e-4 * K ** 0.5 + 1e-4
    os.environ["HULC_NO_GEMM_NT128"] = "1"
    try:
        C2, rs2 = run()
    finally:
        del os.environ["HULC_NO_GEMM_NT128"]
    assert (C - C2).abs().max().item() < 1e-4 * ref.abs().max().item()
