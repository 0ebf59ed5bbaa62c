import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device("cuda", 0)
kn.set_compute("bf16")
torch.manual_seed(0)
for M, N, K in ((2, 128, 32), (32, 128, 32), (64, 128, 32), (32, 128, 64), (32, 2048, 2048), (32, 4096, 128)):
    A = torch.randn(M, K, device=dev); B = torch.randn(N, K, device=dev)
    mask = torch.randn(M, N, device=dev)
    for use_mask in (False, True):
        C = torch.empty(M, N, device=dev)
        kn.gemm(A, B.bfloat16(), C, M, N, K, K, K, N, mask=mask if use_mask else None, ld_mask=N)
        ref32 = A.double() @ B.double().t()
        refbf = A.bfloat16().double() @ B.bfloat16().double().t()
        if use_mask:
            ref32 = ref32 * (mask > 0); refbf = refbf * (mask > 0)
        rel = lambda a, b: ((a.double() - b).norm() / b.norm()).item()
        print(f"M={M} N={N} K={K} mask={use_mask}: vs fp64 {rel(C, ref32):.2e}   vs bf16-rounded-input fp64 {rel(C, refbf):.2e}")
