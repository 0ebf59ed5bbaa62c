import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
shapes = [(128, 128, 2048, 0, 0), (384, 128, 2048, 0, 0), (512, 128, 2048, 0, 0), (2048, 128, 2048, 0, 0), (128, 3136, 2048, 0, 0), (64, 512, 2048, 0, 0),
          (2048, 128, 128, 1, 1), (2048, 2048, 128, 1, 1), (2048, 128, 2048, 1, 1), (2048, 3136, 128, 1, 1), (2048, 384, 128, 1, 1), (2048, 512, 128, 1, 1), (2048, 64, 512, 1, 1),
          (2048, 2048, 2048, 0, 0), (2048, 2048, 32, 0, 0), (128, 3136, 1024, 0, 0), (128, 2048, 1024, 0, 0), (512, 128, 1024, 0, 0), (2048, 128, 1024, 0, 0),
          (1024, 128, 2048, 1, 0), (1024, 128, 384, 1, 0), (1024, 3136, 128, 1, 0), (1024, 2048, 128, 1, 1), (1024, 128, 2048, 1, 1)]
for M, N, K, ak, bk in shapes:
    A = torch.randn((M, K) if ak else (K, M), device=dev)
    B = torch.randn((N, K) if bk else (K, N), device=dev)
    if bk or ak != bk:  # weights are bf16 shadows in the model for the (1,x) shapes
        B = B.to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    lda, ldb = A.stride(0), B.stride(0)
    f = lambda: kn.gemm(A, B, C, M, N, K, lda, ldb, N, a_kmajor=bool(ak), b_kmajor=bool(bk), accumulate=not ak)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 10
    print(f"{(M, N, K, ak, bk)}: {t * 1e3:8.1f} us  {2 * M * N * K / t / 1e9:8.1f} TFLOP/s")
