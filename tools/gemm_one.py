import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
M, N, K, ak, bk = (int(v) for v in sys.argv[1:6])
a16, b16 = (int(v) for v in sys.argv[6:8]) if len(sys.argv) > 7 else (0, bk)
A = torch.randn((M, K) if ak else (K, M), device=dev)
B = torch.randn((N, K) if bk else (K, N), device=dev)
if a16: A = A.to(torch.bfloat16)
if b16: B = B.to(torch.bfloat16)
C = torch.zeros(M, N, device=dev)
for _ in range(20):
    kn.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), N, a_kmajor=bool(ak), b_kmajor=bool(bk), accumulate=not ak)
torch.cuda.synchronize()
