import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
def run(M,N,K,ak,bk,a16,b16,seed=0):
    A = torch.randn((M, K) if ak else (K, M), device=dev)
    B = torch.randn((N, K) if bk else (K, N), device=dev)
    if a16: A = A.to(torch.bfloat16)
    if b16: B = B.to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    f=lambda: kn.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), N, a_kmajor=bool(ak), b_kmajor=bool(bk), accumulate=True, drop_seed=seed)
    for _ in range(5): f()
    torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): f()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/20*1e3
    print(f"M{M} N{N} K{K} ak{ak} bk{bk} a16={a16} b16={b16}: {t:.1f} us  {2*M*N*K/t/1e6:.0f} TF/s")
run(2048,2048,2048,0,0,1,1)
run(2048,2048,2048,1,1,1,1)
run(2048,2048,2048,1,1,0,1)
run(2048,2048,2048,0,0,0,0)
