"""the small M = 2048 GEMMs of the camera encoders' fc tails (forward and data gradient), one by one"""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/n*1e3
for (M,N,K,relu) in ((2048,512,128,True),(2048,64,512,False),(2048,128,512,False),(2048,512,64,False),(2048,128,3136,True),(2048,3136,128,False)):
    A = torch.randn(M,K,device=dev); W = (torch.randn(N,K,device=dev)*0.05).to(torch.bfloat16); b = torch.zeros(N,device=dev)
    C = torch.empty(M,N,device=dev)
    t = timeit(lambda: kn.gemm(A,W,C,M,N,K,K,K,N,bias=b,relu=relu))
    print(f"gemm M={M} N={N} K={K}: {t:.1f} us  {2*M*N*K/t/1e6:.1f} TFLOP/s  {(A.numel()*4+W.numel()*2+C.numel()*4)/t/1e3:.0f} GB/s")
