"""the recurrent decoder's weight-gradient GEMM (2048^3, both operands token-major bf16) through hulc_gemm, graph-timed; HULC_LIB for an A/B"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
M = N = K = int(os.environ.get("MNK", "2048"))
a = torch.randn(K, M, device=dev).to(torch.bfloat16)
b = torch.randn(K, N, device=dev).to(torch.bfloat16)
c = torch.zeros(M, N, device=dev)
rs = torch.zeros(M, device=dev) if not os.environ.get('NO_ROWSUM') else None


def run():
    kn.gemm(a, b, c, M, N, K, M, N, N, a_kmajor=False, b_kmajor=False, compute=kn.BF16, rowsum=rs)


run(); torch.cuda.synchronize(); run(); torch.cuda.synchronize(); run(); torch.cuda.synchronize()
ref = a.float().t() @ b.float()
print("max rel err", float((c - ref).abs().max() / ref.abs().max()), "rowsum err", float((rs - a.float().sum(0)).abs().max()) if rs is not None else None)
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    run(); torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=s):
        for _ in range(10):
            run()
torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
print(f"[{os.environ.get('HULC_LIB', 'lib')[-28:]}] tn gemm {M}^3 rowsum={rs is not None} deep={os.environ.get('HULC_TN128_DEEP', '1')}: {us:.1f} us = {2.0 * M * N * K / us / 1e6:.0f} TFLOP/s")
