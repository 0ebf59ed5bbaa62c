"""time the small GEMM shapes of a training step (split-K heuristics A/B: HULC_TILE_TARGET / HULC_TILE_MINKT)"""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda'); kn.set_compute("bf16")
shapes = [(512, 128, 2048, 0, 0), (128, 128, 2048, 0, 0), (384, 128, 2048, 0, 0), (2048, 128, 128, 1, 1), (2048, 2048, 64, 0, 0), (2048, 384, 128, 1, 1),
          (128, 3136, 2048, 0, 0), (2048, 3136, 128, 1, 1), (184, 2048, 2048, 0, 0), (2048, 184, 2048, 1, 1), (64, 2048, 2048, 1, 1), (32, 2048, 2048, 1, 1)]
for M, N, K, ak, bk in shapes:
    A = torch.randn((M, K) if ak else (K, M), device=dev); B = torch.randn((N, K) if bk else (K, N), device=dev)
    if bk: B = B.to(torch.bfloat16)
    C = torch.zeros(M, N, device=dev)
    f = lambda: kn.gemm(A, B, C, M, N, K, A.stride(0), B.stride(0), N, a_kmajor=bool(ak), b_kmajor=bool(bk))
    for _ in range(5): f()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): f()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
    print(f"gemm {M:5d} {N:5d} {K:5d} ak{ak} bk{bk}: {e0.elapsed_time(e1) / 20 * 1e3:6.1f} us per call (graph replay of 20)")
