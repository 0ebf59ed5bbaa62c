import sys, os, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
for M in (32, 64):
    A = torch.randn(M, 32, 2048, device=dev); W = torch.randn(2048, 2048, device=dev).to(torch.bfloat16)
    pre = torch.randn(M, 32, 2048, device=dev); b = torch.randn(2048, device=dev)
    for sk in ("4",):
        os.environ["HULC_SKINNY_SPLITK"] = sk
        def run():
            for t in range(1, 32):
                kn.gemm(A[:, t-1], W, A[:, t], M, 2048, 2048, 32*2048, 2048, 32*2048, bias=b, add=pre[:, t], ld_add=32*2048, relu=True)
        run(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): run()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} splitk={sk}: {e0.elapsed_time(e1)/5/31*1000:.2f} us per recurrent step (graph replay)")
    # time-major activations (rows of h_t contiguous), fp32 and bf16 storage
    for dt in (torch.float32, torch.bfloat16):
        At = torch.randn(32, M, 2048, device=dev).to(dt); pt = torch.randn(32, M, 2048, device=dev)
        os.environ["HULC_SKINNY_SPLITK"] = "4"
        def run2():
            for t in range(1, 32):
                kn.gemm(At[t-1], W, At[t], M, 2048, 2048, 2048, 2048, 2048, bias=b, add=pt[t], ld_add=2048, relu=True)
        run2(); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g): run2()
        g.replay(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): g.replay()
        e1.record(); torch.cuda.synchronize()
        print(f"M={M} splitk=4 time-major {dt}: {e0.elapsed_time(e1)/5/31*1000:.2f} us per step")
