"""time the fused attention-half kernels alone (hipGraph replay of 20 launches, HIP events)"""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device("cuda", 0)
B, S, E, H, FF = 64, 32, 128, 8, 2048
T = B * S
g = torch.Generator().manual_seed(0)
r = lambda *s: (torch.randn(*s, generator=g) * 0.1).to(dev)
x, Wqkv, bqkv, Wo, bo, gamma, beta = r(T, E), r(3 * E, E), r(3 * E), r(E, E), r(E), r(E) + 1, r(E)
W16, Wo16 = Wqkv.bfloat16(), Wo.bfloat16()
WT16, WoT16 = Wqkv.t().contiguous().bfloat16(), Wo.t().contiguous().bfloat16()
y, pre, mean, rstd = torch.empty(T, E, device=dev), torch.empty(T, E, device=dev), torch.empty(T, device=dev), torch.empty(T, device=dev)
ctx = torch.empty(T, E, dtype=torch.bfloat16, device=dev)
dy, slabs = r(T, E), r(FF // 128, T, E)
dx, d_o, dqkv, lnp = torch.empty(T, E, device=dev), torch.empty(T, E, dtype=torch.bfloat16, device=dev), torch.empty(T, 3 * E, dtype=torch.bfloat16, device=dev), torch.empty(B, 2, E, device=dev)

def timeit(fn, name):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    with torch.cuda.graph(gr, stream=side):
        for _ in range(20): fn()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): gr.replay()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:40s} {e0.elapsed_time(e1) / 100 * 1e3:8.2f} us per launch")

for p in (0.0, 0.1):
    timeit(lambda: kn.txl_attn_fwd(x, W16, bqkv, Wo16, bo, gamma, beta, 1e-5, B, S, H, p, 11, 12, y, pre, mean, rstd, ctx), f"txl_attn_fwd drop={p}")
    timeit(lambda: kn.txl_attn_bwd(x, W16, WT16, WoT16, bqkv, gamma, 1e-5, B, S, H, p, 11, 12, pre, mean, rstd, dy, slabs, FF // 128, T * E, dx, d_o, dqkv, lnp),
           f"txl_attn_bwd drop={p} 16 slabs")
    timeit(lambda: kn.txl_attn_bwd(x, W16, WT16, WoT16, bqkv, gamma, 1e-5, B, S, H, p, 11, 12, pre, mean, rstd, dy, None, 0, 0, dx, d_o, dqkv, lnp),
           f"txl_attn_bwd drop={p} no slabs")
timeit(lambda: kn.advance_step_state(dev), "1-thread kernel (launch floor)")
