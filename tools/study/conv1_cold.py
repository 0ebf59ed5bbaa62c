"""conv1's weight gradient alone in a loop vs behind another big kernel (its state when it runs inside the step): graph-timed, 2048 frames"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")


def gtime(fn, rep=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(rep):
                fn()
    torch.cuda.synchronize(); g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); g.replay(); g.replay(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * rep) * 1e3


N, H = 2048, 200
OH, OW = kn.conv_out_hw(H, H, 8, 8, 4)
xa = torch.rand(N // 2, 3, H, H, device=dev) * 2 - 1
xb = torch.rand(N // 2, 3, H, H, device=dev) * 2 - 1
xf = torch.rand(N, 3, H, H, device=dev) * 2 - 1
dy = torch.randn(N, OH, OW, 32, device=dev).to(torch.bfloat16)
dw, db = torch.empty(32, 192, device=dev), torch.empty(32, device=dev)
big = torch.empty(256 * 1024 * 1024, device=dev)            # 1 GiB fill between the launches: L2 / MALL / TLB state of a kernel inside the step
other = torch.empty(400 * 1024 * 1024 // 2, device=dev, dtype=torch.bfloat16)
wg = lambda: kn.conv2d_bwd_weight(xf, dy, dw, db, N, H, H, 3, 32, 8, 8, 4, True)
t_wg = gtime(wg)
t_fill = gtime(lambda: big.zero_())
t_both = gtime(lambda: (big.zero_(), wg()))
print(f"weight gradient back to back {t_wg:.1f} us | 1 GiB fill {t_fill:.1f} us | fill + weight gradient {t_both:.1f} us -> weight gradient behind the fill {t_both - t_fill:.1f} us")
# dY written just before (as conv2's data gradient does in the step)
t_w = gtime(lambda: dy.copy_(other[: dy.numel()].view_as(dy)))
t_both2 = gtime(lambda: (big.zero_(), dy.copy_(other[: dy.numel()].view_as(dy)), wg()))
print(f"fill + dY written + weight gradient {t_both2:.1f} us -> weight gradient {t_both2 - t_fill - t_w:.1f} us (dY copy alone {t_w:.1f})")
