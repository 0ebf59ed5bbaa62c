"""conv1 on uint8 NHWC frames vs fp32 NCHW frames: forward and weight gradient, 1024 static frames, graph-timed.
HULC_W1_DBG bits (1: no MFMA loop, 8: no prefetch loads) split the weight gradient's time, HULC_C1_DBG bits (1: no tile loop, 2: no
staging after the first band) the forward's.  HULC_LIB=<another build of the library> gives an A/B inside one gpurun call."""
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


N, H = int(os.environ.get("N", "1024")), 200
OH, OW = kn.conv_out_hw(H, H, 8, 8, 4)
xf = torch.rand(N, 3, H, H, device=dev) * 2 - 1
xu = torch.randint(0, 256, (N, H, H, 3), device=dev, dtype=torch.uint8)
sh = torch.randint(0, 21, (N, 2), device=dev, dtype=torch.int32)
w = (torch.randn(32, 192, device=dev) / 14).to(torch.bfloat16)
b = torch.zeros(32, device=dev)
y = torch.empty(N, OH, OW, 32, device=dev, dtype=torch.bfloat16)
dy = torch.randn(N, OH, OW, 32, device=dev).to(torch.bfloat16)
dw, db = torch.empty(32, 192, device=dev), torch.empty(32, device=dev)
tag = f"N={N} dbg={os.environ.get('HULC_W1_DBG', '0')}"
print(f"[{tag}] fwd   fp32 {gtime(lambda: kn.conv2d_fwd(xf, w, b, y, N, H, H, 3, 32, 8, 8, 4, True)):7.1f} us   "
      f"uint8 {gtime(lambda: kn.conv2d_fwd(xu, w, b, y, N, H, H, 3, 32, 8, 8, 4, True, aug_shift=sh, aug_pad=10)):7.1f} us")
print(f"[{tag}] wgrad fp32 {gtime(lambda: kn.conv2d_bwd_weight(xf, dy, dw, db, N, H, H, 3, 32, 8, 8, 4, True)):7.1f} us   "
      f"uint8 {gtime(lambda: kn.conv2d_bwd_weight(xu, dy, dw, db, N, H, H, 3, 32, 8, 8, 4, True, aug_shift=sh, aug_pad=10)):7.1f} us")
