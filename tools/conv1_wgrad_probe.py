"""time conv1's weight gradient (1024 static / gripper frames); HULC_W1_DBG bits switch parts of the kernel off (timing experiments)"""
import sys, torch, os
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
for name, N, H in (("static", 1024, 200), ("gripper", 1024, 84)):
    OH, OW = kn.conv_out_hw(H, H, 8, 8, 4)
    x = torch.randn(N, 3, H, H, device=dev)
    dy = torch.randn(N, OH, OW, 32, device=dev).to(torch.bfloat16)
    dw = torch.empty(32, 192, device=dev); db = torch.empty(32, device=dev)
    f = lambda: kn.conv2d_bwd_weight(x, dy, dw, db, N, H, H, 3, 32, 8, 8, 4, True)
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize()
    print(f"{name:8s} dbg={os.environ.get('HULC_W1_DBG', '0'):3s} old={os.environ.get('HULC_CONV1_WGRAD_OLD', '0')}  {e0.elapsed_time(e1) / 10 * 1e3:.1f} us")
