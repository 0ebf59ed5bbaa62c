"""What a collective's resident workgroups cost the camera encoders' conv backward: the conv backward chain of the step (raw kernel calls, replayed
graph) alone, and beside `n` dummy workgroups that hold CU slots for ~1.2 ms on a second stream (occupy_probe.hip: no memory traffic — the effect is
slot occupancy alone).  The persistent conv kernels launch exactly as many workgroups as the chip has slots and split their frames statically."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from _build import ensure
from hulc2_amd import kernels as kn

so = ctypes.CDLL(ensure("occupy_probe.so"))
so.occupy_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_ulonglong, ctypes.c_void_p, ctypes.c_void_p]
dev = torch.device('cuda')
kn.set_compute("bf16")
N = 2048
bf = torch.bfloat16
x = torch.randn(N, 3, 200, 200, device=dev)
w2t = (torch.randn(32, 4, 4, 64, device=dev) / 512 ** 0.5).to(bf)
w3t = (torch.randn(64, 3, 3, 64, device=dev) / 576 ** 0.5).to(bf)
y1 = torch.randn(N, 49, 49, 32, device=dev).to(bf)
y2 = torch.randn(N, 23, 23, 64, device=dev).to(bf)
g3 = torch.randn(N, 21, 21, 64, device=dev).to(bf)
g2 = torch.empty_like(y2); g1 = torch.empty_like(y1)
dw1 = torch.empty(32, 192, device=dev); db1 = torch.empty(32, device=dev)
dw2 = torch.empty(64, 512, device=dev); db2 = torch.empty(64, device=dev)
dw3 = torch.empty(64, 576, device=dev); db3 = torch.empty(64, device=dev)
sink = torch.zeros(4, device=dev)


def bwd():
    kn.conv2d_bwd_weight(y2, g3, dw3, db3, N, 23, 23, 64, 64, 3, 3, 1, False, dw_oihw=True)
    kn.conv2d_bwd_data(g3, w3t, g2, y2, N, 23, 23, 64, 64, 3, 3, 1)
    kn.conv2d_bwd_weight(y1, g2, dw2, db2, N, 49, 49, 32, 64, 4, 4, 2, False, dw_oihw=True)
    kn.conv2d_bwd_data(g2, w2t, g1, y1, N, 49, 49, 32, 64, 4, 4, 2)
    kn.conv2d_bwd_weight(x, g1, dw1, db1, N, 200, 200, 3, 32, 8, 8, 4, True, dw_oihw=True)


for _ in range(2): bwd()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    with torch.cuda.graph(g, stream=side):
        bwd()
torch.cuda.synchronize()
occ = torch.cuda.Stream()
cur = torch.cuda.current_stream()


def run(n, regs, lds, ticks=120000, rep=5):
    ts = []
    for _ in range(rep):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        if n:
            occ.wait_stream(cur)
            rc = so.occupy_launch(n, regs, lds, ticks, sink.data_ptr(), occ.cuda_stream)
            assert rc == 0, rc
        e0.record(cur)
        g.replay()
        e1.record(cur)
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[len(ts) // 2]


so.stream_launch.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p]
NP = 47_000_000 // 4
arr = [torch.zeros(NP * 4, device=dev) for _ in range(4)]


def run_stream(n, pace, rep=5):
    """conv chain beside a paced streaming pass over 4 x 188 MB: -> (chain us, pass us)"""
    tc, tp = [], []
    for _ in range(rep):
        torch.cuda.synchronize()
        e0, e1, s0, s1 = (torch.cuda.Event(enable_timing=True) for _ in range(4))
        occ.wait_stream(cur)
        s0.record(occ)
        rc = so.stream_launch(n, arr[0].data_ptr(), arr[1].data_ptr(), arr[2].data_ptr(), arr[3].data_ptr(), NP, pace, occ.cuda_stream)
        assert rc == 0, rc
        s1.record(occ)
        e0.record(cur)
        g.replay()
        e1.record(cur)
        torch.cuda.synchronize()
        tc.append(e0.elapsed_time(e1) * 1e3); tp.append(s0.elapsed_time(s1) * 1e3)
    return sorted(tc)[len(tc) // 2], sorted(tp)[len(tp) // 2]


print(f"conv backward chain alone: {run(0, 0, 0):7.1f} us")
for n, regs, lds in ((256, 32, 0), (512, 32, 0), (1024, 32, 0)):
    print(f"beside {n:4d} resident workgroups ({regs} registers, {lds >> 10} KB LDS): {run(n, regs, lds):7.1f} us")
for n, pace in ((256, 0), (256, 1), (256, 2), (256, 4), (128, 0), (128, 1), (128, 2), (64, 0), (512, 0), (512, 2), (512, 4)):
    c, p_ = run_stream(n, pace)
    print(f"beside a streaming pass of {n:4d} workgroups, pace {pace}: chain {c:7.1f} us, pass {p_:7.1f} us ({NP * 16 * 7 / p_ / 1e6:.2f} TB/s)")
sys.exit(0)
for n, regs, lds in ((8, 32, 0), (16, 32, 0), (32, 32, 0), (64, 32, 0), (16, 120, 0), (32, 120, 0), (16, 120, 32768), (32, 120, 32768), (64, 120, 32768)):
    print(f"beside {n:3d} resident workgroups ({regs} registers, {lds >> 10} KB LDS): {run(n, regs, lds):7.1f} us")
