"""Times the six NHWC conv kernels of a camera encoder (conv2 / conv3: forward with sign planes, data gradient from sign planes, weight
gradient) at several frame counts, each launch between a cold-cache fill when COLD=1.  HULC_BAND_DBG / HULC_NO_BAND etc. are read by the
library once per process: run one process per variant.  usage: python tools/conv_probe.py [N ...]"""
import os
import sys

import torch

sys.path.insert(0, '.')
from hulc2_amd import kernels as kn

dev = torch.device('cuda')
Ns = [int(a) for a in sys.argv[1:]] or [1024, 2048]
CAM = os.environ.get("CAM", "static")
REP = int(os.environ.get("REP", "10"))
geo = {"static": ((49, 32, 64, 4, 2), (23, 64, 64, 3, 1)), "grip": ((20, 32, 64, 4, 2), (9, 64, 64, 3, 1))}[CAM]


def timeit(fn):
    """REP launches captured into one hipGraph and replayed: GPU-side back-to-back time (a Python launch loop is CPU-bound below ~12 us)"""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(REP):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * REP) * 1e3


def planes(act, Cin):
    pos = (act.float() > 0).reshape(-1, Cin // 32, 32).to(torch.int64)
    w = (pos << torch.arange(32, device=dev)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).t().contiguous().reshape(-1)


tag = " ".join(f"{k}={v}" for k, v in os.environ.items() if k.startswith("HULC_"))
for N in Ns:
    row = []
    for (H, Cin, Cout, K, s) in geo:
        OH = (H - K) // s + 1
        x = torch.relu(torch.randn(N, H, H, Cin, device=dev)).to(torch.bfloat16)
        w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
        w2d = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(torch.bfloat16)
        wt = w.permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
        b = torch.zeros(Cout, device=dev)
        y = torch.empty(N, OH, OH, Cout, device=dev, dtype=torch.bfloat16)
        ybits = torch.empty(N * OH * OH * (Cout // 32), device=dev, dtype=torch.int32)
        dy = torch.randn(N, OH, OH, Cout, device=dev).to(torch.bfloat16)
        dx = torch.empty(N, H, H, Cin, device=dev, dtype=torch.bfloat16)
        xbits = planes(x, Cin)
        dw = torch.empty(Cout, Cin * K * K, device=dev)
        db = torch.empty(Cout, device=dev)
        flops = 2.0 * N * OH * OH * Cout * Cin * K * K
        fwd_bits = ybits if K == 4 else None          # conv2's forward writes planes (conv3's output feeds the spatial softmax)
        tf = timeit(lambda: kn.conv2d_fwd(x, w2d, b, y, N, H, H, Cin, Cout, K, K, s, False, relu_bits=fwd_bits))
        td = timeit(lambda: kn.conv2d_bwd_data(dy, wt, dx, x, N, H, H, Cin, Cout, K, K, s, compute=kn.BF16, relu_bits=xbits))
        tw = timeit(lambda: kn.conv2d_bwd_weight(x, dy, dw, db, N, H, H, Cin, Cout, K, K, s, False))
        row.append(f"conv{2 if K == 4 else 3}: fwd {tf:6.1f} ({flops / tf / 1e6:4.0f} TF) dgrad {td:6.1f} ({flops / td / 1e6:4.0f}) wgrad {tw:6.1f} ({flops / tw / 1e6:4.0f})")
    print(f"[{CAM} N={N:5d} {tag}] " + " | ".join(row), flush=True)
