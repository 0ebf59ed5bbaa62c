"""time split of the persistent recurrent kernel (forward direction, S = 32, B = 64, H = 2048): HULC_RNN_DBG 0 = as shipped, 1 = no state loads / MFMAs,
2 = no barrier, 3 = neither.   HULC_RNN_PIPE=0|1 python tools/rnn_probe.py"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
S, B, H = 32, 64, 2048
g = torch.Generator().manual_seed(0)
w = [(torch.randn(H, H, generator=g) * 0.02).to(dev).to(torch.bfloat16) for _ in range(3)]
z = torch.zeros(S + 2, B, 2 * H, device=dev)
pre = torch.randn(S, B, H, generator=g).to(dev)
b = [torch.zeros(H, device=dev) for _ in range(3)]
def run():
    return kn.rnn_wavefront(z[0], B * 2 * H, S, B, H, w[0], w[1], w[2], False, add1=pre, add1_step=B * H, ld_add1=H, bias1=(b[0], None), bias2=(b[1], b[2]), relu=True, zero_edges=True)
for dbg in (0, 1, 2, 3):
    os.environ["HULC_RNN_DBG"] = str(dbg)
    for _ in range(3): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): run()
    e1.record(); torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e3
    print(f"PIPE={os.environ.get('HULC_RNN_PIPE', '1')} DBG={dbg}: {t:.1f} us per launch, {t / (S + 1):.2f} us per wave step")
os.environ.pop("HULC_RNN_DBG")
