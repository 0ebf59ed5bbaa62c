import os, sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
S, H = 32, 2048
g = torch.Generator().manual_seed(0)
w = [((torch.rand(H, H, generator=g) * 2 - 1) * H ** -0.5).to(dev).to(torch.bfloat16) for _ in range(3)]
b = [torch.zeros(H, device=dev) for _ in range(3)]
for B in (64, 48, 32, 16, 8):
    pre0 = torch.randn(S, B, H, generator=g).to(dev)
    for tr in (False, True):
        zbuf = torch.zeros(S + 2, B, 2 * H, device=dev)
        f = lambda: kn.rnn_wavefront(zbuf[0], B * 2 * H, S, B, H, w[0], w[1], w[2], tr, add1=pre0, add1_step=B * H, ld_add1=H, bias1=(b[0], None), bias2=(b[1], b[2]), relu=True)
        for _ in range(2): f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): f()
        e1.record(); torch.cuda.synchronize()
        print(f"rnn_wavefront B={B} transposed={tr}: {e0.elapsed_time(e1) / 5 * 1e3:.1f} us per pass, {e0.elapsed_time(e1) / 5 / (S + 1) * 1e3:.2f} us per wave step")
