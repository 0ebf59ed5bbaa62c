"""conv_band_planes (HULC_BAND_PLANES=1) against the shipped band kernels: outputs bit for bit (conv3 forward, conv3 / conv2 data gradients)"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300


def planes(act, Cin):
    pos = (act.float() > 0).reshape(-1, Cin // 32, 32).to(torch.int64)
    w = (pos << torch.arange(32, device=dev)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).t().contiguous().reshape(-1)


def run(which, H, Cin, Cout, K, s, N):
    OH = (H - K) // s + 1
    g = torch.Generator(device=dev).manual_seed(5)
    x = torch.relu(torch.randn(N, H, H, Cin, device=dev, generator=g)).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, K, K, device=dev, generator=g) / (Cin * K * K) ** 0.5
    w2d = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(torch.bfloat16)
    wt = w.permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    b = torch.randn(Cout, device=dev, generator=g)
    y = torch.zeros(N, OH, OH, Cout, device=dev, dtype=torch.bfloat16)
    ybits = torch.zeros(N * OH * OH * (Cout // 32), device=dev, dtype=torch.int32)
    dy = torch.randn(N, OH, OH, Cout, device=dev, generator=g).to(torch.bfloat16)
    dx = torch.zeros(N, H, H, Cin, device=dev, dtype=torch.bfloat16)
    xbits = planes(x, Cin)
    os.environ["HULC_BAND_PLANES"] = which
    kn.conv2d_fwd(x, w2d, b, y, N, H, H, Cin, Cout, K, K, s, False, relu_bits=ybits if K == 4 else None)
    kn.conv2d_bwd_data(dy, wt, dx, x, N, H, H, Cin, Cout, K, K, s, compute=kn.BF16, relu_bits=xbits)
    torch.cuda.synchronize()
    return y, dx


for name, geo in (("conv2", (49, 32, 64, 4, 2)), ("conv3", (23, 64, 64, 3, 1))):
    a = run("0", *geo, N)
    b = run("1", *geo, N)
    print(name, "fwd equal", bool(torch.equal(a[0], b[0])), "dgrad equal", bool(torch.equal(a[1], b[1])),
          "| max diffs", float((a[0].float() - b[0].float()).abs().max()), float((a[1].float() - b[1].float()).abs().max()), flush=True)
