"""per-wave phase split of a unit in conv_band_planes (HULC_BAND_PLANES=1 + HULC_BAND_STAMPS): conv3 forward / data gradient, conv2 data gradient"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
N = 2048
os.environ["HULC_BAND_PLANES"] = "1"
st = torch.zeros(256 * 8 * 5, dtype=torch.int64, device=dev)


def planes(act, Cin):
    pos = (act.float() > 0).reshape(-1, Cin // 32, 32).to(torch.int64)
    w = (pos << torch.arange(32, device=dev)).sum(-1)
    return torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).t().contiguous().reshape(-1)


def report(name):
    torch.cuda.synchronize()
    t = st.view(256, 8, 5).double()
    units = t[:, :, 4].clamp(min=1)
    per = t[:, :, :4] / units.unsqueeze(-1)
    m = per.mean((0, 1))
    print(f"{name}: cycles per unit and wave  issue {m[0]:7.0f} | tiles {m[1]:7.0f} | wait(vmcnt) {m[2]:7.0f} | barrier {m[3]:7.0f} | sum {m.sum():7.0f}  (units per workgroup {units.mean():.1f})")
    for w in range(8):
        mw = per[:, w].mean(0)
        print(f"   wave {w}: issue {mw[0]:6.0f} tiles {mw[1]:6.0f} wait {mw[2]:6.0f} barrier {mw[3]:6.0f}")


for (H, Cin, Cout, K, s) in ((23, 64, 64, 3, 1), (49, 32, 64, 4, 2)):
    OH = (H - K) // s + 1
    x = torch.relu(torch.randn(N, H, H, Cin, device=dev)).to(torch.bfloat16)
    w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
    w2d = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(torch.bfloat16)
    wt = w.permute(1, 2, 3, 0).contiguous().to(torch.bfloat16)
    b = torch.zeros(Cout, device=dev)
    y = torch.empty(N, OH, OH, Cout, device=dev, dtype=torch.bfloat16)
    dy = torch.randn(N, OH, OH, Cout, device=dev).to(torch.bfloat16)
    dx = torch.empty(N, H, H, Cin, device=dev, dtype=torch.bfloat16)
    xbits = planes(x, Cin)
    os.environ["HULC_BAND_STAMPS"] = hex(st.data_ptr())
    if K == 3:
        for _ in range(2):
            st.zero_(); kn.conv2d_fwd(x, w2d, b, y, N, H, H, Cin, Cout, K, K, s, False)
        report("conv3 forward")
    for _ in range(2):
        st.zero_(); kn.conv2d_bwd_data(dy, wt, dx, x, N, H, H, Cin, Cout, K, K, s, compute=kn.BF16, relu_bits=xbits)
    report(f"conv{2 if K == 4 else 3} data gradient")
    os.environ["HULC_BAND_STAMPS"] = ""
