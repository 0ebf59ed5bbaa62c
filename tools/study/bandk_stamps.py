"""per-wave phase split of a unit in conv_band_kernel's conv2-forward instance (HULC_BANDK_STAMPS): issue of the next band's loads | tile loop |
barrier | LDS stores + barrier"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
N = int(os.environ.get("N", "2048"))
H, Cin, Cout, K, s = 49, 32, 64, 4, 2
OH = (H - K) // s + 1
x = torch.relu(torch.randn(N, H, H, Cin, device=dev)).to(torch.bfloat16)
w = torch.randn(Cout, Cin, K, K, device=dev) / (Cin * K * K) ** 0.5
w2d = w.permute(0, 2, 3, 1).reshape(Cout, -1).contiguous().to(torch.bfloat16)
b = torch.zeros(Cout, device=dev)
y = torch.empty(N, OH, OH, Cout, device=dev, dtype=torch.bfloat16)
ybits = torch.empty(N * OH * OH * (Cout // 32), device=dev, dtype=torch.int32)
f = lambda: kn.conv2d_fwd(x, w2d, b, y, N, H, H, Cin, Cout, K, K, s, False, relu_bits=ybits)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    f()
e1.record(); torch.cuda.synchronize()
st = torch.zeros(512 * 8 * 5, dtype=torch.int64, device=dev)
os.environ["HULC_BANDK_STAMPS"] = hex(st.data_ptr())
for _ in range(2):
    st.zero_(); f()
torch.cuda.synchronize()
os.environ["HULC_BANDK_STAMPS"] = ""
t = st.view(512, 8, 5).double(); t = t[t[:, 0, 4] > 0]
units = t[:, :, 4].clamp(min=1); per = t[:, :, :4] / units.unsqueeze(-1)
names = ("issue", "tiles", "barrier", "store+barrier")
m = per.mean((0, 1))
print(f"conv2 forward {e0.elapsed_time(e1) * 100:.0f} us per {N} frames, {t.shape[0]} workgroups x {units.mean():.1f} units; cycles per unit and wave: "
      + " | ".join(f"{n} {v:7.0f}" for n, v in zip(names, m)) + f" | sum {m.sum():7.0f}")
for wv in range(8):
    mw = per[:, wv].mean(0)
    print(f"   wave {wv}: " + " ".join(f"{n} {v:7.0f}" for n, v in zip(names, mw)))
