"""per-wave phase split of a unit (one 4-row band of one frame) in conv1's weight gradient: HULC_W1_STAMPS instance of conv1_wgrad_kernel"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
N, H = int(os.environ.get("N", "2048")), 200
OH, OW = kn.conv_out_hw(H, H, 8, 8, 4)
x = torch.rand(N, 3, H, H, device=dev) * 2 - 1
dy = torch.randn(N, OH, OW, 32, device=dev).to(torch.bfloat16)
dw, db = torch.empty(32, 192, device=dev), torch.empty(32, device=dev)
st = torch.zeros(512 * 8 * 7, dtype=torch.int64, device=dev)
f = lambda: kn.conv2d_bwd_weight(x, dy, dw, db, N, H, H, 3, 32, 8, 8, 4, True)
for _ in range(2):
    f()
os.environ["HULC_W1_STAMPS"] = hex(st.data_ptr())
for _ in range(2):
    st.zero_(); f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record(); torch.cuda.synchronize()
t = st.view(512, 8, 7).double()
t = t[t[:, 0, 6] > 0]                                    # the workgroups that ran (HULC_CONV1_SLOTS may launch fewer than 512)
units = t[:, :, 6].clamp(min=1)
per = t[:, :, :6] / units.unsqueeze(-1)
names = ("issue", "mfma", "bar1", "wait", "store", "bar2")
m = per.mean((0, 1))
print(f"conv1 weight gradient, stamped instance {e0.elapsed_time(e1) * 1e3:.0f} us for {N} frames; cycles per unit and wave: "
      + " | ".join(f"{n} {v:7.1f}" for n, v in zip(names, m)) + f" | sum {m.sum():8.1f}  (units per workgroup {units.mean():.1f})")
for w in range(8):
    mw = per[:, w].mean(0)
    print(f"   wave {w}: " + " ".join(f"{n} {v:7.1f}" for n, v in zip(names, mw)))
