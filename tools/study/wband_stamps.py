"""per-wave phase split of a unit in the conv2 / conv3 weight-gradient band kernel (HULC_WB_STAMPS instance of conv_wgrad_band_kernel)"""
import os, sys
import torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
kn.set_compute("bf16")
N = int(os.environ.get("N", "2048"))
st = torch.zeros(512 * 8 * 7, dtype=torch.int64, device=dev)
names = ("issue", "mfma", "bar1", "wait", "store", "bar2")
for (H, Cin, Cout, K, s) in ((49, 32, 64, 4, 2), (23, 64, 64, 3, 1)):
    OH = (H - K) // s + 1
    x = torch.relu(torch.randn(N, H, H, Cin, device=dev)).to(torch.bfloat16)
    dy = torch.randn(N, OH, OH, Cout, device=dev).to(torch.bfloat16)
    dw, db = torch.empty(Cout, Cin * K * K, device=dev), torch.empty(Cout, device=dev)
    f = lambda: kn.conv2d_bwd_weight(x, dy, dw, db, N, H, H, Cin, Cout, K, K, s, False)
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        f()
    e1.record(); torch.cuda.synchronize()
    plain = e0.elapsed_time(e1) * 100
    os.environ["HULC_WB_STAMPS"] = hex(st.data_ptr())
    for _ in range(2):
        st.zero_(); f()
    torch.cuda.synchronize()
    os.environ["HULC_WB_STAMPS"] = ""
    t = st.view(512, 8, 7).double()
    t = t[t[:, 0, 6] > 0]
    units = t[:, :, 6].clamp(min=1)
    per = t[:, :, :6] / units.unsqueeze(-1)
    m = per.mean((0, 1))
    print(f"conv{2 if K == 4 else 3} weight gradient ({plain:.0f} us with its reduce, {N} frames, {t.shape[0]} workgroups, {units.mean():.1f} units each); cycles per unit and wave: "
          + " | ".join(f"{n} {v:7.0f}" for n, v in zip(names, m)) + f" | sum {m.sum():8.0f}")
    for w in range(8):
        mw = per[:, w].mean(0)
        print(f"   wave {w}: " + " ".join(f"{n} {v:7.0f}" for n, v in zip(names, mw)))
