"""camera-encoder head fc2(relu(fc1(x))) at the benchmark's shape (2048 rows, 128 -> 512 -> 64): one launch per direction (csrc/mlp2_rows.hip)
against the two-GEMM path (HULC_NO_MLP2_ROWS=1), per-kernel times from HIP events"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hulc2_amd import functional as HF, kernels as kn

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
T = int(os.environ.get("T", 2048))
fc1, fc2 = torch.nn.Linear(128, 512).to(dev), torch.nn.Linear(512, 64).to(dev)
x = torch.randn(T, 128, device=dev, requires_grad=True)
r = torch.randn(T, 64, device=dev)
for mode in ("fused", "gemm"):
    if mode == "gemm":
        os.environ["HULC_NO_MLP2_ROWS"] = "1"
    def step():
        for q in list(fc1.parameters()) + list(fc2.parameters()):
            q.grad = None
        y = HF.mlp2_rows(x, fc1.weight, fc1.bias, fc2.weight, fc2.bias)
        (y * r).sum().backward()
    for _ in range(5):
        step()
    kn.start_timing()
    for _ in range(20):
        step()
    rec = kn.stop_timing()
    print(mode)
    for k, (n, ms, fl, by) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
        print(f"  {ms / 20 * 1e3:8.1f} us/step  {n // 20:3d} launches  {k}")
