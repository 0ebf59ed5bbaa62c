"""goal-encoder MLPs (128 / 384 -> 2048 -> 2048 -> 32 on 32 rows): two hulc_mlp_chain launches against one hulc_mlp_chain2 launch, per direction"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hulc2_amd import functional as HF, kernels as kn

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
da, db_ = (128, 2048, 2048, 32), (384, 2048, 2048, 32)
la = [torch.nn.Linear(a, b).to(dev) for a, b in zip(da[:-1], da[1:])]
lb = [torch.nn.Linear(a, b).to(dev) for a, b in zip(db_[:-1], db_[1:])]
xa = torch.randn(32, 128, device=dev, requires_grad=True)
xb = torch.randn(32, 384, device=dev)
ra, rb = torch.randn(32, 32, device=dev), torch.randn(32, 32, device=dev)
layers = lambda ls: [(l.weight, l.bias, i < len(ls) - 1) for i, l in enumerate(ls)]
for mode in ("paired", "single"):
    def step():
        for l in la + lb:
            l.weight.grad = l.bias.grad = None
        if mode == "paired":
            ya, yb = HF.dual_mlp(xa, layers(la), xb, layers(lb))
        else:
            ya, yb = HF.mlp(xa, layers(la)), HF.mlp(xb, layers(lb))
        ((ya * ra).sum() + (yb * rb).sum()).backward()
    for _ in range(5):
        step()
    kn.start_timing()
    for _ in range(20):
        step()
    rec = kn.stop_timing()
    print(mode)
    for k, (n, ms, fl, by) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
        if "chain" in str(k[0]):
            print(f"  {ms / 20 * 1e3:8.1f} us/step  {n // 20:3d} launches  {k}")
