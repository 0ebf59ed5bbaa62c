"""Times the whole-trunk transformer launches (csrc/txl_block.hip) at the benchmark's shape: 64 sequences x 32 positions, 2 layers, dropout 0.1.
HULC_TXL_DBG (1 skip attention, 2 skip the feed-forward loops, 4 skip the exchanges; results invalid) splits the time; HULC_TXL_NO_SHARE=1 /
HULC_TXL_SHARE=2 change the workgroups per sequence."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from hulc2_amd import functional as HF, kernels as kn
import test_txl_block_gpu as T

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
B, S = int(os.environ.get("B", 64)), 32
enc, pos = T._trunk(3, 2, 0.1)
enc, pos = enc.to(dev), pos.to(dev)
emb = torch.randn(B, S, 128, device=dev, requires_grad=True)
r = torch.randn(B, 128, device=dev)
ids = torch.arange(S, device=dev)
layers = T._layer_params(enc)


def step():
    for q in enc.parameters():
        q.grad = None
    y = HF.transformer_trunk_pooled(emb, pos.weight, ids, layers, 8, 0.1, 1)
    (y * r).sum().backward()


for _ in range(5):
    step()
kn.start_timing()
for _ in range(20):
    step()
rec = kn.stop_timing()
for k, (n, ms, fl, by) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    print(f"{ms / 20 * 1e3:8.1f} us/step  {n // 20:3d} launches  {k}")
