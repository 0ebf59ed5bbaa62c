"""The persistent MLP-chain launches of a step (csrc/mlp_chain.hip) timed alone, hipGraph replay of 10 repeats each: the prior (5 layers, 64 rows)
forward and its data-gradient chain, the goal-encoder pair (2 x 32 rows) forward / backward.  HULC_LIB selects another build for an A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hulc2_amd import functional as HF, kernels as kn

dev = torch.device("cuda", 0)
kn.set_compute("bf16")


def graph_time(fn, rep=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            for _ in range(rep):
                fn()
    torch.cuda.synchronize()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay(); g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (2 * rep) * 1e3


layers = lambda ls: [(l.weight, l.bias, i < len(ls) - 1) for i, l in enumerate(ls)]
with torch.no_grad():
    dims = (160, 2048, 2048, 2048, 2048, 1024)
    prior = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(64, 160, device=dev)
    print(f"prior forward   (64 rows, 5 layers)      {graph_time(lambda: HF.mlp(x, layers(prior))):7.1f} us")
    x32 = torch.randn(32, 160, device=dev)
    print(f"prior forward   (32 rows, 5 layers)      {graph_time(lambda: HF.mlp(x32, layers(prior))):7.1f} us")
    da, db_ = (128, 2048, 2048, 32), (384, 2048, 2048, 32)
    la = [torch.nn.Linear(a, b).to(dev) for a, b in zip(da[:-1], da[1:])]
    lb = [torch.nn.Linear(a, b).to(dev) for a, b in zip(db_[:-1], db_[1:])]
    xa, xb = torch.randn(32, 128, device=dev), torch.randn(32, 384, device=dev)
    print(f"goal pair forward (2 x 32 rows, 3 layers) {graph_time(lambda: HF.dual_mlp(xa, layers(la), xb, layers(lb))):7.1f} us")
# forward + backward (the data-gradient chains run as chain launches too); weight gradients are not part of the chain
xg = torch.randn(64, 160, device=dev, requires_grad=True)
r = torch.randn(64, 1024, device=dev)


def fb():
    for l in prior:
        l.weight.grad = l.bias.grad = None
    xg.grad = None
    (HF.mlp(xg, layers(prior)) * r).sum().backward()


kn.start_timing()
for _ in range(10):
    fb()
rec = kn.stop_timing()
for k, (n, ms, fl, by) in sorted(rec.items(), key=lambda kv: -kv[1][1]):
    if "chain" in str(k[0]):
        print(f"  prior fwd+bwd (events, eager): {ms / 10 * 1e3:8.1f} us/step  {n // 10:3d} launches  {k}")
