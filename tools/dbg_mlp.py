import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import functional as HF, kernels as kn
dev = torch.device("cuda", 0)
kn.set_compute(sys.argv[1] if len(sys.argv) > 1 else "bf16")
torch.manual_seed(0)
for (M, dims) in ((32, (4096, 128, 32)), (32, (32, 128, 32)), (2, (4096, 128, 32)), (64, (4096, 128, 32)), (32, (384, 2048, 2048, 32))):
    ws = [torch.nn.Linear(a, b) for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(M, dims[0]); r = torch.randn(M, dims[-1])
    xr = x.clone().requires_grad_(True)
    h = xr
    for i, l in enumerate(ws):
        h = l(h)
        if i < len(ws) - 1: h = torch.relu(h)
    (h * r).sum().backward()
    want = [l.weight.grad.clone() for l in ws] + [xr.grad]
    for l in ws: l.weight.grad = None; l.bias.grad = None
    wsd = [l.to(dev) for l in ws]
    xd = x.to(dev).requires_grad_(True)
    y = HF.mlp(xd, [(l.weight, l.bias, i < len(ws) - 1) for i, l in enumerate(wsd)])
    (y * r.to(dev)).sum().backward()
    got = [l.weight.grad for l in wsd] + [xd.grad]
    rel = lambda a, b: ((a.cpu().double() - b.double()).norm() / b.double().norm()).item()
    print(M, dims, "fwd", f"{rel(y.detach(), h.detach()):.2e}", "dW", [f"{rel(a, b):.2e}" for a, b in zip(got[:-1], want[:-1])], "dx", f"{rel(got[-1], want[-1]):.2e}")
