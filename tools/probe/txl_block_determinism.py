import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from hulc2_amd import kernels as kn
import test_txl_block_gpu as T

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
share = os.environ.get("SHARE", "1") == "1"
if not share:
    os.environ["HULC_TXL_NO_SHARE"] = "1"
nbad = 0
for it in range(int(os.environ.get("ITERS", "150"))):
    B, S = [(4, 19), (64, 32), (11, 32), (5, 7)][it % 4]
    enc, pos = T._trunk(3, 2, 0.1)
    enc, pos = enc.to(dev), pos.to(dev)
    g = torch.Generator().manual_seed(4)
    emb, r = torch.randn(B, S, 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
    y, dx, got = T._run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    if it % 3 == 0:
        yu, dxu, unf = T._run(enc, pos, emb, r, 0.1, 0x5EED0001, block=False)
    junk = [torch.randn((it * 7919) % 100000 + 1000, device=dev) for _ in range(it % 5)]
    y2, dx2, got2 = T._run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    bad = []
    if not torch.equal(y, y2):
        bad.append(("y", (y - y2).abs().max().item()))
    if not torch.equal(dx, dx2):
        d = (dx - dx2).abs()
        idx = d.nonzero()
        bad.append(("dx", d.max().item(), idx.shape[0], idx[:, 0].unique().tolist(), idx[:, 1].unique().tolist()[:40], idx[:, 2].unique().tolist()[:64]))
    for k in got:
        if not torch.equal(got[k], got2[k]):
            bad.append((k, (got[k] - got2[k]).abs().max().item()))
    if bad:
        nbad += 1
        print(it, B, S, bad, flush=True)
print("iterations with a difference:", nbad, "share", share)
kn.check_faults(dev)
