import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from hulc2_amd import kernels as kn
import test_txl_block_gpu as T

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
for B, S in [(64, 32), (4, 19), (4, 19), (5, 7)]:
    enc, pos = T._trunk(3, 2, 0.1)
    enc, pos = enc.to(dev), pos.to(dev)
    g = torch.Generator().manual_seed(4)
    emb, r = torch.randn(B, S, 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
    y, dx, got = T._run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    junk = [torch.randn(1 << 20, device=dev) for _ in range(8)]          # perturb the allocator's free blocks
    yu, dxu, unf = T._run(enc, pos, emb, r, 0.1, 0x5EED0001, block=False)
    del junk
    y2, dx2, got2 = T._run(enc, pos, emb, r, 0.1, 0x5EED0001, block=True)
    print(B, S, "y equal", torch.equal(y, y2), "dx equal", torch.equal(dx, dx2))
    if not torch.equal(dx, dx2):
        d = (dx - dx2).abs()
        idx = d.nonzero()
        print("  differing elements:", idx.shape[0], "of", dx.numel(), "max", d.max().item(), "batches", idx[:, 0].unique().tolist(), "tokens", idx[:, 1].unique().tolist()[:40],
              "features", idx[:, 2].unique().tolist()[:40])
    for k in got:
        if not torch.equal(got[k], got2[k]):
            print("  grad differs:", k, (got[k] - got2[k]).abs().max().item())
kn.check_faults(dev)
