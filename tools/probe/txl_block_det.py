"""determinism probe of the shared-sequence transformer block: repeats of the same launch, which tensors differ"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from hulc2_amd import kernels as kn
import test_txl_block_gpu as T

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
for B, S, p in [(4, 19, 0.1), (4, 19, 0.0), (4, 32, 0.1), (8, 19, 0.1), (64, 19, 0.1), (64, 32, 0.1), (3, 32, 0.0)]:
    enc, pos = T._trunk(3, 2, p)
    enc, pos = enc.to(dev), pos.to(dev)
    g = torch.Generator().manual_seed(4)
    emb, r = torch.randn(B, S, 128, generator=g).to(dev), torch.randn(B, 128, generator=g).to(dev)
    for share in (True, False):
        if share:
            os.environ.pop("HULC_TXL_NO_SHARE", None)
        else:
            os.environ["HULC_TXL_NO_SHARE"] = "1"
        base = T._run(enc, pos, emb, r, p, 0x5EED0001, block=True)
        bad = {}
        for rep in range(6):
            y, dx, got = T._run(enc, pos, emb, r, p, 0x5EED0001, block=True)
            if not torch.equal(y, base[0]):
                bad["y"] = max(bad.get("y", 0), (y - base[0]).abs().max().item())
            if not torch.equal(dx, base[1]):
                bad["dx"] = max(bad.get("dx", 0), (dx - base[1]).abs().max().item())
            for k in got:
                if not torch.equal(got[k], base[2][k]):
                    bad[k] = max(bad.get(k, 0), (got[k] - base[2][k]).abs().max().item())
        print(f"B={B} S={S} p={p} share={share}: {'deterministic' if not bad else bad}", flush=True)
kn.check_faults(dev)
