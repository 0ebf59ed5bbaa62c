"""uninitialised-memory hunt: poison the allocator's free blocks with NaN, run an eager step, see what turns non-finite (debugging aid)"""
import os, sys, torch
sys.path.insert(0, '.')
dev = torch.device("cuda", 0)
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
kn.set_compute("bf16")
B, S = int(os.environ.get("B", 4)), int(os.environ.get("S", 16))
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42); m.train()
tr = ArenaTrainer(m, lr=2e-4, overlap=False)
batch = syn.make_batch(42, B, S, device=dev)
for db in batch.values(): db.pop("plan_idx", None)
names = {id(p): n for n, p in m.named_parameters()}
def poison():
    torch.cuda.synchronize()
    junk = [torch.full((int(n),), float("nan"), device=dev) for n in (1, 7, 100, 1000, 5000, 20000, 100000, 10**6, 4 * 10**6, 16 * 10**6, 64 * 10**6) for _ in range(12)]
    del junk
    torch.cuda.synchronize()
for i in range(4):
    poison()
    l = tr.step(batch, i)
    torch.cuda.synchronize()
    bad = [names[id(p)] for p, off in zip(tr.params, tr.offsets) if not torch.isfinite(tr.flat_g[off:off + p.numel()]).all()]
    print(f"eager step {i} loss {float(l):.4f} non-finite grads {len(bad)} {bad[:5]}")
