"""Where do the device-to-device copies of a training step come from?  One eager step under torch.profiler with Python stacks; prints every
aten::copy_ / cat / clone / contiguous that launches a Memcpy DtoD (or a copy kernel), grouped by the innermost hulc2_amd frame."""
import collections
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True)).to(dev)
syn.fill_state_dict_(m.state_dict(), 1)
m.train()
tr = ArenaTrainer(m)
batch = syn.make_batch(1, 32, 32, device=dev)
for _ in range(3):
    tr.step(batch)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(batch)
    torch.cuda.synchronize()
groups = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::cat", "aten::clone", "aten::contiguous", "aten::_to_copy", "aten::index", "aten::index_select", "aten::fill_", "aten::zero_"):
        st = [f for f in (ev.stack or []) if "hulc2_amd" in f or "bench.py" in f]
        groups[(ev.name, st[0] if st else "?")] += 1
for (name, where), c in sorted(groups.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{c:4d}  {name:18s} {where}")
