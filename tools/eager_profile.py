"""Host-side profile of the Lightning-style eager loop (bench.py's secondary_lightning_loop, cooperative kernels): cProfile over 10 steps,
top functions by own time and by cumulative time — where the ~3.5 ms of host time per step on top of the kernels go."""
import cProfile
import pstats
import sys
import time

import torch

sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
model = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(model.state_dict(), 42)
model.train()
batch = syn.make_batch(42, 32, 32, device=dev)
for db in batch.values():
    db.pop("plan_idx", None)
opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=2e-4)
scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)


def step(i):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss = model.training_step(batch, i)
    scaler.scale(loss).backward()
    scaler.step(opt)
    scaler.update()
    return loss


for i in range(5):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(10):
    step(i)
torch.cuda.synchronize()
print(f"eager step: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms")
# phases, host time only (no sync inside)
ph = {"zero": 0.0, "fwd": 0.0, "bwd": 0.0, "opt": 0.0}
for i in range(10):
    t = time.perf_counter(); opt.zero_grad(set_to_none=True); ph["zero"] += time.perf_counter() - t
    t = time.perf_counter()
    with torch.autocast("cuda", dtype=torch.float16):
        loss = model.training_step(batch, i)
    ph["fwd"] += time.perf_counter() - t
    t = time.perf_counter(); scaler.scale(loss).backward(); ph["bwd"] += time.perf_counter() - t
    t = time.perf_counter(); scaler.step(opt); scaler.update(); ph["opt"] += time.perf_counter() - t
torch.cuda.synchronize()
print("host time per phase (ms/step, launches queue ahead of the GPU):", {k: round(v / 10 * 1e3, 2) for k, v in ph.items()})
pr = cProfile.Profile()
pr.enable()
for i in range(10):
    step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
st.sort_stats("cumulative").print_stats(30)
