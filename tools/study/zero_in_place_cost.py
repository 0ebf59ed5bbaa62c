"""where the 0.6 ms of the zero-in-place closure order go: GPU event times of the four segments of a step, set_to_none True / False"""
import os, sys, time
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
dev = torch.device("cuda:0")
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42)
m.train()
batch = syn.make_batch(42, 32, 32, device=dev)
for db in batch.values():
    db.pop("plan_idx", None)
opt = m.configure_optimizers()["optimizer"]
scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)


def run(stn, n=30):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in range(n)]
    host = [0.0] * 4
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        e = ev[i]
        e[0].record(); h0 = time.perf_counter()
        with torch.autocast("cuda", dtype=torch.float16):
            loss = m.training_step(batch, i)
        e[1].record(); h1 = time.perf_counter()
        opt.zero_grad(set_to_none=stn)
        e[2].record(); h2 = time.perf_counter()
        scaler.scale(loss).backward()
        e[3].record(); h3 = time.perf_counter()
        scaler.step(opt); scaler.update()
        e[4].record(); h4 = time.perf_counter()
        for k, d in enumerate((h1 - h0, h2 - h1, h3 - h2, h4 - h3)):
            host[k] += d
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n * 1e3
    seg = [sum(ev[i][k].elapsed_time(ev[i][k + 1]) for i in range(5, n)) / (n - 5) for k in range(4)]
    print(f"set_to_none={stn}: wall {wall:.3f} ms/step | GPU segments ms: training_step {seg[0]:.3f} zero_grad {seg[1]:.3f} backward {seg[2]:.3f} step {seg[3]:.3f} (sum {sum(seg):.3f}) | "
          f"host ms: {host[0] / n * 1e3:.3f} {host[1] / n * 1e3:.3f} {host[2] / n * 1e3:.3f} {host[3] / n * 1e3:.3f}")


for _ in range(4):
    with torch.autocast("cuda", dtype=torch.float16):
        loss = m.training_step(batch, 0)
    opt.zero_grad(set_to_none=True)
    scaler.scale(loss).backward(); scaler.step(opt); scaler.update()
run(True); run(False); run(True); run(False)
