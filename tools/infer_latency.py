"""Control-loop latency of Hulc2.step: batch 1, one 200x200 + 84x84 frame per step, language goal,
replan every 30 steps (reference default).  Eager launches; prints per-step wall latency (median / p95) and validation throughput."""
import sys, time, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42)
m.eval()
batch = syn.make_batch(1, 1, 64, device=dev)
vis = batch["vis"]
goal = {"lang": batch["lang"]["lang"][:1]}
m.reset()
lat = []
for s in range(64):
    obs = {"rgb_obs": {k: v[:1, s:s + 1] for k, v in vis["rgb_obs"].items()}, "depth_obs": {},
           "robot_obs": vis["robot_obs"][:1, s:s + 1], "robot_obs_raw": vis["state_info"]["robot_obs"][:1, s:s + 1]}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    a = m.step(obs, goal)
    torch.cuda.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
lat = sorted(lat[4:])
print(f"Hulc2.step batch-1 latency: median {lat[len(lat) // 2]:.2f} ms, p95 {lat[int(len(lat) * 0.95)]:.2f} ms, max {lat[-1]:.2f} ms "
      f"(replan steps included; 30 Hz control needs < 33 ms)")
vb = syn.make_batch(2, 32, 32, device=dev)
for db in vb.values():
    db.pop("plan_idx", None)
for _ in range(2):
    m.validation_step(vb, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    m.validation_step(vb, 0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 5
print(f"validation_step (2 x 32 sequences): {dt * 1e3:.2f} ms -> {64 / dt:.0f} sequences/s")
