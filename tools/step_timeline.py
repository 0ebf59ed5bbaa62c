"""The launches of ONE training step in issue order with their HIP-event durations (eager launches on one stream: HULC_ENC_STREAMS=0), and
the running sum — where the step's kernel time goes, launch by launch.  Framework (aten) launches are not in the list (kernels.py times its
own entry points); `gap` = time between the end of the previous timed launch and the start of this one as the events saw it (eager: host)."""
import os, sys
os.environ.setdefault("HULC_ENC_STREAMS", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
model = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(model.state_dict(), 42)
model.train()
tr = ArenaTrainer(model, lr=2e-4, overlap=False)
batch = syn.make_batch(42, 32, 32, device=dev)
for db in batch.values():
    db.pop("plan_idx", None)
for i in range(3):
    tr.step(batch, i)
torch.cuda.synchronize()
kn.start_timing()
tr.step(batch, 3)
rec, kn._timing = kn._timing, None
torch.cuda.synchronize()
t0 = rec[0][1]
total = 0.0
prev_end = 0.0
for key, e0, e1, fl, by in rec:
    d = e0.elapsed_time(e1) * 1e3
    s = t0.elapsed_time(e0) * 1e3
    total += d
    print(f"{d:8.1f} us  sum {total:8.1f}  {str(key)[:150]}")
print(f"{len(rec)} timed launches, {total:.1f} us of kernels")
