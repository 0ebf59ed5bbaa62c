import os, sys, faulthandler
from pathlib import Path
import torch
faulthandler.enable()
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
dev = torch.device("cuda:0")


def loop(order, stn, steps=5):
    kn.reset_step_state(dev)
    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 19)
    m.train()
    batch = syn.make_batch(19, 2, 8, device=dev)
    for db in batch.values():
        db.pop("plan_idx", None)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    for i in range(steps):
        if not order:
            opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = m.training_step(batch, i)
        if order:
            opt.zero_grad(set_to_none=stn)
        scaler.scale(loss).backward()
        scaler.step(opt)
        scaler.update()
        print(order, stn, i, float(loss), flush=True)
    return m


which = sys.argv[1] if len(sys.argv) > 1 else "both"
if which in ("std", "both"):
    a = loop(False, True)
if which in ("closure", "both"):
    b = loop(True, False)
print("done")
