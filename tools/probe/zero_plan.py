"""which slices of the gradient arena ArenaTrainer still zeroes every step (parameters whose gradient arrives through autograd's `grad +=`)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
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
names = {id(p): n for n, p in model.named_parameters()}
print("zero ranges:", [(a, b, (b - a) * 4 / 1e6) for a, b in tr._zero_ranges], "MB total", sum(b - a for a, b in tr._zero_ranges) * 4 / 1e6)
from hulc2_amd import gradsink
for p, off in zip(tr.params, tr.offsets):
    if id(p) in tr._autograd_written and not gradsink.written(p):
        print("  autograd-accumulated:", names.get(id(p)), tuple(p.shape), off)
