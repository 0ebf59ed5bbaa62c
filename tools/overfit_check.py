"""Sanity: repeated optimizer steps on one synthetic batch must drive the loss down (graph replay and eager agree on the trend)."""
import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
for drop, inject in ((0.0, True), (0.0, False), (0.1, True), (0.1, False)):
    for mode in ("graph", "eager"):
        kn.reset_step_state(dev)
        m = instantiate(default_model_config(gripper_control=True, dropout_p=drop)).to(dev)
        syn.fill_state_dict_(m.state_dict(), 42)
        m.train()
        tr = ArenaTrainer(m, lr=2e-4)
        batch = syn.make_batch(42, 8, 32, device=dev)
        if not inject:
            for db in batch.values():
                db.pop("plan_idx", None)
        losses = []
        if mode == "graph":
            tr.capture(batch)
            for i in range(100):
                losses.append(float(tr.replay()))
        else:
            for i in range(102):
                losses.append(float(tr.step(batch, i)))
            losses = losses[2:]
        print(f"drop={drop} inject_plan={inject} {mode:5s}", " ".join(f"{l:.3f}" for l in losses[::11]))
