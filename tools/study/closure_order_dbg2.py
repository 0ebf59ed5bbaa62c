import os, sys
from pathlib import Path
import torch
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from hulc2_amd import kernels as kn, synthetic as syn, stepnode
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
dev = torch.device("cuda:0")
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
ls = m.logit_scale
orig_take = stepnode.StepNode._take_live_grads
def take(self, dests):
    tr = self.keeper
    i = [k for k, p in enumerate(tr.params) if p is ls][0]
    g = ls.grad
    print("   take: ls.grad", None if g is None else (hex(g.data_ptr()), float(g)), "dest", None if dests[i] is None else hex(dests[i].data_ptr()), "view", hex(self.views[i].data_ptr()))
    held = orig_take(self, dests)
    print("   held:", None if held is None else (held[0] is not None, [k for k, _ in held[1]].count(i), [(k) for k, _, _ in held[2]]))
    return held
stepnode.StepNode._take_live_grads = take
for i in range(5):
    with torch.autocast("cuda", dtype=torch.float16):
        loss = m.training_step(batch, i)
    node = m.__dict__["_hulc_step_node"]
    lo = node.keeper.flat_g.data_ptr(); hi = lo + 4 * node.keeper.flat_g.numel()
    def w(g):
        return None if g is None else (("arena" if lo <= g.data_ptr() < hi else "other"), float(g))
    print(i, "before zero_grad", w(ls.grad))
    opt.zero_grad(set_to_none=False)
    print(i, "after zero_grad", w(ls.grad))
    scaler.scale(loss).backward()
    print(i, "after backward", w(ls.grad), "graph" if node.graph_fwd is not None else "eager")
    scaler.step(opt)
    scaler.update()
    print(i, "after step", w(ls.grad))
