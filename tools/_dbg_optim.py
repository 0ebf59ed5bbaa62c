import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn, trainer as T
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.optim import Adam
dev = torch.device("cuda", 0)
kn.set_compute("bf16")
batch = syn.make_batch(5, 2, 8, device=dev)
for db in batch.values():
    db.pop("plan_idx", None)
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
syn.fill_state_dict_(m.state_dict(), 11)
m.train()
opt = Adam(m.parameters(), lr=2e-4)
with torch.autocast("cuda", dtype=torch.float16):
    loss = m.training_step(batch, 0)
loss.backward()
ps = [p for p in opt.param_groups[0]["params"] if p.requires_grad]
print("keeper", m.__dict__.get("_hulc_shadow_keeper"), "arenas", len(T._ARENAS))
tr = T._ARENAS.get(ps[0].untyped_storage().data_ptr())
print("tr by storage", tr is not None, "n opt params", len(ps), "n arena", len(tr.params) if tr else None)
if tr:
    mine = {id(p) for p in tr.params}
    print("not in arena:", [n for n, p in m.named_parameters() if id(p) not in mine][:10])
    bad = [(p.shape, p.dtype) for p, off in zip(tr.params, tr.offsets) if p.data_ptr() != tr.flat_p.data_ptr() + off * 4 or p.dtype != torch.float32]
    print("misplaced", bad[:5])
print("arena_of", T.arena_of(ps) is not None)
print("grads None:", [n for n, p in m.named_parameters() if p.requires_grad and p.grad is None][:10])
print("grad dtypes:", {p.grad.dtype for p in ps if p.grad is not None})
