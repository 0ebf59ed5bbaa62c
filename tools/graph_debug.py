import sys, torch
sys.path.insert(0, ".")
from hulc2_amd import kernels as kn, synthetic as syn, shadow
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
dev = torch.device("cuda:0")
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
syn.fill_state_dict_(m.state_dict(), 5); m.train()
tr = ArenaTrainer(m, overlap=False)
batch = syn.make_batch(5, 2, 8, device=dev)
tr.capture(batch)      # 2 eager steps + capture
snap = [t.clone() for t in (tr.flat_p, tr.exp_avg, tr.exp_avg_sq, kn.step_state(dev), tr.flat_bf16)]
def restore():
    for t, s in zip((tr.flat_p, tr.exp_avg, tr.exp_avg_sq, kn.step_state(dev), tr.flat_bf16), snap): t.copy_(s)
# eager fwd/bwd from the snapshot
le = float(tr._forward_backward(batch, 0)); ge = tr.flat_g.clone()
restore()
tr.graph_fb.replay(); torch.cuda.synchronize(); lg = float(tr.static_loss); gg = tr.flat_g.clone()
print("loss eager", le, "graph", lg)
names = [n for n, p in m.named_parameters() if p.requires_grad]
bad = []
for n, p, off in zip(names, tr.params, tr.offsets):
    a, b = ge[off:off+p.numel()], gg[off:off+p.numel()]
    d = (a - b).abs().max().item(); s = a.abs().max().item()
    if d > 1e-6 * max(s, 1e-12): bad.append((n, d, s))
print("params with differing grads:", len(bad), "of", len(names))
for n, d, s in bad[:40]: print(f"  {n:70s} diff {d:.3e} scale {s:.3e}")
restore(); tr.graph_fb.replay(); torch.cuda.synchronize(); print("graph again", float(tr.static_loss), "max grad diff vs first replay", (tr.flat_g - gg).abs().max().item())
