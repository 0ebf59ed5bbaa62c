"""Where do the small torch kernels of a step come from?  Lists aten ops with their Python call sites."""
import sys, torch, collections
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42)
m.train()
tr = ArenaTrainer(m, lr=2e-4)
batch = syn.make_batch(42, 32, 32, device=dev)
for db in batch.values():
    db.pop("plan_idx", None)
for i in range(2):
    tr.step(batch, i)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=True) as prof:
    tr.step(batch, 2)
torch.cuda.synchronize()
want = ("aten::add", "aten::add_", "aten::copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::mul", "aten::div", "aten::sum",
        "aten::_to_copy", "aten::index", "aten::stack", "aten::where", "aten::mean", "aten::neg", "aten::sub", "aten::eq", "aten::index_put_",
        "aten::slice_backward", "aten::select_backward", "aten::masked_fill_", "aten::arange", "aten::expand")
agg = collections.Counter()
for ev in prof.key_averages(group_by_input_shape=True):
    if ev.key in want:
        agg[(ev.key, str(ev.input_shapes)[:110])] += ev.count
for (k, shp), n in sorted(agg.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print(f"{n:4d}  {k:22s} {shp}")

# ---- call sites (first frame inside hulc2_amd) of the ops that launch a kernel
sites = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::fill_", "aten::add_", "aten::add", "aten::cat", "aten::zero_", "aten::mul", "aten::div", "aten::index_select"):
        fr = next((f for f in (ev.stack or []) if "hulc2_amd" in f), "?")
        sites[(ev.name, fr.split("hulc2_amd/")[-1][:90])] += 1
print("\ncall sites:")
for (k, fr), n in sorted(sites.items(), key=lambda kv: -kv[1])[:60]:
    print(f"{n:4d}  {k:14s} {fr}")
