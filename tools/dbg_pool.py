"""which allocations made WHILE capturing the training graphs land in the default allocator pool (and are therefore recyclable by eager
allocations between replays)?  round 2 debugging aid"""
import os, sys, torch
sys.path.insert(0, '.')
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42); m.train()
tr = ArenaTrainer(m, lr=2e-4, overlap=False)
batch = syn.make_batch(42, 4, 16, device=dev)
for db in batch.values(): db.pop("plan_idx", None)
for i in range(2): tr.step(batch, i)
torch.cuda.synchronize()
torch.cuda.memory._record_memory_history(max_entries=200000, stacks="python")
tr.capture(batch)
torch.cuda.synchronize()
snap = torch.cuda.memory._snapshot()
torch.cuda.memory._record_memory_history(enabled=None)
segs = [(s["address"], s["address"] + s["total_size"], tuple(s.get("segment_pool_id", (0, 0)))) for s in snap["segments"]]
def pool_of(addr):
    for a, b, pid in segs:
        if a <= addr < b: return pid
    return None
tr_ev = snap["device_traces"][0]
print("trace events:", len(tr_ev), "segments:", len(segs), "pools:", sorted({p for _, _, p in segs}))
def names_of(ev):
    return [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", [])]
idx = [i for i, ev in enumerate(tr_ev) if ev["action"] == "alloc" and any(n.startswith("trainer.py") and n.endswith(":capture") for n in names_of(ev))
       and not any(n.endswith(":step") for n in names_of(ev))]
i0, i1 = min(idx), max(idx)
print("capture window: events", i0, "..", i1)
shown = 0
for ev in tr_ev[i0:i1 + 1]:
    if ev["action"] != "alloc": continue
    pid = pool_of(ev["addr"])
    if pid == (0, 0) or pid is None:
        loc = [n for n in names_of(ev) if "site-packages" not in n and "dist-packages" not in n][:7]
        print(f"DEFAULT-POOL alloc inside the capture window: size {ev['size']} stream {ev.get('stream')} :: " + " <- ".join(loc))
        shown += 1
print("done", shown)
