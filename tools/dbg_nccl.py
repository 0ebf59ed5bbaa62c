"""where does the NaN come from with an RCCL process group + the barrier RNN kernel? (round 2 debugging aid)"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29611")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
mode = sys.argv[1] if len(sys.argv) > 1 else "nccl"
if mode != "none":
    dist.init_process_group(mode, rank=0, world_size=1, **({"device_id": dev} if mode == "nccl" else {}))
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
kn.set_compute("bf16")
m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
syn.fill_state_dict_(m.state_dict(), 42); m.train()
tr = ArenaTrainer(m, lr=2e-4, overlap=False, force_comm=(mode != "none"))
batch = syn.make_batch(42, 4, 16, device=dev)
for db in batch.values(): db.pop("plan_idx", None)
names = {id(p): n for n, p in m.named_parameters()}
def report(tag):
    torch.cuda.synchronize()
    bad = [names[id(p)] for p, off in zip(tr.params, tr.offsets) if not torch.isfinite(tr.flat_g[off:off + p.numel()]).all()]
    badp = [names[id(p)] for p, off in zip(tr.params, tr.offsets) if not torch.isfinite(tr.flat_p[off:off + p.numel()]).all()]
    print(tag, "non-finite grads:", bad[:6], len(bad), "| params:", badp[:4], len(badp), "| fault", int(kn.fault_word(dev).item()))
for i in range(int(os.environ.get('NEAGER', '3'))):
    l = tr.step(batch, i); report(f"eager step {i} loss {float(l):.4f}")
if len(sys.argv) > 2 and sys.argv[2] == "graph":
    tr.capture(batch)
    how = os.environ.get("BETWEEN", "report")
    for i in range(6):
        l = tr.replay()
        if how == "report": report(f"replay {i} loss {float(l):.4f}")
        elif how == "sync": torch.cuda.synchronize()
        elif how == "item": print("loss", float(l))
        elif how == "fault": print("fault", int(kn.fault_word(dev).item()))
        elif how == "isfinite": print(bool(torch.isfinite(tr.flat_g).all()))
        elif how == "alloc":
            torch.cuda.synchronize()
            sizes = [int(v) for v in os.environ.get("SIZES", "1,7,100,1000,5000,20000,100000,1000000,4000000").split(",")]
            junk = [torch.full((int(n),), float("nan"), device=dev) for n in sizes for _ in range(8)]
            del junk
        elif how == "empty": torch.cuda.synchronize(); torch.cuda.empty_cache()
    report(f"final ({how}) loss {float(l):.4f}")
