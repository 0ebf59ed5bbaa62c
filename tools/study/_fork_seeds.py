"""forked step graph vs one chain (HULC_FORK=1 / 0), three eager optimizer steps from SEED: how many parameters' gradients differ by more than 1e-5
(relative L2) per step.  HULC_A3_NOTWIN=1: the gripper camera's flatten-linear on the bf16 map instead of the exact one.  Round 6, seeds 3 5 7 11 13:
exact map 0/52/104, 0/0/0, 52/106/106, 0/0/0, 52/106/106 — bf16 map 0/0/0, 0/0/56, 0/0/0, 0/0/0, 0/0/0."""
import os, sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch
from hulc2_amd import kernels as kn, synthetic as syn
from hulc2_amd.compat import instantiate
from hulc2_amd.config import default_model_config
from hulc2_amd.trainer import ArenaTrainer
dev = torch.device("cuda:0")
SEED = int(os.environ.get("SEED", "3"))

def run(fork, n):
    os.environ["HULC_FORK"] = fork
    kn.reset_step_state(dev); kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), SEED); m.train()
    tr = ArenaTrainer(m, lr=2e-4, overlap=False)
    batch = syn.make_batch(SEED, 4, 16, device=dev)
    for db in batch.values(): db.pop("plan_idx", None)
    gs = []
    for i in range(n):
        tr.step(batch, i); gs.append(tr.flat_g.clone())
    torch.cuda.synchronize()
    names = {id(p): nm for nm, p in m.named_parameters()}
    return tr, gs, [names[id(p)] for p in tr.params]

tra, ga, names = run(os.environ.get("ARR_A", "1"), 3)
trb, gb, _ = run(os.environ.get("ARR_B", "0"), 3)
for step in range(3):
    worst = []
    for nm, off, p in zip(names, tra.offsets, tra.params):
        a, b = ga[step][off:off + p.numel()], gb[step][off:off + p.numel()]
        rel = ((a - b).norm() / (b.norm() + 1e-30)).item()
        if rel > 1e-5: worst.append((rel, nm))
    worst.sort(reverse=True)
    print("step", step, "params with gradient rel diff > 1e-5:", len(worst))

