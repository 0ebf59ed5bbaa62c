"""Framework (aten) launches on the hot path: VERDICT r02 next #8 — the glue ops of a step (concatenations, gradient fan-in adds, fills, strided
copies) were 30 launches of 160; they became kernel epilogues / fused nodes, and this test keeps them from creeping back.  What is left is
counted and named: autograd's fan-in adds of tensors with two consumers."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
pytestmark = pytest.mark.gpu


def test_aten_ops_of_a_training_step(dev):
    from aten_trace import launches
    from hulc2_amd import kernels as kn, synthetic as syn
    from hulc2_amd.compat import instantiate
    from hulc2_amd.config import default_model_config
    from hulc2_amd.trainer import ArenaTrainer
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
    rows = launches(lambda: tr.step(batch, 3))
    listing = "\n".join(f"{op:16s} {shp} {where}" for op, shp, where in rows)
    print(listing)
    big = [r for r in rows if any(len(s) and torch.Size(s).numel() >= 1 << 20 for s in r[1])]
    assert not big, "a framework op touches a large tensor on the hot path:\n" + "\n".join(map(str, big))
    # allowed: gradient fan-in adds of tensors with two consumers (goal, pooled feature) and the prior's input concatenation
    allowed = {"add", "add_", "cat", "clone"}
    other = [r for r in rows if r[0] not in allowed]
    assert not other, "unexpected framework ops on the hot path:\n" + "\n".join(map(str, other))
    assert len(rows) <= 6, f"{len(rows)} framework ops per step (26 at the end of round 2, 5 now):\n" + listing
    tr.close()
