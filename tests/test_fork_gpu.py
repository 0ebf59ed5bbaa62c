"""Round 6: the step's independent branches as branches of one graph (hulc2_amd/models/hulc2.py `_training_step_impl`, kernels.coop_share_scope,
include/hulc2_amd.h hulc_set_coop_share).

reference: hulc2/models/hulc2.py:228-233 — the prior (goal encoders -> plan proposal) and the posterior (plan recognition) are computed from the
same perceptual embedding and do not depend on each other; the contrastive head (hulc2.py:472-508) needs the pooled posterior features and the
goals only.  What must hold: the forked step computes what the one-chain step computes (the transformer trunk shares a sequence between 2 instead
of 4 workgroups: a different summation order of its partial tiles, fp32 rounding), cooperative launches on their share of the device give the
bits of the whole-device launches, and a scope of 0 keeps every cooperative launch out."""
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import functional as HF, kernels as kn, synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402
from hulc2_amd.trainer import ArenaTrainer  # noqa: E402


class _env:
    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        for k, v in self.kv.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v

    def __exit__(self, *exc):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _steps(dev, fork, graph, n=4, B=4, S=16):
    kn.reset_step_state(dev)
    kn.set_compute("bf16")
    with _env(HULC_FORK=fork):
        m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
        syn.fill_state_dict_(m.state_dict(), 3)
        m.train()
        tr = ArenaTrainer(m, lr=2e-4, overlap=False)
        batch = syn.make_batch(3, B, S, device=dev)
        for db in batch.values():
            db.pop("plan_idx", None)
        losses = []
        if graph:                                 # (capture() takes two eager steps of its own in front of the capture)
            for i in range(2):
                losses.append(float(tr.step(batch, i)))
            tr.capture(batch)
            losses += [None, None]
            for i in range(n - 4):
                losses.append(float(tr.replay()))
        else:
            for i in range(n):
                losses.append(float(tr.step(batch, i)))
        torch.cuda.synchronize()
        kn.check_faults(dev)
        return losses, tr.flat_p.clone(), tr.flat_g.clone()


def test_forked_step_equals_the_one_chain_step(dev):
    """four optimizer steps, eager: losses, the last gradient arena and the parameters of the forked arrangement against HULC_FORK=0 — equal up to
    the trunk's summation order (2 instead of 4 workgroups per sequence), i.e. fp32 rounding that the bf16 weights of later steps amplify to 1e-4"""
    if kn.device_cu_count(dev) < 256:
        pytest.skip("the cooperative launches are gated off on this device")
    l1, p1, g1 = _steps(dev, "1", False)
    l0, p0, g0 = _steps(dev, "0", False)
    assert abs(l1[0] - l0[0]) <= 2e-6 * abs(l0[0]), (l1, l0)
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 2e-4 * abs(b), (l1, l0)
    assert (g1 - g0).norm().item() <= 2e-2 * g0.norm().item()
    # parameters: Adam moves an element by ~lr per step whatever the size of its gradient, so an element whose near-zero gradient changes sign
    # walks up to 4 steps x lr = 8e-4 away.  Whether that happens within four steps is a matter of the seed and of the arithmetic elsewhere
    # (tools/study/_fork_seeds.py: the two arrangements stay within 4e-9 of each other for most seeds and part ways — in the 52 tensors upstream
    # of the contrastive head, whose gradient is the small remainder of a sum that cancels — for some, with either flatten-linear operand);
    # asserted: nothing moves further than Adam can, and the parameters stay within 5 % of a step of each other on average
    d = (p1 - p0).abs()
    assert d.max().item() <= 4 * 2e-4 * 1.01 and d.mean().item() <= 0.05 * 2e-4, (d.max().item(), d.mean().item())


def test_forked_graph_replay_equals_forked_eager_steps(dev):
    """the branches inside ONE captured graph (fork, joins, halved cooperative grids baked in) reproduce the eager forked steps bit for bit"""
    if kn.device_cu_count(dev) < 256:
        pytest.skip("the cooperative launches are gated off on this device")
    le, pe, ge = _steps(dev, "1", False, n=7)
    lg, pg, gg = _steps(dev, "1", True, n=7)
    assert le[:2] == lg[:2] and le[4:] == lg[4:], (le, lg)
    assert torch.equal(pe, pg) and torch.equal(ge, gg)


def test_chain_on_half_the_device_gives_the_same_bits(dev):
    """hulc_mlp_chain on 128 workgroups (coop share 2): a 2048-wide layer is 128 column tiles — the other half of the whole-device grid only took
    part in the barriers; a chain wider than its share is refused (the caller then uses per-layer GEMMs); share 0 refuses every chain"""
    kn.set_compute("bf16")
    if not kn.mlp_chain_ok(64, 160, [2048, 2048, 1024], dev):
        pytest.skip("the chain kernel is gated off on this device")
    torch.manual_seed(1)
    dims = (160, 2048, 2048, 1024)
    ls = [torch.nn.Linear(a, b).to(dev) for a, b in zip(dims[:-1], dims[1:])]
    x = torch.randn(64, 160, device=dev)
    layers = [(l.weight, l.bias, i < len(ls) - 1) for i, l in enumerate(ls)]
    with torch.no_grad():
        y1 = HF.mlp(x, layers)
        with kn.coop_share_scope(2):
            y2 = HF.mlp(x, layers)
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert kn.mlp_chain_ok(64, 128, [4096, 128], dev) and not kn.mlp_chain_ok(64, 128, [4096, 128], dev, share=2)
    with kn.coop_share_scope(0):
        assert not kn.mlp_chain_ok(64, 160, [2048, 2048, 1024], dev)
        y0 = HF.mlp(x, layers)                                     # per-layer GEMMs: same values up to the exchange's bf16 rounding
    assert ((y0 - y1).norm() / y1.norm()).item() < 4e-3
    kn.check_faults(dev)


def test_set_coop_share_is_validated_and_restored(dev):
    from hulc2_amd import lib as L
    so = L.load()
    assert so.hulc_set_coop_share(2) == 1 and so.hulc_set_coop_share(1) == 2
    assert so.hulc_set_coop_share(3) < 0 and "1, 2 or 4" in so.hulc_last_error().decode()
    assert so.hulc_set_coop_share(1) == 1
