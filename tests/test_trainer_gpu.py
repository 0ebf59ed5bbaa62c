"""ArenaTrainer / training_step behaviours the advisor flagged after round 1, on the GPU:
  * training_step walks the device RNG word itself — fresh dropout masks and plan samples under ANY trainer
  * weights loaded after the trainer exists reach the kernels (bf16 / transposed / conv-layout shadows re-derived)
  * optimizer state save / restore continues a run bit for bit
  * a barrier-kernel timeout raises on the host and the Adam kernel skips that update (no NaN weights)"""
import os
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import kernels as kn, synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402
from hulc2_amd.lib import HulcKernelError  # noqa: E402
from hulc2_amd.trainer import ArenaTrainer  # noqa: E402


def _model(dev, seed, dropout_p=0.0):
    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=dropout_p)).to(dev)
    syn.fill_state_dict_(m.state_dict(), seed)
    m.train()
    return m


def _batch(dev, seed, B=2, S=8, sampled=False):
    b = syn.make_batch(seed, B, S, device=dev)
    if sampled:
        for db in b.values():
            db.pop("plan_idx", None)
    return b


@pytest.mark.parametrize("dropout_p", [0.0, 0.1])
def test_training_step_draws_fresh_randomness_without_trainer(dev, dropout_p):
    """Lightning + torch.optim path: nobody but training_step advances the RNG word.  Two consecutive calls must see different plan
    samples (and dropout masks); the same word reproduces the same draw; eval mode leaves the word alone."""
    m = _model(dev, 3, dropout_p)
    batch = _batch(dev, 3, sampled=True)
    kn.reset_step_state(dev)
    w0 = int(kn.step_state(dev)[0])
    l1 = float(m.training_step(batch, 0))
    w1 = int(kn.step_state(dev)[0])
    l2 = float(m.training_step(batch, 1))
    w2 = int(kn.step_state(dev)[0])
    assert w0 != w1 != w2 and l1 != l2, (w0, w1, w2, l1, l2)
    assert int(kn.step_state(dev)[1]) == 0                 # the optimizer step count belongs to the trainer
    kn.reset_step_state(dev)
    assert float(m.training_step(batch, 0)) == l1          # same word, same draw
    m.eval()
    w = int(kn.step_state(dev)[0])
    with torch.no_grad():
        m.training_step(_batch(dev, 3), 0)                  # (injected plan indices, dropout off in eval mode)
    assert int(kn.step_state(dev)[0]) == w


def test_trainer_step_advances_rng_once_and_counts_steps(dev):
    m = _model(dev, 4, 0.1)
    tr = ArenaTrainer(m)
    batch = _batch(dev, 4, sampled=True)
    kn.reset_step_state(dev)
    a = kn.step_state(dev).clone()
    tr.step(batch, 0)
    b = kn.step_state(dev).clone()
    ref = (int(a[0]) * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
    assert (int(b[0]) & 0xFFFFFFFFFFFFFFFF) == ref and int(b[1]) == int(a[1]) + 1      # ONE walk per step (the trainer's), one count


def test_weights_loaded_after_trainer_reach_the_kernels(dev):
    """Lightning restores the checkpoint after configure_optimizers: the kernel-side shadows must follow (the post-hook)"""
    batch = _batch(dev, 7)
    fresh = _model(dev, 21)
    tr_f = ArenaTrainer(fresh)
    want = tr_f._forward_backward(batch, 0)
    g_want = tr_f.flat_g.clone()
    m = _model(dev, 20)
    tr = ArenaTrainer(m)
    tr._forward_backward(batch, 0)                          # shadows of the OLD weights are live and cached
    src = {k: v.clone() for k, v in fresh.state_dict().items()}
    m.load_state_dict(src)
    got = tr._forward_backward(batch, 0)
    assert torch.equal(got, want), (float(got), float(want))
    # gradient slices by parameter (the two trainers were built alike: same arena layout)
    assert tr.offsets == tr_f.offsets and torch.equal(tr.flat_g, g_want)


def test_optimizer_state_roundtrip_continues_bitwise(dev):
    batch = _batch(dev, 9, sampled=True)
    kn.reset_step_state(dev)
    m = _model(dev, 9, 0.1)
    tr = ArenaTrainer(m)
    for i in range(2):
        tr.step(batch, i)
    model_sd = {k: v.clone() for k, v in m.state_dict().items()}
    opt_sd = tr.state_dict()
    want = [float(tr.step(batch, i)) for i in range(2, 5)]
    p_want = tr.flat_p.clone()
    kn.reset_step_state(dev, seed=12345)                    # whatever the process did in between
    m2 = _model(dev, 1, 0.1)
    tr2 = ArenaTrainer(m2)
    m2.load_state_dict(model_sd)
    tr2.load_state_dict(opt_sd)
    got = [float(tr2.step(batch, i)) for i in range(2, 5)]
    assert got == want, (got, want)
    assert torch.equal(tr2.flat_p, p_want)


def test_barrier_timeout_raises_and_adam_skips(dev):
    """HULC_RNN_DBG=4 makes the persistent RNN kernel report a barrier timeout: the sticky fault word is set, the fused Adam leaves the
    weights untouched, check_faults raises; afterwards training continues normally."""
    if kn.device_cu_count(dev) < 256:
        pytest.skip("the barrier kernel is gated off on this device")
    m = _model(dev, 11)
    tr = ArenaTrainer(m)
    batch = _batch(dev, 11)
    tr.step(batch, 0)
    kn.check_faults(dev)
    before = tr.flat_p.clone()
    os.environ["HULC_RNN_DBG"] = "4"
    try:
        tr.step(batch, 1)
    finally:
        del os.environ["HULC_RNN_DBG"]
    torch.cuda.synchronize()
    assert torch.equal(tr.flat_p, before), "Adam must not consume the gradients of a faulted step"
    with pytest.raises(HulcKernelError, match="barrier timed out"):
        kn.check_faults(dev)
    kn.check_faults(dev)                                    # cleared by the raise
    loss = float(tr.step(batch, 2))
    assert loss == loss and not torch.equal(tr.flat_p, before)


def test_bench_exits_nonzero_on_barrier_timeout():
    """the same fault inside bench.py's timed region: non-zero exit and no JSON line (never a NaN number)"""
    import subprocess
    if not torch.cuda.is_available() or kn.device_cu_count(torch.device("cuda:0")) < 256:
        pytest.skip("needs a whole MI355X")
    env = dict(os.environ, HULC_RNN_DBG="4")
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "2", "--seq-len", "8",
                        "--no-cpu-baseline", "--no-secondary"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "barrier timed out" in r.stderr, r.stderr[-2000:]
    assert not [l for l in r.stdout.splitlines() if l.strip().startswith("{")]


def test_graph_replay_is_immune_to_eager_work_between_replays(dev):
    """Regression (round 2): a hipMemsetAsync captured inside the recurrent kernel's launcher left the barrier header stale on later
    replays as soon as anything else allocated / filled memory between two replays (NaN from the second replay on) — exactly what an
    RCCL all-reduce or a data loader does.  Replays interleaved with NaN-filled scratch allocations must reproduce the undisturbed run
    bit for bit."""
    def run(disturb):
        kn.reset_step_state(dev)
        m = _model(dev, 13, 0.1)
        tr = ArenaTrainer(m, overlap=False)
        batch = _batch(dev, 13, B=4, S=16, sampled=True)
        for i in range(2):
            tr.step(batch, i)
        tr.capture(batch)
        out = []
        for _ in range(4):
            out.append(float(tr.replay()))
            if disturb:
                torch.cuda.synchronize()
                junk = [torch.full((n,), float("nan"), device=dev) for n in (1, 100, 5000, 100000, 4 * 10**6) for _ in range(6)]
                del junk
        torch.cuda.synchronize()
        return out, tr.flat_p.clone()
    a, pa = run(False)
    b, pb = run(True)
    assert all(x == x for x in b), b
    assert a == b and torch.equal(pa, pb), (a, b)


def test_training_step_under_lightning_amp_gives_the_same_bits(dev):
    """Lightning `precision: 16` (conf/trainer/play_trainer.yaml:3; SURVEY §8b "Lightning wraps the step in autocast"): `training_step` inside
    `torch.autocast("cuda", float16)`, `GradScaler.scale(loss).backward()`, `unscale_` — loss and every gradient must be the bits of the plain
    call (the hooks switch autocast off; the 2^16 loss scale is exact through every backward kernel)."""
    m = _model(dev, 5)
    batch = _batch(dev, 5, B=2, S=8)
    names = [n for n, _ in m.named_parameters()]

    def grads():
        return {n: (p.grad.detach().clone() if p.grad is not None else None) for n, p in m.named_parameters()}

    kn.reset_step_state(dev)
    for p in m.parameters():
        p.grad = None
    loss = m.training_step(batch, 0)
    assert loss.dtype == torch.float32
    loss.backward()
    ref_loss, ref = loss.detach().clone(), grads()

    kn.reset_step_state(dev)
    for p in m.parameters():
        p.grad = None
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    with torch.autocast("cuda", dtype=torch.float16):
        loss2 = m.training_step(batch, 0)
        assert torch.is_autocast_enabled()                  # the hook restored the caller's context
    assert loss2.dtype == torch.float32 and torch.equal(loss2.detach(), ref_loss)
    scaler.scale(loss2).backward()
    scaler.unscale_(opt)
    got = grads()
    n_checked = 0
    for n in names:
        if ref[n] is None:
            assert got[n] is None, n
            continue
        assert got[n] is not None and got[n].dtype == torch.float32, n
        assert torch.equal(got[n], ref[n]), f"{n}: max diff {(got[n] - ref[n]).abs().max().item():.3e}"
        n_checked += 1
    assert n_checked > 60
    scaler.step(opt)                                        # finite gradients: the scaler lets the optimizer step
    scaler.update()
    assert scaler.get_scale() == 65536.0
    # validation / rollout hooks under the same context
    m.eval()
    with torch.autocast("cuda", dtype=torch.float16):
        out = m.validation_step(_batch(dev, 5), 0)
    assert all(v.dtype in (torch.float32, torch.int64, torch.int32) for v in out.values())


def test_external_optimizer_loop_keeps_weight_copies_fresh_with_a_shadows_only_trainer(dev):
    """Round 4: under an external optimizer (Lightning + torch.optim.Adam, hulc2/training.py:79-82) the first training-mode step installs
    ArenaTrainer(shadows_only=True) — parameters re-homed into one arena, all kernel-side weight copies re-derived by a handful of launches
    when the optimizer has stepped.  The loop's losses must be the bits of the lazy per-parameter path (HULC_NO_AUTO_SHADOWS=1), a nudge of
    ONE parameter must be seen, and copy.deepcopy(model) must not drag the keeper along."""
    import copy
    import os

    def run(auto):
        if auto:
            os.environ.pop("HULC_NO_AUTO_SHADOWS", None)
        else:
            os.environ["HULC_NO_AUTO_SHADOWS"] = "1"
        try:
            kn.reset_step_state(dev)
            m = _model(dev, 9)
            batch = _batch(dev, 9, B=2, S=8)
            opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
            losses = []
            for i in range(3):
                opt.zero_grad(set_to_none=True)
                loss = m.training_step(batch, i)
                loss.backward()
                opt.step()
                losses.append(float(loss))
            return m, batch, losses
        finally:
            os.environ.pop("HULC_NO_AUTO_SHADOWS", None)

    m_lazy, _, want = run(False)
    assert "_hulc_shadow_keeper" not in m_lazy.__dict__
    m, batch, got = run(True)
    keeper = m.__dict__.get("_hulc_shadow_keeper")
    assert keeper is not None and keeper.shadows_only and keeper.step_node and keeper.flat_g.numel() == keeper.total   # (round 5: the step node's gradient arena)
    assert got == want, (got, want)
    assert all(p.data_ptr() == keeper.flat_p.data_ptr() + 4 * off for p, off in zip(keeper.params, keeper.offsets))
    # one parameter nudged in place: the next step must run on the new value
    kn.reset_step_state(dev)
    a = float(m.training_step(batch, 0))
    with torch.no_grad():
        m.plan_proposal.fc_model[2].weight.mul_(1.5)
    kn.reset_step_state(dev)
    b = float(m.training_step(batch, 0))
    assert a != b
    m2 = copy.deepcopy(m)
    assert m2.__dict__.get("_hulc_shadow_keeper") is None
    kn.reset_step_state(dev)
    c = float(m2.training_step(batch, 0))
    assert c == b and m2.__dict__["_hulc_shadow_keeper"] is not keeper
    # a full trainer takes over from the keeper
    from hulc2_amd.trainer import ArenaTrainer
    tr = ArenaTrainer(m)
    assert not tr.shadows_only
    float(tr.step(batch, 0))
