"""The optimizer leg of the hot path: hulc_adam_step against torch.optim.Adam, and the whole training loop
(forward + backward + arena Adam + shadow refresh, several steps) against the CPU oracle driven by torch.optim.Adam."""
import sys
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
pytestmark = pytest.mark.gpu

from hulc2_amd import param_spec, synthetic as syn  # noqa: E402
from hulc2_amd.compat import instantiate  # noqa: E402
from hulc2_amd.config import default_model_config  # noqa: E402
from oracle import hulc2_oracle as O  # noqa: E402  (checker only)


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    return torch.device("cuda", 0)


@pytest.mark.parametrize("device_step", [False, True])
def test_adam_kernel_matches_torch(dev, device_step):
    """torch.optim.Adam(lr, betas=(0.9, 0.999), eps=1e-8) semantics (hulc2.py:185-198), 1/world gradient scale folded in,
    bf16 shadow = the updated weights rounded to nearest even"""
    from hulc2_amd import kernels as kn

    n = 100003 // 8 * 8
    g = torch.Generator().manual_seed(3)
    p0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** (i - 2)) for i in range(5)]
    ref = torch.nn.Parameter(p0.clone().double())
    opt = torch.optim.Adam([ref], lr=2e-4)
    p, m, v = p0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    shadow = torch.zeros(n, dtype=torch.bfloat16, device=dev)
    kn.reset_step_state(dev)
    world = 4.0
    for i, gr in enumerate(grads):
        ref.grad = gr.double()
        opt.step()
        if device_step:
            kn.advance_step_state(dev)
        kn.adam_step(p, (gr * world).to(dev), m, v, shadow, n, 2e-4, 0.9, 0.999, 1e-8, 0.0, i + 1, grad_scale=1.0 / world,
                     step_state_dev=kn.step_state(dev) if device_step else None)
    torch.cuda.synchronize()
    err = (p.double().cpu() - ref.detach()).abs().max().item()
    assert err < 1.5e-6, f"Adam parameters after 5 steps: max err {err:.3e}"      # fp32 rounding of O(1) parameters (ulp 2.4e-7 .. 4.8e-7)
    assert torch.equal(shadow, p.to(torch.bfloat16)), "bf16 shadow must be the rounded updated weights"
    kn.reset_step_state(dev)


def _oracle_batch(raw):
    out = {}
    for mname, db in raw.items():
        out[mname] = dict(rgb_static=db["rgb_obs"]["rgb_static"], rgb_gripper=db["rgb_obs"]["rgb_gripper"], actions=db["actions"],
                          robot_obs=db["state_info"]["robot_obs"], plan_idx=db["plan_idx"])
        if mname == "lang":
            out[mname].update(lang=db["lang"], use_for_aux_lang_loss=db["use_for_aux_lang_loss"])
    return out


@pytest.mark.parametrize("mode,tol", [("fp32", 2e-4), ("bf16", 1.5e-2)])   # bf16: Adam normalises gradients, so rounding-level gradient differences move the trajectory by ~0.5 %
def test_training_loop_tracks_oracle(dev, mode, tol):
    """four optimizer steps on one batch: per-step losses of the HIP trainer (eager and hipGraph replay) follow the oracle's"""
    from hulc2_amd import kernels as kn
    from hulc2_amd.trainer import ArenaTrainer

    B, S, seed, steps = 2, 8, 17, 4
    raw = syn.make_batch(seed, B, S)
    sd = {k: torch.empty(s) for k, s in param_spec.trainable_shapes().items()}
    syn.fill_state_dict_(sd, seed)
    for t in sd.values():
        t.requires_grad_(True)
    opt = torch.optim.Adam(list(sd.values()), lr=2e-4)
    want = []
    for _ in range(steps):
        opt.zero_grad(set_to_none=True)
        loss = O.training_step(sd, _oracle_batch(raw), dict(gripper_control=True))["total_loss"]
        loss.backward()
        opt.step()
        want.append(float(loss))
    assert want[-1] < want[0], "the oracle itself must be learning"

    kn.set_compute(mode)
    try:
        for use_graph in (False, True):
            kn.reset_step_state(dev)
            m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
            syn.fill_state_dict_(m.state_dict(), seed)
            m.train()
            tr = ArenaTrainer(m, lr=2e-4)
            batch = syn.make_batch(seed, B, S, device=dev)
            if use_graph:
                got = [float(tr.step(batch, i)) for i in range(2)]      # capture() itself runs eager steps: count them
                tr2_losses = []
                tr.capture(batch)                                        # +2 eager steps (steps 3 and 4 of the sequence)
                got = got + tr2_losses
                # after capture the parameters have seen 4 updates: the first replay is step 5 — compare the eager prefix only
                assert all(abs(a - b) <= tol * abs(b) for a, b in zip(got, want[:2])), (mode, "graph prefix", got, want)
                l5 = float(tr.replay())
                assert l5 == l5 and l5 < want[0], "replay continues the descent"
            else:
                got = [float(tr.step(batch, i)) for i in range(steps)]
                assert all(abs(a - b) <= tol * abs(b) for a, b in zip(got, want)), (mode, "eager", got, want)
    finally:
        kn.set_compute("bf16")
        kn.reset_step_state(dev)


def test_first_write_overwrite_equals_zeroed_arena(dev, monkeypatch):
    """From the second step on the trainer stops zeroing the gradient slices the backward kernels write (their first writer overwrites,
    gradsink.first_write): parameters after 4 steps plus one vision-only step (sinks the plan expects but nobody writes) are bit-identical to a trainer that zeroes the whole arena every step."""
    from hulc2_amd import kernels as kn
    from hulc2_amd.trainer import ArenaTrainer

    kn.set_compute("bf16")
    batch = syn.make_batch(5, 2, 8, device=dev)
    finals, planned = [], []
    for full in (True, False):
        if full:
            monkeypatch.setenv("HULC_FULL_ZERO_GRAD", "1")
        else:
            monkeypatch.delenv("HULC_FULL_ZERO_GRAD", raising=False)
        m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
        syn.fill_state_dict_(m.state_dict(), 21)
        m.train()
        tr = ArenaTrainer(m, lr=2e-4)
        kn.reset_step_state(dev)
        for i in range(4):
            tr.step(batch, i)
        tr.step({"vis": batch["vis"]}, 4)                              # a step in which the language-only parameters get no gradient at all
        torch.cuda.synchronize()
        finals.append(tr.flat_p.clone())
        planned.append(tr._zero_ranges)
    assert planned[0] is None and planned[1] is not None
    zeroed = sum(b - a for a, b in planned[1])
    assert zeroed < 0.2 * finals[0].numel(), "most of the arena is written by sinks and no longer zeroed"
    assert torch.equal(finals[0], finals[1])


@pytest.mark.parametrize("shapes", [[(2048, 128), (128, 2048), (2048, 2048)], [(64, 72), (100, 36), (33, 17), (8, 8), (520, 24)]])
def test_transposed_shadow_tiles(dev, shapes):
    """hulc_transpose_bf16_tiles (the per-step refresh of the transposed bf16 weight shadows): all three access widths (16-, 8-, 2-byte:
    dimensions / offsets that are multiples of 8, of 4, or neither) against torch's transpose, bit for bit"""
    from hulc2_amd import kernels as kn
    offs, tiles, total = [], [], 0
    for r, c in shapes:
        offs.append(total)
        tiles += [(total, r, c, i, j) for i in range((r + 63) // 64) for j in range((c + 63) // 64)]
        total += r * c
    src = torch.randn(total, device=dev).to(torch.bfloat16)
    dst = torch.zeros(total, dtype=torch.bfloat16, device=dev)
    kn.transpose_bf16_tiles(src, dst, torch.tensor(tiles, dtype=torch.int64, device=dev))
    torch.cuda.synchronize()
    for (r, c), off in zip(shapes, offs):
        want = src[off:off + r * c].view(r, c).t().contiguous().view(-1)
        assert torch.equal(dst[off:off + r * c], want), (r, c)


def test_derived_copies_after_a_step_equal_a_fresh_derivation(dev):
    """Round 4: the Adam kernel writes the split operands' rounding remainders itself (hulc_adam_step_lo, <= 8 element ranges of a second shadow
    arena) and the other derived weight copies come from two launches (hulc_derive_copies: transposed tiles + conv repacks; hulc_gather_chunks2:
    fragment-packed copies of the shadow and of the remainders).  After optimizer steps every derived copy must be bit-identical to what
    refresh_shadows() derives from the fp32 arena with the single-purpose kernels (cast, transpose, gather, residual, repack)."""
    from hulc2_amd import kernels as kn
    from hulc2_amd.trainer import ArenaTrainer

    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=0.1)).to(dev)
    syn.fill_state_dict_(m.state_dict(), 3)
    m.train()
    tr = ArenaTrainer(m, lr=2e-4)
    assert tr.flat_lo is not None and 1 <= len(tr.lo_ranges) <= 8 and tr.lo_frag_idx is not None
    batch = syn.make_batch(9, 2, 8, device=dev)
    kn.reset_step_state(dev)
    for i in range(3):
        tr.step(batch, i)
    torch.cuda.synchronize()
    names = ["flat_bf16", "flat_bf16_t", "frag_shadow", "lo_frag", "conv_shadow"]
    got = {n: getattr(tr, n).clone() for n in names}
    lo_got = tr.flat_lo.clone()
    tr.refresh_shadows()
    torch.cuda.synchronize()
    for n in names:
        assert torch.equal(got[n].view(torch.int16), getattr(tr, n).view(torch.int16)), n
    # the remainders: identical on every registered operand (between merged ranges the fused pass may write more than the operands)
    for off, cnt, _ in tr.lo_seg.tolist():
        assert torch.equal(lo_got[off:off + cnt].view(torch.int16), tr.flat_lo[off:off + cnt].view(torch.int16)), off
    # and they are what they claim to be: hi + lo reproduces the fp32 weight to ~2^-17 relative
    off, cnt, _ = max(tr.lo_seg.tolist(), key=lambda r: r[1])
    w = tr.flat_p[off:off + cnt]
    err = (w - (tr.flat_bf16[off:off + cnt].float() + tr.flat_lo[off:off + cnt].float())).abs().max() / w.abs().max()
    assert float(err) < 2e-5, float(err)


def test_adam_lo_ranges_are_validated(dev):
    """hulc_adam_step_lo refuses more than 8 ranges / unaligned starts (error code + message, no launch)"""
    from hulc2_amd import kernels as kn
    from hulc2_amd.lib import HulcKernelError
    n = 64
    p, g, mm, v = (torch.zeros(n, device=dev) for _ in range(4))
    sh, lo = torch.zeros(n, dtype=torch.bfloat16, device=dev), torch.zeros(n, dtype=torch.bfloat16, device=dev)
    with pytest.raises(HulcKernelError):
        kn.adam_step(p, g, mm, v, sh, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, lo=lo, lo_ranges=[(2, 8)])
    with pytest.raises(HulcKernelError):
        kn.adam_step(p, g, mm, v, sh, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, lo=lo, lo_ranges=[(4 * i, 4 * i + 4) for i in range(9)])
    g.fill_(1.0)
    kn.adam_step(p, g, mm, v, sh, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, lo=lo, lo_ranges=[(8, 16), (32, 64)])
    torch.cuda.synchronize()
    want = (p - sh.float()).to(torch.bfloat16)
    inside = torch.zeros(n, dtype=torch.bool, device=dev)
    inside[8:16] = True
    inside[32:64] = True
    assert torch.equal(lo[inside].view(torch.int16), want[inside].view(torch.int16)) and float(lo[~inside].float().abs().max()) == 0.0


def test_drop_in_adam_takes_the_arena_step(dev):
    """hulc2_amd.optim.Adam in the reference's own loop (hulc2.py:185-198: `optimizer._target_`; training.py:79-82: training_step -> backward ->
    optimizer.step under fp16 autocast + GradScaler): once Hulc2 has moved its parameters into the keeper's arena every step is the fused arena
    launch.  Fed the SAME gradients, torch.optim.Adam on clones of the parameters lands on the same values after three steps — parameters within
    2e-6 of their scale + 1e-3 lr, moments within 1e-6 of their largest entry (fp32 rounding of two evaluation orders) — the state_dict has torch's
    layout (no entries for the two parameters the step never reaches), and each optimizer resumes from the other's checkpoint."""
    import copy
    from hulc2_amd import kernels as kn
    from hulc2_amd.optim import Adam

    kn.set_compute("bf16")
    try:
        lr = 2e-4
        batch = syn.make_batch(5, 2, 8, device=dev)
        for db in batch.values():
            db.pop("plan_idx", None)
        m = instantiate(default_model_config(gripper_control=True, dropout_p=0.0)).to(dev)
        syn.fill_state_dict_(m.state_dict(), 11)
        m.train()
        opt = Adam(m.parameters(), lr=lr)
        clones = [torch.nn.Parameter(p.detach().clone()) for p in m.parameters()]
        ref = torch.optim.Adam(clones, lr=lr)
        kn.reset_step_state(dev)
        scaler = torch.amp.GradScaler("cuda", init_scale=1024.0)
        for i in range(3):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = m.training_step(batch, i)
            scaler.scale(loss).backward()
            scaler.unscale_(opt)
            for c, p in zip(clones, m.parameters()):
                c.grad = None if p.grad is None else p.grad.detach().clone()
            scaler.step(opt)
            scaler.update()
            ref.step()
        torch.cuda.synchronize()
        assert opt.fused_launches == 3, "every step of hulc2_amd.optim.Adam should have been the arena launch"
        for (n, p), c in zip(m.named_parameters(), clones):
            assert float((p - c).abs().max()) <= 2e-6 * max(float(c.abs().max()), 1.0) + 1e-3 * lr, n
        sa, sb = ref.state_dict(), opt.state_dict()
        assert sa["param_groups"][0]["params"] == sb["param_groups"][0]["params"] and sa["state"].keys() == sb["state"].keys()
        assert len(sb["state"]) == len(clones) - 2                   # (plan_recognition.layernorm is not on the step's path: no gradient, no state)
        for k in sa["state"]:
            assert float(sa["state"][k]["step"]) == float(sb["state"][k]["step"]) == 3.0
            for key in ("exp_avg", "exp_avg_sq"):
                x, y = sa["state"][k][key], sb["state"][k][key]
                assert float((x - y).abs().max()) <= 1e-6 * max(float(x.abs().max()), 1e-30), (k, key)
        # a step in which a parameter that HAS moments misses its gradient: torch skips it (no update, its step stays behind) — this optimizer takes
        # torch's per-tensor path for that step and is fused again in the next one, on the state that path left
        for i in (3, 4):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = m.training_step(batch, i)
            loss.backward()
            if i == 3:
                next(iter(m.parameters())).grad = None
            for c, p in zip(clones, m.parameters()):
                c.grad = None if p.grad is None else p.grad.detach().clone()
            opt.step()
            ref.step()
        assert opt.fused_launches == 3, "two parameters now have different step counts: torch's per-tensor path from here on"
        for (n, p), c in zip(m.named_parameters(), clones):
            assert float((p - c).abs().max()) <= 3e-6 * max(float(c.abs().max()), 1.0) + 1e-3 * lr, n
        sa, sb = ref.state_dict(), opt.state_dict()
        assert float(sb["state"][0]["step"]) == float(sa["state"][0]["step"]) == 4.0 and float(sb["state"][1]["step"]) == 5.0
        twin = torch.optim.Adam(m.parameters(), lr=lr)
        twin.load_state_dict(copy.deepcopy(sb))                      # torch's optimizer resumes from this one's checkpoint and the other way round
        opt.load_state_dict(copy.deepcopy(sa))
    finally:
        kn.reset_step_state(dev)
