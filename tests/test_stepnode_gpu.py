"""The training step as ONE autograd node (hulc2_amd/stepnode.py) inside the reference's own trainer loop.

reference: hulc2/training.py:72-82 (Lightning: training_step -> backward -> optimizer.step), conf/trainer/play_trainer.yaml:3 (`precision: 16`:
torch.autocast + GradScaler).  The node must be invisible to that loop: same losses, same gradients, same parameters as the plain call
(HULC_NO_STEP_NODE=1 = round 4's loop of ~160 autograd Functions), eager or as two replayed hipGraphs; a batch at new addresses is copied
into the graphs' input buffers; no_grad calls and validation take the plain path; gradients still attached at backward time are added back; hulc2_amd.optim.Adam reads the
gradient arena in place and applies a GradScaler's device scalars without a host synchronisation."""
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


def _model(dev, seed, dropout_p=0.1):
    kn.set_compute("bf16")
    m = instantiate(default_model_config(gripper_control=True, dropout_p=dropout_p)).to(dev)
    syn.fill_state_dict_(m.state_dict(), seed)
    m.train()
    return m


def _batch(dev, seed, B=2, S=8):
    b = syn.make_batch(seed, B, S, device=dev)
    for db in b.values():
        db.pop("plan_idx", None)
    return b


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


def _amp_loop(dev, steps, opt_cls=torch.optim.Adam, seed=31, B=2, S=8, fresh_batches=False, keep=()):
    """the Lightning precision-16 loop; -> (model, losses, {step: {name: scaled gradient}} for the steps in `keep`, parameters at the end)"""
    kn.reset_step_state(dev)
    m = _model(dev, seed)
    batch = _batch(dev, seed, B, S)
    opt = opt_cls([p for p in m.parameters() if p.requires_grad], lr=2e-4)
    scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
    losses, grads = [], {}
    for i in range(steps):
        if fresh_batches:            # the same values at new addresses, as a data loader delivers them
            batch = {k: {kk: ({k3: v3.clone() for k3, v3 in vv.items()} if isinstance(vv, dict) else (vv.clone() if torch.is_tensor(vv) else vv))
                         for kk, vv in db.items()} for k, db in batch.items()}
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.float16):
            loss = m.training_step(batch, i)
        scaler.scale(loss).backward()
        if i in keep:
            grads[i] = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()}
        scaler.step(opt)
        scaler.update()
        losses.append(float(loss))
    torch.cuda.synchronize()
    kn.check_faults(dev)
    return m, losses, grads, {n: p.detach().clone() for n, p in m.named_parameters()}, opt


def _same(a, b, what):
    assert a.keys() == b.keys()
    bad = []
    for n in a:
        if a[n] is None or b[n] is None:
            if not (a[n] is None and b[n] is None):
                bad.append((n, "None on one side"))
        elif not torch.equal(a[n], b[n]):
            bad.append((n, float((a[n] - b[n]).abs().max())))
    assert not bad, f"{what}: {len(bad)} tensors differ, e.g. {bad[:4]}"


def test_graphed_node_equals_eager_node_bitwise(dev):
    """two replayed hipGraphs against the same node run eagerly (HULC_NO_STEP_GRAPH=1): losses of five steps, the scaled gradients of the
    first and second replayed step, and the parameters after five optimizer steps — bit for bit.  The node captures once and replays after it."""
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH="1"):
        m_e, l_e, g_e, p_e, _ = _amp_loop(dev, 5, keep=(2, 3))
    node_e = m_e.__dict__["_hulc_step_node"]
    assert node_e.captures == 0 and node_e.eager_steps == 5
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None):
        m_g, l_g, g_g, p_g, _ = _amp_loop(dev, 5, keep=(2, 3))
    node = m_g.__dict__["_hulc_step_node"]
    assert node.disabled is None, node.disabled
    assert node.captures == 1 and node.replays == 3 and node.eager_steps == 2 and node.input_copies == 0
    assert l_g == l_e, (l_g, l_e)
    for i in (2, 3):
        _same(g_g[i], g_e[i], f"scaled gradients of step {i}")
    _same(p_g, p_e, "parameters after five steps")


def test_step_node_equals_the_plain_loop(dev):
    """the node (gradient sinks, grouped weight-gradient launch, arena views handed to autograd) against round 4's loop of per-Function autograd
    (HULC_NO_STEP_NODE=1) under autocast(fp16) + GradScaler: the forward is the same launches — the loss of the first step bit for bit — and so is
    the backward except where a gradient sink selects another kernel for the same sum: the bias gradient of the gripper encoder's flatten-linear is
    a row sum inside the grouped weight-gradient launch instead of one inside the tiled GEMM (two summation orders: 2e-7 relative).  Every other
    gradient of the first step is identical bit for bit; over four optimizer steps losses and parameters stay within fp32 rounding of each other."""
    with _env(HULC_NO_STEP_NODE="1"):
        m_p, l_p, g_p, p_p, _ = _amp_loop(dev, 4, keep=(0,))
    assert "_hulc_step_node" not in m_p.__dict__
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None):
        m_n, l_n, g_n, p_n, _ = _amp_loop(dev, 4, keep=(0,))
    assert m_n.__dict__["_hulc_step_node"].replays == 2
    assert l_n[0] == l_p[0], (l_n, l_p)
    other_kernel = {"perceptual_encoder.rgb_gripper_encoder.conv_model.7.bias"}
    a, b = g_n[0], g_p[0]
    assert a.keys() == b.keys()
    for n in a:
        assert (a[n] is None) == (b[n] is None), n
        if a[n] is None:
            continue
        if n in other_kernel:
            assert float((a[n] - b[n]).abs().max()) <= 2e-6 * float(b[n].abs().max()), n
        else:
            assert torch.equal(a[n], b[n]), (n, float((a[n] - b[n]).abs().max()))
    for x, y in zip(l_n, l_p):
        assert abs(x - y) <= 1e-5 * abs(y), (l_n, l_p)
    for n in p_p:                       # (Adam moves an element by ~lr per step whatever its gradient's size: a 2e-7 change of a near-zero gradient is visible)
        assert float((p_n[n] - p_p[n]).abs().max()) <= 0.05 * 2e-4 * 4, n


def test_batches_at_new_addresses_are_copied_into_the_graph_inputs(dev):
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH="1"):
        _, l_e, _, p_e, _ = _amp_loop(dev, 5, fresh_batches=True)
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None):
        m, l_g, _, p_g, _ = _amp_loop(dev, 5, fresh_batches=True)
    node = m.__dict__["_hulc_step_node"]
    assert node.replays == 3 and node.input_copies >= 2            # (the small tensors of a batch are copied into the graphs' input buffers)
    # ... the frame tensors are not: conv1's captured launches read them through device pointer slots (hulc_conv_desc.x_slot), verified by the
    # capture's self-check (slots redirected to copies, the graphs' own buffers poisoned: same bits), and every step moved four pointers
    assert node.slots_ok and len(node.slot_idx) == 4 and node.slot_updates >= 3, (node.slots_ok, node.slot_idx, node.slot_updates)
    assert l_g == l_e
    _same(p_g, p_e, "parameters")
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None, HULC_NO_FRAME_SLOTS="1"):
        m2, l_c, _, p_c, _ = _amp_loop(dev, 5, fresh_batches=True)
    assert not m2.__dict__["_hulc_step_node"].slots_ok and l_c == l_e
    _same(p_c, p_e, "parameters (frames copied)")


def test_accumulation_validation_and_layout_changes_leave_the_graphs(dev):
    """a live .grad (gradient accumulation over two calls) stays on the graphs and the sum is the sum (round 6); what the graphs cannot express: another
    batch layout -> eager node, then a new capture; validation_step right after optimizer.step() reads the NEW weights (the keeper's
    copies are refreshed at every forward entry point)"""
    kn.reset_step_state(dev)
    m = _model(dev, 7, dropout_p=0.0)
    batch = syn.make_batch(7, 2, 8, device=dev)                      # (injected plan indices: every call draws the same plan)
    opt = torch.optim.Adam(m.parameters(), lr=2e-4)
    for i in range(4):
        opt.zero_grad(set_to_none=True)
        m.training_step(batch, i).backward()
    node = m.__dict__["_hulc_step_node"]
    assert node.disabled is None and node.replays == 2
    g1 = {n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None}
    m.training_step(batch, 4).backward()                            # accumulate on top: the node keeps the attached gradients and adds them back
    assert node.replays == 3 and node.accum_steps == 1
    for n, p in m.named_parameters():
        if n in g1:
            assert torch.equal(p.grad, 2 * g1[n]), n                 # (same plan, no dropout: the second pass is the first, bit for bit)
    # another layout
    small = syn.make_batch(7, 2, 4, device=dev)
    for i in range(3):
        opt.zero_grad(set_to_none=True)
        m.training_step(small, i).backward()
    assert node.captures == 2 and node.replays == 4
    # back to the first layout (the last, smaller batch of an epoch, then the next epoch's full batches): its graphs were kept — no new capture
    opt.zero_grad(set_to_none=True)
    m.training_step(batch, 7).backward()
    assert node.captures == 2 and node.replays == 5
    opt.zero_grad(set_to_none=True)
    m.training_step(small, 8).backward()
    assert node.captures == 2 and node.replays == 6
    # no_grad call of a training-mode model: plain path, no gradient
    with torch.no_grad():
        out = m.training_step(small, 0)
    assert not out.requires_grad
    # validation after an optimizer step == the lazy per-parameter path on the same weights
    opt.zero_grad(set_to_none=True)
    m.training_step(small, 9).backward()
    opt.step()
    m.eval()
    from hulc2_amd.models.hulc2 import Hulc2
    Hulc2._plan_calls = 0                                           # (validation samples its plans from a per-call counter stream)
    got = m.validation_step(syn.make_batch(8, 2, 4, device=dev), 0)
    with _env(HULC_NO_AUTO_SHADOWS="1"):
        m2 = _model(dev, 1, dropout_p=0.0)
        m2.load_state_dict({k: v.clone() for k, v in m.state_dict().items()})
        m2.eval()
        Hulc2._plan_calls = 0
        want = m2.validation_step(syn.make_batch(8, 2, 4, device=dev), 0)
    for k in want:
        assert torch.equal(got[k], want[k]), k


@pytest.mark.parametrize("set_to_none", [False, True])
def test_lightning_closure_order_stays_on_the_graphs(dev, set_to_none):
    """ADVICE r05 (medium) / VERDICT r05 weak #14.  pytorch-lightning 1.8's automatic optimization runs, inside `optimizer.step(closure)`:
    training_step -> optimizer.zero_grad() -> backward — so from the second step on the previous step's `.grad` (views of the node's
    gradient arena) is attached while training_step runs, and with the reference's torch 1.12 default `set_to_none=False` it is STILL attached,
    zeroed in place, when backward runs.  Round 5 sent every such step down the plain path (7.9 ms, no graphs).  Now the node runs regardless:
    both orders must replay the graphs and reach bit-identical losses and parameters; the in-place-zeroed variant goes through the add-back."""
    def loop(lightning_order):
        kn.reset_step_state(dev)
        m = _model(dev, 19)
        batch = _batch(dev, 19)
        opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=2e-4)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        losses = []
        for i in range(6):
            if not lightning_order:
                opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = m.training_step(batch, i)
            if lightning_order:
                opt.zero_grad(set_to_none=set_to_none)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(float(loss))
        torch.cuda.synchronize()
        kn.check_faults(dev)
        return m, losses, {n: p.detach().clone() for n, p in m.named_parameters()}
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None):
        m_a, l_a, p_a = loop(False)
        m_b, l_b, p_b = loop(True)
    na, nb = m_a.__dict__["_hulc_step_node"], m_b.__dict__["_hulc_step_node"]
    assert na.replays == 4 and nb.replays == 4 and nb.disabled is None, (na.replays, nb.replays, nb.disabled)
    assert nb.accum_steps == (0 if set_to_none else 5), nb.accum_steps
    assert l_a == l_b, (l_a, l_b)
    _same(p_b, p_a, "parameters after six steps in Lightning's closure order")


def test_arena_zero_grad_in_place_needs_no_add_back(dev):
    """The reference's unchanged loop: `configure_optimizers()` (hulc2_amd.optim.Adam by default) + Lightning's closure order +
    `zero_grad(set_to_none=False)` (torch 1.12's default).  The optimizer zeroes the gradient arena with one fill and marks it; the node's
    backward finds the mark (and nothing written since) and neither copies nor adds: same bits as the plain order with the same optimizer.
    A torch operation on a gradient between zero_grad and backward moves the arena's version: that step goes through the add-back and keeps
    what was written."""
    def loop(lightning_order, poke=False):
        kn.reset_step_state(dev)
        m = _model(dev, 19)
        batch = _batch(dev, 19)
        opt = m.configure_optimizers()["optimizer"]
        from hulc2_amd.optim import Adam
        assert isinstance(opt, Adam)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        losses = []
        for i in range(6):
            if not lightning_order:
                opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = m.training_step(batch, i)
            if lightning_order:
                opt.zero_grad(set_to_none=False)
                if i >= 1:
                    assert all(float(p.grad.abs().max()) == 0.0 for p in list(m.parameters())[:3] if p.grad is not None)
                if poke and i == 3:
                    fg = m.__dict__["_hulc_step_node"].keeper.flat_g
                    lo, hi = fg.data_ptr(), fg.data_ptr() + 4 * fg.numel()
                    next(p for p in m.parameters() if p.grad is not None and lo <= p.grad.data_ptr() < hi).grad.add_(0.0)
            scaler.scale(loss).backward()
            scaler.step(opt)
            scaler.update()
            losses.append(float(loss))
        torch.cuda.synchronize()
        kn.check_faults(dev)
        assert opt.fused_launches == 6
        return m, losses, {n: p.detach().clone() for n, p in m.named_parameters()}
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None, HULC_TORCH_ADAM=None):
        m_a, l_a, p_a = loop(False)
        m_b, l_b, p_b = loop(True)
        m_c, l_c, p_c = loop(True, poke=True)
    nb, nc = m_b.__dict__["_hulc_step_node"], m_c.__dict__["_hulc_step_node"]
    assert nb.replays == 4 and nb.zeroed_steps == 5 and nb.accum_steps == 0, (nb.replays, nb.zeroed_steps, nb.accum_steps)
    assert nc.zeroed_steps == 4 and nc.accum_steps == 1, (nc.zeroed_steps, nc.accum_steps)
    assert l_a == l_b == l_c, (l_a, l_b, l_c)
    _same(p_b, p_a, "parameters after six steps, arena zero_grad in place")
    _same(p_c, p_a, "parameters after six steps, one of them through the add-back")


def test_logged_values_of_a_replayed_step_survive_the_next_replay(dev):
    """the graph path handed out the captured graph's static log tensors; a logger that keeps them saw step N+1's values"""
    kn.reset_step_state(dev)
    m = _model(dev, 23)
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-2)
    kept = []
    for i in range(6):
        opt.zero_grad(set_to_none=True)
        loss = m.training_step(_batch(dev, 23 + i), i)
        kept.append((m.logged["train/total_loss"], float(m.logged["train/total_loss"]), float(loss)))
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    assert m.__dict__["_hulc_step_node"].replays >= 3
    for t, at_the_time, loss in kept:
        assert float(t) == at_the_time == loss
    assert len({v for _, v, _ in kept}) == len(kept)


def test_drop_in_adam_reads_the_gradient_arena_and_takes_the_scaler_on_device(dev):
    """hulc2_amd.optim.Adam behind the node: gradients are the arena views (no copy), GradScaler hands its scale / found_inf over as device
    scalars (`_step_supports_amp_scaling`: no found_inf.item() in the loop).  Fed the SAME (scaled) gradients, torch.optim.Adam on clones —
    stepped with the gradients unscaled by hand — lands on the same parameters to fp32 rounding of two evaluation orders; a step with an inf
    gradient is skipped on the device and does not count."""
    from hulc2_amd.optim import Adam
    lr = 2e-4
    with _env(HULC_NO_STEP_NODE=None, HULC_NO_STEP_GRAPH=None):
        kn.reset_step_state(dev)
        m = _model(dev, 31)
        batch = _batch(dev, 31)
        params = [p for p in m.parameters() if p.requires_grad]
        opt = Adam(params, lr=lr)
        clones = [torch.nn.Parameter(p.detach().clone()) for p in params]
        ref = torch.optim.Adam(clones, lr=lr)
        scaler = torch.amp.GradScaler("cuda", init_scale=65536.0)
        for i in range(5):
            opt.zero_grad(set_to_none=True)
            with torch.autocast("cuda", dtype=torch.float16):
                loss = m.training_step(batch, i)
            scaler.scale(loss).backward()
            for c, p in zip(clones, params):
                c.grad = None if p.grad is None else p.grad.detach() / 65536.0
            scaler.step(opt)                                       # no unscale_ before it: the kernel divides by the scale itself
            scaler.update()
            ref.step()
        torch.cuda.synchronize()
        kn.check_faults(dev)
    assert opt.fused_launches == 5 and scaler.get_scale() == 65536.0
    tr = m.__dict__["_hulc_shadow_keeper"]
    assert m.__dict__["_hulc_step_node"].replays == 3
    assert opt._arena[1].data_ptr() == tr.flat_g.data_ptr(), "the optimizer reads the keeper's gradient arena in place"
    for (n, p), c in zip([(n, p) for n, p in m.named_parameters() if p.requires_grad], clones):
        assert float((p - c).abs().max()) <= 2e-6 * max(float(c.abs().max()), 1.0) + 1e-3 * lr, n
    assert float(opt.state_dict()["state"][0]["step"]) == 5.0
    # an inf in one gradient: the scaler's found_inf reaches the kernel, nothing moves, the step count stays, the scale halves
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.float16):
        loss = m.training_step(batch, 0)
    scaler.scale(loss).backward()
    next(iter(m.parameters())).grad.view(-1)[0] = float("inf")
    scaler.step(opt)
    scaler.update()
    torch.cuda.synchronize()
    for n, p in m.named_parameters():
        assert torch.equal(p, before[n]), n
    assert scaler.get_scale() == 32768.0
    assert float(opt.state_dict()["state"][0]["step"]) == 5.0


def test_drop_in_adam_late_first_gradient_and_fused_flag_take_torchs_path(dev):
    """a parameter whose FIRST gradient arrives after the others have stepped starts at step 1 in torch (its own bias correction):
    the drop-in takes torch's per-tensor path for it, the state_dict says step 1; `fused=True` is never the arena launch"""
    from hulc2_amd.optim import Adam
    with _env(HULC_NO_STEP_NODE=None):
        kn.reset_step_state(dev)
        m = _model(dev, 3, dropout_p=0.0)
        batch = syn.make_batch(3, 2, 8, device=dev)
        params = [p for p in m.parameters() if p.requires_grad]
        opt = Adam(params, lr=1e-3)
        clones = [torch.nn.Parameter(p.detach().clone()) for p in params]
        ref = torch.optim.Adam(clones, lr=1e-3)
        held = params[0]
        for i in range(3):
            opt.zero_grad(set_to_none=True)
            m.training_step(batch, i).backward()
            if i < 2:
                held.grad = None                                     # this parameter sees its first gradient at global step 3
            for c, p in zip(clones, params):
                c.grad = None if p.grad is None else p.grad.detach().clone()
            opt.step()
            ref.step()
        torch.cuda.synchronize()
        assert opt.fused_launches == 2
        sd, sr = opt.state_dict(), ref.state_dict()
        assert float(sd["state"][0]["step"]) == float(sr["state"][0]["step"]) == 1.0 and float(sd["state"][1]["step"]) == 3.0
        assert float((held - clones[0]).abs().max()) <= 2e-6 * float(clones[0].abs().max()) + 1e-6
        opt2 = Adam(params, lr=1e-3, fused=True)
        opt2.zero_grad(set_to_none=True)
        m.training_step(batch, 5).backward()
        opt2.step()
        assert opt2.fused_launches == 0


def test_step_node_under_torch_ddp(dev):
    """torch's own DistributedDataParallel around the node (Lightning's DDPStrategy, hulc2/training.py:72-75) on a one-rank RCCL group: the
    reducer's hooks fire on the arena views the node hands to autograd; gradients equal the undistributed loop's bit for bit, the
    cooperative kernels stay on and report no fault"""
    import socket
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    own = False
    if not dist.is_initialized():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
        own = True
    try:
        class _Step(torch.nn.Module):
            def __init__(self, m):
                super().__init__()
                self.module = m

            def forward(self, b, i):
                return self.module.training_step(b, i)

        def run(wrap):
            kn.reset_step_state(dev)
            m = _model(dev, 17)
            batch = _batch(dev, 17)
            opt = torch.optim.Adam(m.parameters(), lr=2e-4)
            f = DDP(_Step(m), device_ids=[dev.index or 0], static_graph=True) if wrap else _Step(m)
            out = []
            for i in range(4):
                opt.zero_grad(set_to_none=True)
                loss = f(batch, i)
                loss.backward()
                out.append({n: (None if p.grad is None else p.grad.detach().clone()) for n, p in m.named_parameters()})
                opt.step()
            torch.cuda.synchronize()
            kn.check_faults(dev)
            return m, out
        _, want = run(False)
        m, got = run(True)
        assert m.__dict__["_hulc_step_node"].replays == 2
        for i in range(4):
            _same(got[i], want[i], f"gradients of step {i} under DDP")
    finally:
        if own:
            dist.destroy_process_group()


def test_paired_conv1_on_small_frames_falls_back_to_the_per_input_loop(dev):
    """a two-modality conv stack on 36 x 36 frames — below what the conv1 band kernels take with a second frame tensor (x2) — used to
    raise from the backward; forward and backward now run the per-input launches and equal the single-tensor results"""
    from hulc2_amd import functional as HF
    kn.set_compute("bf16")
    g = torch.Generator(device="cpu").manual_seed(5)
    xs = [torch.rand(4, 3, 36, 36, generator=g).mul(2).sub(1).to(dev) for _ in range(2)]
    ws = [(torch.randn(32, 3, 8, 8, generator=g) * 0.05).to(dev).requires_grad_(), (torch.randn(32, generator=g) * 0.1).to(dev).requires_grad_(),
          (torch.randn(64, 32, 4, 4, generator=g) * 0.05).to(dev).requires_grad_(), (torch.randn(64, generator=g) * 0.1).to(dev).requires_grad_(),
          (torch.randn(64, 64, 3, 3, generator=g) * 0.05).to(dev).requires_grad_(), (torch.randn(64, generator=g) * 0.1).to(dev).requires_grad_()]
    a = HF.conv_stack(xs, ws)
    a.float().sum().backward()
    pair = [p.grad.clone() for p in ws]
    for p in ws:
        p.grad = None
    b = HF.conv_stack(torch.cat(xs), ws)
    b.float().sum().backward()
    assert torch.equal(a, b)
    for p, q in zip(ws, pair):
        assert torch.allclose(p.grad, q, rtol=2e-2, atol=2e-2 * float(q.abs().max())), (p.shape, float((p.grad - q).abs().max()))


def test_adam_kernel_takes_the_scalers_device_scalars(dev):
    """hulc_adam_step_amp (ABI 5): loss_scale as a device float gives the bits of the host-side 1 / scale; found_inf != 0 leaves parameters,
    moments and shadow untouched; hulc_step_count_advance_if counts only the steps that were taken"""
    n = 1000
    g = torch.Generator(device="cpu").manual_seed(1)
    p0 = torch.randn(n, generator=g).to(dev)
    gr = (torch.randn(n, generator=g) * 1024.0).to(dev)

    def run(**kw):
        p, m, v = p0.clone(), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        sh = torch.zeros(n, dtype=torch.bfloat16, device=dev)
        kn.adam_step(p, gr, m, v, sh, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, **kw)
        torch.cuda.synchronize()
        return p, m, v, sh
    a = run(grad_scale=1.0 / 1024.0)
    b = run(loss_scale_dev=torch.tensor([1024.0], device=dev), found_inf_dev=torch.zeros(1, device=dev))
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    c = run(loss_scale_dev=torch.tensor([1024.0], device=dev), found_inf_dev=torch.ones(1, device=dev))
    assert torch.equal(c[0], p0) and float(c[1].abs().max()) == 0.0 and float(c[3].float().abs().max()) == 0.0
    st = torch.tensor([0, 7], dtype=torch.int64, device=dev)
    kn.step_count_advance_if(st, torch.ones(1, device=dev))
    kn.step_count_advance_if(st, torch.zeros(1, device=dev))
    kn.step_count_advance_if(st, None)
    assert st.tolist() == [0, 9]
