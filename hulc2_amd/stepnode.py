"""One autograd node per training step — the reference's own trainer loop at the speed of the captured graphs.

reference: hulc2/training.py:72-82 (Lightning: `training_step` -> `loss.backward()` -> `optimizer.step()`, precision 16 =
`torch.autocast` + `GradScaler`, conf/trainer/play_trainer.yaml:3; DDPStrategy on top).  Whoever drives that loop sees `Hulc2.training_step`
return a scalar loss and calls `backward()` on (a multiple of) it.  Until round 4 the loss hung on ~160 autograd Functions whose Python
forward / backward bodies and ctypes launches made the loop host-bound: 7.3 ms per step against 3.4 ms for ArenaTrainer's replayed graphs.

Here the whole step is ONE node whose inputs are the parameters:

  forward   runs the step's forward (`Hulc2._training_step_impl`) under `torch.enable_grad()` and keeps the inner loss,
  backward  runs the inner backward with the incoming gradient as its root — under a GradScaler that is the loss scale, a device scalar the
            first backward kernel multiplies by (a power of two: exact through every kernel) — while the keeper's gradient sinks are live
            (trainer.ArenaTrainer(shadows_only=True, step_node=True)): the backward kernels write the weight gradients straight into one
            gradient arena (one grouped launch, the first writer of a slice overwrites it), and the node hands the arena views to autograd
            as the parameters' gradients.  AccumulateGrad takes them without a copy; DDP's reducer hooks, `GradScaler.unscale_`,
            gradient clipping and any torch optimizer see ordinary `.grad` tensors.

From the third step of an unchanged configuration on (same batch layout, same arithmetic mode) both halves are two hipGraphs captured on a
side stream — forward (RNG word advance, forward, losses) and backward (root = a static device scalar) — and the node's forward / backward
are one `replay()` each: ~12 000 Python-level calls per step become ~10.  The batch of the capturing call is the graphs' input buffer;
a later batch at other addresses is copied into it (one multi-tensor copy), one at the same addresses (a resident batch, an HBM episode
store's fixed windows) costs nothing.  Anything the graphs cannot express falls back to the eager node or to the plain call: gradients
being accumulated over several calls (a live `.grad`), a batch with host data (sentences for the on-device tokenizer), an active capture
or per-launch timing, a capture that fails (kept as `disabled`).  HULC_NO_STEP_NODE=1 keeps round 4's loop, HULC_NO_STEP_GRAPH=1 the eager node.
"""
import os
import warnings
from typing import List, Optional, Tuple

import torch

from . import gradsink
from . import kernels as kn
from . import shadow


def _rebuild(obj, it):
    """the same nest of dicts / lists with its tensors taken, in _leaves' order, from the iterator `it`"""
    if torch.is_tensor(obj):
        return next(it)
    if isinstance(obj, dict):
        return {k: _rebuild(v, it) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_rebuild(v, it) for v in obj)
    return obj


def _leaves(obj, prefix: str, out: list) -> bool:
    """flatten a batch into (path, tensor) pairs in a fixed order; False when it holds anything a graph cannot take as an input"""
    if torch.is_tensor(obj):
        out.append((prefix, obj))
        return True
    if isinstance(obj, dict):
        return all(_leaves(v, f"{prefix}/{k}", out) for k, v in obj.items())
    if isinstance(obj, (list, tuple)):
        return all(_leaves(v, f"{prefix}/{i}", out) for i, v in enumerate(obj))
    return obj is None or isinstance(obj, (bool, int))


class _EagerStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, node, batch, batch_idx, *params):
        loss, logs = node._inner_forward(batch, batch_idx)
        ctx.node, ctx.inner = node, loss
        node._logs = logs
        return loss.detach()

    @staticmethod
    def backward(ctx, g):
        inner, ctx.inner = ctx.inner, None
        if inner is None:
            raise RuntimeError("hulc2_amd step node: backward through one training step twice (the inner graph is freed by its first backward)")
        node = ctx.node
        held = node._take_live_grads(node.views)
        outs = node._inner_backward(inner, g)
        node._give_back_live_grads(held, outs)
        return (None, None, None, *outs)


class _GraphStepFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, node, *params):
        node.graph_fwd.replay()
        ctx.node = node
        ctx.cap = (node.graph_bwd, node.static_g, node.static_outs)        # (the node may have switched to another batch layout's graphs by the time of backward)
        return node.static_loss.clone()                   # (the caller may keep the loss beyond the next replay)

    @staticmethod
    def backward(ctx, g):
        graph_bwd, static_g, static_outs = ctx.cap
        held = ctx.node._take_live_grads(static_outs)
        static_g.copy_(g.reshape(static_g.shape))
        graph_bwd.replay()
        ctx.node._give_back_live_grads(held, static_outs)
        return (None, *[None if o is None else o.detach() for o in static_outs])


class StepNode:
    """the step node of one Hulc2 module under its shadows-only keeper (`Hulc2._step_node`)"""

    STASHED = 3                       # captured batch layouts kept beside the current one (least recently used goes first)
    EAGER_STEPS = 2                   # eager-node steps of a configuration before it is captured (the second one runs in sink-overwrite mode)
    _STATE = ("eager_seen", "graph_fwd", "graph_bwd", "static_loss", "static_g", "static_outs", "static_leaves", "static_logs", "slot_idx", "slots_ok",
              "slot_table", "slot_ptrs", "_ring", "_ring_pos")          # what belongs to ONE captured batch layout

    def __init__(self, model, keeper):
        import weakref
        self._model = weakref.ref(model)
        self.keeper = keeper
        self.dev = keeper.dev
        self.views = [keeper.flat_g[off:off + p.numel()].view(p.shape) for p, off in zip(keeper.params, keeper.offsets)]
        # members of a fused group (the decoder's four heads as one matrix view) are written through the GROUP's sink
        self.group_of = [None] * len(keeper.params)
        for pv, gv, off, shape in keeper.fused:
            for i, (p, o) in enumerate(zip(keeper.params, keeper.offsets)):
                if off <= o < off + gv.numel():
                    self.group_of[i] = pv
        self.sig = None
        self._stash = {}                       # signature -> the captured state of a layout that is not the current one (at most one)
        self.eager_seen = 0
        self.graph_fwd = self.graph_bwd = None
        self.static_loss = self.static_g = None
        self.static_outs: List[Optional[torch.Tensor]] = []
        self.static_leaves: List[torch.Tensor] = []
        self.static_logs = []
        self.disabled: Optional[str] = "HULC_NO_STEP_GRAPH" if os.environ.get("HULC_NO_STEP_GRAPH") else None
        self.replays = self.eager_steps = self.captures = self.input_copies = self.accum_steps = self.zeroed_steps = self.evictions = 0      # (tests / bench read these)
        self._seen = {}                        # signature -> eager-node steps taken (a loop that alternates layouts still reaches its captures)
        # frame slots (hulc_conv_desc.x_slot): the big frame tensors are read by conv1's captured launches through device pointer slots
        self.slot_idx: List[int] = []          # leaves read through slots
        self.slots_ok = False                  # ... verified by the self-check of _capture
        self.slot_table = None                 # int64 [leaves]: the device slots
        self.slot_ptrs: List[int] = []         # what the slots hold now
        self.slot_updates = 0
        self._ring, self._ring_pos = [], 0     # pinned staging buffers of the slot table + the events that guard their reuse
        self._live_inputs = None               # the caller's current batch tensors: the backward graph reads the frames again
        self._logs = []

    def __deepcopy__(self, memo):
        return None                    # (copy.deepcopy(model): the copy builds its own keeper and node on its first training step)

    # ---- the two halves of a step ------------------------------------------------------------------------------------------------
    def _inner_forward(self, batch, batch_idx) -> Tuple[torch.Tensor, list]:
        dev = self.dev
        kn.advance_step_state(dev, rng=True, step=False)      # fresh dropout masks / plan sample (inside a capture: part of the graph)
        kn._rng_fresh.pop(dev, None)
        shadow.bump_epoch()                                   # every lazily cached weight operand is re-made inside this step (and inside a capture)
        kn.wgrad_reset(dev)
        with torch.enable_grad(), gradsink.active(self.keeper):
            loss, logs = self._model()._training_step_impl(batch, batch_idx)
        return loss, [(n, v.detach() if torch.is_tensor(v) else v, kw) for n, v, kw in logs]

    def _inner_backward(self, inner: torch.Tensor, g: torch.Tensor, leaves=None) -> list:
        tr = self.keeper
        tr._zero_arena()
        with gradsink.active(tr):
            res = torch.autograd.grad(inner, tr.params if leaves is None else leaves, grad_outputs=g.reshape(inner.shape), allow_unused=True)
            kn.wgrad_flush(self.dev)                          # (autograd's end-of-pass callback already issued it: no-op unless the pass was cut short)
        # the gripper camera's encoder runs forward AND backward on a side stream (concat_encoders.forward_multi); its weight-gradient kernels
        # write the gradient arena from there.  autograd's end-of-pass join covers only streams on which a leaf received a gradient — here the
        # leaves' AccumulateGrad nodes belong to the caller's stream and the Functions return None — so the join is made explicitly
        from .models.perceptual_encoders.concat_encoders import _side_streams
        side = _side_streams.get(self.dev)
        if side is not None:
            kn.join_stream(self.dev, side)
        tr._settle_sinks()
        outs = []
        for i, (p, v, r) in enumerate(zip(tr.params, self.views, res)):
            pv = self.group_of[i]
            sunk = gradsink.written(p) or (pv is not None and gradsink.written(pv))
            if r is None:
                outs.append(v.detach() if sunk else None)     # (a parameter nobody wrote has no gradient, as in the plain loop: .grad stays None)
            else:
                outs.append(v + r if sunk else r)             # the rare parameter autograd itself produced a gradient for
        return outs

    # ---- dispatch -----------------------------------------------------------------------------------------------------------------
    def _signature(self, leaves) -> tuple:
        m = self._model()
        return (tuple((path, tuple(t.shape), t.dtype, t.device) for path, t in leaves), kn.base_mode(), kn.get_compute(),
                os.environ.get("HULC_FP32_SITES"), kn.concurrent_streams(), kn.fork_branches(), float(m.kl_beta), float(m.kl_balancing_mix),
                float(m.clip_auxiliary_loss_beta), bool(m.use_clip_auxiliary_loss))

    def usable(self) -> bool:
        """the keeper's parameter list is still the model's trainable set.  (A live `.grad` is NOT a reason to leave the node — round 6, ADVICE
        r05: Lightning's closure runs training_step -> zero_grad -> backward, so from the second step on the previous step's gradients are
        still attached when training_step runs; what they mean is decided when backward runs: _take_live_grads.)"""
        return all(p.requires_grad for p in self.keeper.params)

    # ---- gradients that are still attached when backward runs -----------------------------------------------------------------------------
    def _take_live_grads(self, dests):
        """Called in front of the inner backward.  `dests[i]` is the tensor the backward is about to OVERWRITE and hand to autograd as parameter
        i's gradient (an arena view, or a buffer of the captured graph).  A parameter whose `.grad` is that very memory — the previous step's
        gradient left attached: `optimizer.zero_grad(set_to_none=False)` (the default of the torch 1.12 the reference pins) zeroed it in
        place, or nobody zeroed it because gradients are being accumulated over several calls — would see `grad += grad` from AccumulateGrad.
        So: keep a copy of what is there (ONE copy of the arena), detach those `.grad`s, let the backward overwrite, add the copy back in
        (_give_back_live_grads, one launch): autograd then installs the views again and `.grad` = old + new, bit for bit what the plain loop's
        accumulation gives (zeros + new = new).  A `.grad` that lives elsewhere (DDP's bucket views, a user's tensor) is left alone:
        AccumulateGrad adds into it as always.  A copy and an add of the arena per step on this path; `set_to_none=True` costs nothing."""
        tr = self.keeper
        lo = tr.flat_g.data_ptr()
        hi = lo + tr.flat_g.numel() * 4
        # hulc2_amd.optim.Adam.zero_grad(set_to_none=False) filled the whole arena with zeros and no torch operation has written it since (the
        # views share the arena's version counter; this node's own kernels are the only other writers, and every backward of it forgets the
        # mark): nothing to keep, nothing to add back
        known_zero = getattr(tr, "grads_zeroed_at", None) is not None and tr.grads_zeroed_at == tr.flat_g._version
        tr.grads_zeroed_at = None
        taken, others = [], []
        for i, (p, d) in enumerate(zip(tr.params, dests)):
            g = p.grad
            if g is None:
                continue
            if lo <= g.data_ptr() < hi:                          # an arena view: every backward of this node overwrites the arena
                taken.append((i, g))
            elif d is not None and g.data_ptr() == d.data_ptr():  # a buffer of the captured backward graph (a gradient autograd itself produced)
                others.append((i, g, g.clone()))
            else:
                continue
            p.grad = None
        if not taken and not others:
            return None
        if known_zero and not others:
            self.zeroed_steps += 1
            return (None, taken, others)
        self.accum_steps += 1
        return (tr.flat_g.clone() if taken else None, taken, others)

    def _give_back_live_grads(self, held, outs) -> None:
        if held is None:
            return
        prev, taken, others = held
        tr = self.keeper
        if prev is None:
            for i, g in taken:                                    # the arena was known to be zero: a parameter without a gradient in this pass gets its
                if outs[i] is None:                               # (zero) gradient back, everything else is zero + new = new
                    g.zero_()
                    tr.params[i].grad = g
        else:
            # the old values go back into the slices of the parameters they were taken from — and nowhere else: the slice of a parameter whose
            # `.grad` lives outside the arena (AccumulateGrad cloned instead of adopting the view) is never zeroed by zero_grad, what a copy of
            # the arena holds there is the previous step's gradient
            taken_idx = {i for i, _ in taken}
            in_arena, dst, src = [], [], []
            for i, g in taken:
                o = outs[i]
                off = tr.offsets[i]
                if o is None:                                     # no gradient for this parameter in this pass: what was attached stays attached
                    g.copy_(prev[off:off + g.numel()].view(g.shape))
                    tr.params[i].grad = g
                elif o.data_ptr() == self.views[i].data_ptr():
                    in_arena.append(i)                            # the arena view itself
                else:                                             # a sum autograd made next to the arena (view + its own term)
                    dst.append(o)
                    src.append(prev[off:off + o.numel()].view(o.shape))
            # the arena views: ONE add over the whole arena once the copy is zero wherever it must not land (the slices of parameters that were
            # not taken — a handful: per-tensor adds over ~100 views cost 0.6 ms of launches, this costs the two passes)
            # (must be zero in the copy: slices the backward wrote for a parameter that was NOT taken, and the taken ones handled above;
            #  a parameter without a gradient this step and without an attached one has an all-zero slice on both sides)
            skip = [i for i in range(len(tr.params))
                    if (i in taken_idx and (outs[i] is None or outs[i].data_ptr() != self.views[i].data_ptr()))
                    or (i not in taken_idx and outs[i] is not None)]
            if len(skip) <= 16:
                for i in skip:
                    off = tr.offsets[i]
                    if not any(outs[i] is d_ for d_ in dst):      # (its copy is still needed by the per-tensor add below)
                        prev[off:off + tr.params[i].numel()].zero_()
                if dst:
                    torch._foreach_add_(dst, src)
                    for i in skip:
                        off = tr.offsets[i]
                        prev[off:off + tr.params[i].numel()].zero_()
                tr.flat_g.add_(prev)
            else:
                dst += [self.views[i] for i in in_arena]
                src += [prev[tr.offsets[i]:tr.offsets[i] + self.views[i].numel()].view(self.views[i].shape) for i in in_arena]
                if dst:
                    torch._foreach_add_(dst, src)
        for i, g, old in others:
            if outs[i] is None:
                g.copy_(old)
                tr.params[i].grad = g
            else:
                outs[i].add_(old)

    def __call__(self, batch, batch_idx) -> Tuple[torch.Tensor, list]:
        tr = self.keeper
        leaves: list = []
        graphable = (self.disabled is None and _leaves(batch, "", leaves) and kn._timing is None
                     and not torch.cuda.is_current_stream_capturing())
        if graphable:
            sig = self._signature(leaves)
            if sig != self.sig:
                # another batch layout (the last, smaller batch of an epoch; validation-sized batches; another KL weight): its graphs are kept
                # next to the current ones — the two most recent layouts stay captured
                st = self._stash.pop(sig, None)
                if self.sig is not None:
                    self._seen[self.sig] = self.eager_seen       # (kept per layout: alternating layouts each reach EAGER_STEPS and are captured)
                    if self.graph_fwd is not None:
                        self._stash[self.sig] = {k: getattr(self, k) for k in self._STATE}
                        while len(self._stash) > self.STASHED:
                            old_sig = next(iter(self._stash))
                            self._stash.pop(old_sig)
                            self._seen.pop(old_sig, None)
                            self.evictions += 1
                            if self.evictions == 8:
                                warnings.warn(f"hulc2_amd step node: more than {self.STASHED + 1} batch layouts / loss weights rotate through training_step; "
                                              "their graphs are being re-captured (StepNode.STASHED raises the number kept)")
                self._drop_graphs()
                self.sig, self.eager_seen = sig, self._seen.get(sig, 0)
                while len(self._seen) > 16:
                    self._seen.pop(next(iter(self._seen)))
                if st is not None:
                    for k, v in st.items():
                        setattr(self, k, v)
            if self.graph_fwd is None and self.eager_seen >= self.EAGER_STEPS:
                try:
                    self._capture(batch, batch_idx, [t for _, t in leaves])
                except Exception as e:                        # noqa: BLE001 - the step must still be taken
                    self._drop_graphs()
                    self.disabled = f"capture failed: {type(e).__name__}: {e}"
                    warnings.warn(f"hulc2_amd step node: {self.disabled}; the step stays on the eager node")
            if self.graph_fwd is not None:
                src, dst = [], []
                moved = False
                slot = set(self.slot_idx) if self.slots_ok else ()
                for i, ((_, t), st) in enumerate(zip(leaves, self.static_leaves)):
                    if i in slot and t.is_contiguous() and t.data_ptr() % 16 == 0:
                        if self.slot_ptrs[i] != t.data_ptr():        # a frame tensor at a new address: its slot follows it, nothing is copied
                            self.slot_ptrs[i] = t.data_ptr()
                            moved = True
                    elif t.data_ptr() != st.data_ptr():
                        src.append(t)
                        dst.append(st)
                        if i in slot and self.slot_ptrs[i] != st.data_ptr():
                            self.slot_ptrs[i] = st.data_ptr()
                            moved = True
                if src:
                    torch._foreach_copy_(dst, src)
                    self.input_copies += 1
                if moved:
                    self._write_slots()
                self._live_inputs = [t for _, t in leaves]
                self.replays += 1
                loss = _GraphStepFn.apply(self, *tr.params)
                return loss, self._logs_of_this_call()
        self.eager_seen += 1
        self.eager_steps += 1
        loss = _EagerStepFn.apply(self, batch, batch_idx, *tr.params)
        logs, self._logs = self._logs, []
        return loss, logs

    def _logs_of_this_call(self) -> list:
        """the captured graph's logged values are its static device tensors: the next replay overwrites them, and a logger keeps what it is
        given (`self.log(..., on_step=True)`, `_MiniLightningModule.logged`) — ADVICE r05.  One stack launch copies the tensor-valued entries
        (values and device-valued `batch_size` keywords) into a buffer of this call; the entries handed out are views of it."""
        flat = []
        for _, v, kw in self.static_logs:
            if torch.is_tensor(v):
                flat.append(v)
            flat += [x for x in kw.values() if torch.is_tensor(x)]
        fresh = [None] * len(flat)
        by_dtype = {}
        for i, t in enumerate(flat):
            if t.dim() == 0:
                by_dtype.setdefault(t.dtype, []).append(i)
            else:
                fresh[i] = t.clone()
        for idx in by_dtype.values():                             # (losses: fp32 scalars; the contrastive head's row count: one more dtype)
            for i, c in zip(idx, torch.stack([flat[i] for i in idx]).unbind(0)):
                fresh[i] = c
        it = iter(fresh)
        return [(n, next(it) if torch.is_tensor(v) else v, {k: (next(it) if torch.is_tensor(x) else x) for k, x in kw.items()})
                for n, v, kw in self.static_logs]

    def _drop_graphs(self) -> None:
        self.graph_fwd = self.graph_bwd = None
        self.static_loss = None
        self.static_outs, self.static_leaves, self.static_logs = [], [], []
        self.slot_idx, self.slots_ok, self.slot_table, self.slot_ptrs, self._live_inputs = [], False, None, [], None
        self._ring, self._ring_pos = [], 0

    def _write_slots(self) -> None:
        """slot_ptrs -> the device table, stream-ordered in front of the next replay: through a small ring of pinned buffers (an asynchronous
        copy from pageable memory would synchronise the stream; a buffer is reused only once its copy has run)"""
        if not self._ring:
            self._ring = [(torch.empty(len(self.slot_ptrs), dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
        buf, ev = self._ring[self._ring_pos % len(self._ring)]
        if buf.numel() != len(self.slot_ptrs):
            self._ring = [(torch.empty(len(self.slot_ptrs), dtype=torch.int64).pin_memory(), torch.cuda.Event()) for _ in range(8)]
            buf, ev = self._ring[self._ring_pos % len(self._ring)]
        elif self._ring_pos >= len(self._ring):
            ev.synchronize()
        self._ring_pos += 1
        buf.copy_(torch.tensor(self.slot_ptrs, dtype=torch.int64))
        self.slot_table.copy_(buf, non_blocking=True)
        ev.record(torch.cuda.current_stream(self.dev))
        self.slot_updates += 1

    def _param_slots(self):
        """(module, attribute name, index into keeper.params) of every place a module holds one of the keeper's parameters"""
        index = {id(p): i for i, p in enumerate(self.keeper.params)}
        slots = []
        for mod in self._model().modules():
            for name, p in mod._parameters.items():
                if p is not None and id(p) in index:
                    slots.append((mod, name, index[id(p)]))
        return slots

    def _capture(self, batch, batch_idx, leaves) -> None:
        """forward and backward of this batch layout as two hipGraphs on a side stream (shared memory pool).  A warm-up pass on that stream
        comes first: per-stream workspaces and lazily made buffers exist before the capture; the RNG word it drew from is put back, so the
        loop's sequence of draws is the eager loop's.
        Inside the capture the modules hold detached leaf ALIASES of the parameters (same memory): the real parameters' AccumulateGrad nodes
        are alive on the caller's stream (the previous step's loss, the outer node, DDP's reducer all reference them) and autograd would
        synchronise the capturing stream with that stream — which pulls it into the capture (hipErrorStreamCaptureUnjoined).  The aliases'
        accumulators are born on the capturing stream and die with the captured graph."""
        dev = self.dev
        torch.cuda.synchronize(dev)
        cur = torch.cuda.current_stream(dev)
        side = kn.capture_stream(dev)
        words = kn.step_state(dev).clone()
        one = torch.ones((), dtype=torch.float32, device=dev)
        # frame slots: which of the batch's tensors would conv1's band launches read through device pointer slots?  (recorded by the warm-up)
        want_slots = kn.base_mode() == "bf16" and not os.environ.get("HULC_NO_FRAME_SLOTS")
        cand = {t.data_ptr(): i for i, t in enumerate(leaves)
                if want_slots and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.dim() >= 4 and t.numel() >= (1 << 18)
                and t.data_ptr() % 16 == 0}
        kn._frame_slots, kn._frame_slots_probe = (cand or None), True
        kn._frame_slots_used = set()
        side.wait_stream(cur)
        try:
            with torch.cuda.stream(side):
                loss, _ = self._inner_forward(batch, batch_idx)
                self._inner_backward(loss, one)
                del loss
                kn.step_state(dev).copy_(words)
        finally:
            kn._frame_slots, kn._frame_slots_probe = None, False
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        slot_idx = sorted(cand[p] for p in kn._frame_slots_used if p in cand)
        # the graphs' input buffers: the caller's own tensors (a resident batch then costs nothing), except the slot-read frame tensors — private
        # copies, so that the self-check below may poison them and a batch that cannot be read in place (not contiguous) has a place to go
        statics = [t.clone() if i in set(slot_idx) else t for i, t in enumerate(leaves)]
        cap_batch = _rebuild(batch, iter(statics)) if slot_idx else batch
        table = torch.tensor([t.data_ptr() for t in statics], dtype=torch.int64, device=dev) if slot_idx else None
        self.static_g = torch.ones((), dtype=torch.float32, device=dev)
        g_f, g_b = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        # (a process group's watchdog thread may query events while this thread captures: only this thread's calls are policed then)
        mode = {"capture_error_mode": "thread_local"} if (torch.distributed.is_available() and torch.distributed.is_initialized()) else {}
        slots = self._param_slots()
        aliases = [p.detach().requires_grad_(True) for p in self.keeper.params]
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        try:
            for mod, name, i in slots:
                mod._parameters[name] = aliases[i]
            gradsink.set_aliases(aliases, self.keeper.params)
            if slot_idx:
                kn._frame_slots = {statics[i].data_ptr(): table.data_ptr() + 8 * i for i in slot_idx}
                kn._frame_slots_used = set()
            with kn.no_gc():
                with torch.cuda.graph(g_f, stream=side, **mode):
                    loss, logs = self._inner_forward(cap_batch, batch_idx)
                with torch.cuda.graph(g_b, pool=g_f.pool(), stream=side, **mode):
                    outs = self._inner_backward(loss, self.static_g, leaves=aliases)
        finally:
            kn._frame_slots = None
            for mod, name, i in slots:
                mod._parameters[name] = self.keeper.params[i]
            gradsink.clear_aliases()
        self.static_loss, self.static_logs, self.static_outs = loss.detach(), logs, outs
        del loss, aliases
        self.static_leaves = statics
        self.graph_fwd, self.graph_bwd = g_f, g_b
        self.slot_idx, self.slot_table, self.slot_ptrs = slot_idx, table, [t.data_ptr() for t in statics]
        self.slots_ok = bool(slot_idx) and self._slots_self_check()
        self.captures += 1
        torch.cuda.synchronize(dev)

    def _slots_self_check(self) -> bool:
        """Does EVERY captured reader of a slot-driven frame tensor go through its slot?  Replay both graphs twice from the same RNG word — once
        as captured, once with the slots pointed at fresh copies of the frames and the graphs' own buffers filled with NaN: loss and the
        checksums of the gradient arena must agree bit for bit.  If they do not (some kernel of the capture read a frame tensor at its baked-in
        address: another arithmetic mode's forward, a model that hands the frames to more than conv1), the slots stay on the private buffers
        and every batch is copied into them, as without slots."""
        dev, tr = self.dev, self.keeper
        words = kn.step_state(dev).clone()

        def probe():
            kn.step_state(dev).copy_(words)
            self.static_g.fill_(1.0)
            self.graph_fwd.replay()
            self.graph_bwd.replay()
            g = tr.flat_g.double()
            return self.static_loss.clone(), g.sum(), (g * g).sum()
        ref = probe()
        alts = {i: self.static_leaves[i].clone() for i in self.slot_idx}
        self.slot_table.copy_(torch.tensor([alts[i].data_ptr() if i in alts else p for i, p in enumerate(self.slot_ptrs)], dtype=torch.int64))
        for i in self.slot_idx:
            self.static_leaves[i].fill_(float("nan"))
        got = probe()
        for i in self.slot_idx:
            self.static_leaves[i].copy_(alts[i])
        self.slot_table.copy_(torch.tensor(self.slot_ptrs, dtype=torch.int64))
        kn.step_state(dev).copy_(words)
        torch.cuda.synchronize(dev)
        ok = all(bool(torch.equal(a, b)) for a, b in zip(ref, got))
        if not ok:
            warnings.warn("hulc2_amd step node: a captured kernel reads a frame tensor at its baked-in address; frame slots are off, "
                          "batches at new addresses are copied into the graphs' input buffers")
        return ok
