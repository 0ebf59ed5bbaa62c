"""Native training loop for the hot path: flat HBM arenas, fused Adam, RCCL gradient all-reduce.

The reference trains through pytorch-lightning (hulc2/training.py:64-82: Trainer.fit -> training_step ->
backward -> DDP bucketed all-reduce -> Adam step).  Lightning is a third-party layer outside the hot-path scope
(SURVEY.md §2 L3); this file is the MI355X-native equivalent of exactly that inner loop:

  * all trainable parameters live in ONE contiguous fp32 arena (params / grads / exp_avg / exp_avg_sq share
    offsets) plus a bf16 shadow arena the MFMA kernels read — 47 M params = 188 MB x4 + 94 MB of 288 GB HBM
  * the optimizer is one HBM-streaming kernel launch over the arena (hulc_adam_step) that also refreshes the
    bf16 shadow — torch.optim.Adam semantics (hulc2.py:185-198, lr 2e-4)
  * data parallelism: gradient buckets are contiguous slices of the grad arena, all-reduced in place (no
    flatten/unflatten copies) on a side HIP stream as soon as autograd has produced every gradient of the
    bucket; the 1/world scale is folded into the Adam kernel.  One process per GPU, torch.distributed
    ("nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
from typing import Dict, List, Optional

import contextlib
import os
import weakref

import torch
import torch.distributed as dist

from . import gradsink
from . import kernels as kn
from . import shadow

# parameter arenas by the address of their storage (weak: an arena lives as long as its trainer / the model's keeper)
_ARENAS: "weakref.WeakValueDictionary[int, ArenaTrainer]" = weakref.WeakValueDictionary()


def arena_of(params) -> "Optional[ArenaTrainer]":
    """the ArenaTrainer whose fp32 arena holds exactly these parameters (same objects, every one a view of the arena), else None"""
    params = list(params)
    if not params or params[0].device.type != "cuda":
        return None
    tr = _ARENAS.get(params[0].untyped_storage().data_ptr())
    if tr is None or len(params) != len(tr.params):
        return None
    mine = {id(p) for p in tr.params}
    base = tr.flat_p.data_ptr()
    for p in params:
        if id(p) not in mine:
            return None
    for p, off in zip(tr.params, tr.offsets):
        if p.data_ptr() != base + off * 4 or p.dtype != torch.float32:
            return None
    return tr


class GradComm:
    """Sums slices of the fp32 gradient arena over the ranks, in place (replaces the NCCL ring inside Lightning's DDPStrategy,
    hulc2/training.py:72-75).

      algo     "ring"   one dist.all_reduce per slice (RCCL picks the algorithm; on xGMI a ring is bound by ONE of the 7 links)
               "direct" all_to_all_single -> local sum of the W contributions of this rank's chunk in rank order (hulc_sum_chunks, fp32
                        accumulation) -> all_gather_into_tensor: every rank talks to every peer at once, so all 7 links carry 1/W of the
                        payload each way; the result is bit-identical on all ranks (one rank reduces a chunk, everyone receives it)
      payload  "fp32"   the arena slice as it is
               "bf16"   rounded to bf16 for the wire (half the bytes), widened back into the arena afterwards; changes the arithmetic
                        (gradients pick up one bf16 rounding, the sum of "direct" a second one), hence a switch
    Defaults (HULC_ALLREDUCE / HULC_GRAD_PAYLOAD override them):
      payload  fp32 at every world size — the reference's DDP all-reduce is fp32 (hulc2/training.py:72-75), and the arithmetic of the
               gradients must not depend on how many ranks a job has.  bf16 is opt-in (HULC_GRAD_PAYLOAD=bf16 / grad_payload="bf16"): half the
               bytes for one bf16 rounding of the summands; with bf16 and algo "auto" the exchange is always "direct" (hulc_sum_chunks
               accumulates in fp32: one rounding) — "ring" with a bf16 buffer lets RCCL add in bf16 at every hop (W - 1 roundings) and runs
               only when asked for by name.
      algo     "auto" (fp32): on GPUs with world >= 2 both algorithms are TIMED once on the real fabric when the trainer is built (three
               reductions of a 32 MB slice each, max over ranks; the ranks agree on every step of the probe before they enter a collective)
               and the faster one is kept — whether RCCL's own multi-ring all-reduce or the direct exchange wins on an 8-GPU xGMI mesh is a
               property of the node.  Ring and direct sum in different orders, so the choice is part of a run's numerics: it is printed
               (bench.py's line, `describe()`), stored in ArenaTrainer.state_dict()["comm"], re-applied by load_state_dict, and pinned with
               HULC_ALLREDUCE=ring|direct.  On CPU (gloo) auto = ring."""

    def __init__(self, flat_grad: torch.Tensor, group=None, algo: Optional[str] = None, payload: Optional[str] = None, force: bool = False):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (force and dist.is_initialized())
        self.algo = algo or os.environ.get("HULC_ALLREDUCE", "auto")
        self.payload = payload or os.environ.get("HULC_GRAD_PAYLOAD") or self.default_payload(self.world)
        if self.algo not in ("ring", "direct", "auto") or self.payload not in ("fp32", "bf16"):
            raise ValueError(f"gradient all-reduce: algo {self.algo!r} (ring | direct | auto), payload {self.payload!r} (fp32 | bf16)")
        self.flat_grad = flat_grad
        self.on_gpu = flat_grad.is_cuda
        self._bufs = {}
        self.probe_ms: Dict[str, float] = {}
        self.chosen_by = "pinned"
        if self.algo == "auto":
            fabric = self.active and self.on_gpu and self.world > 1 and dist.get_backend(group) == "nccl"     # (gloo: host staging, nothing to tune)
            if self.payload == "bf16":
                self.algo, self.chosen_by = "direct", "rule (bf16 payload: fp32-accumulating exchange)"
            elif fabric:
                self.algo, self.chosen_by = self._probe(), "probe"
            else:
                self.algo, self.chosen_by = "ring", "rule (no fabric to probe)"

    @staticmethod
    def default_payload(world: int) -> str:
        return "fp32"

    def pin(self, algo: str) -> None:
        """re-apply a recorded choice (ArenaTrainer.load_state_dict): a resumed run keeps the summation order it started with"""
        if algo not in ("ring", "direct"):
            raise ValueError(f"gradient all-reduce: cannot pin {algo!r}")
        if algo != self.algo:
            self.algo, self.chosen_by = algo, "checkpoint"
            self._bufs = {}

    def _agree(self, ok: bool) -> bool:
        """True when EVERY rank says ok — a rank that failed locally (allocation) must not leave its peers alone inside a collective"""
        t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=self.flat_grad.device)
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.group)
        return bool(int(t.item()))

    PROBE_ELEMS = 8 << 20          # 32 MB of fp32 gradients: enough to be bandwidth-bound on xGMI, small next to the 288 GB

    def _probe(self) -> str:
        """time both algorithms on a bounded scratch slice (the gradients are not touched) and keep the faster; every rank sees the same
        max-over-ranks timings, so every rank makes the same choice.  Local failures (an allocation) are agreed on with an all_reduce(MIN)
        BEFORE the ranks enter the algorithm's collectives; the staging buffers of the probe are dropped afterwards."""
        keep = self.flat_grad
        n = min(keep.numel(), self.PROBE_ELEMS)
        scratch = None
        try:
            scratch = torch.zeros(n, dtype=keep.dtype, device=keep.device)
        except Exception as e:                                      # noqa: BLE001
            self.probe_error = f"scratch: {type(e).__name__}: {e}"
        if not self._agree(scratch is not None):
            return "ring"
        self.flat_grad = scratch
        try:
            for algo in ("ring", "direct"):
                self.algo = algo
                ok = True
                try:
                    self.reserve()                                  # local: staging buffers of this algorithm
                except Exception as e:                              # noqa: BLE001
                    ok, self.probe_error = False, f"{algo}: {type(e).__name__}: {e}"
                if not self._agree(ok):
                    self.probe_ms[algo] = float("inf")
                    continue
                self.reduce(0, n)                                   # warm-up: communicators (an error HERE is every rank's error: it propagates)
                if self.on_gpu:
                    torch.cuda.synchronize()
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(3):
                        self.reduce(0, n)
                    e1.record()
                    torch.cuda.synchronize()
                    ms = e0.elapsed_time(e1) / 3
                else:                                               # (host process groups: the same probe on the wall clock — what the CPU tests run)
                    import time
                    t0 = time.perf_counter()
                    for _ in range(3):
                        self.reduce(0, n)
                    ms = (time.perf_counter() - t0) / 3 * 1e3
                t = torch.tensor([ms], dtype=torch.float32, device=keep.device)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)      # (every rank sees the slowest rank's time)
                self.probe_ms[algo] = round(float(t.item()), 4)
        finally:
            self.flat_grad = keep
            self._bufs = {}                                         # (reserve() re-creates what the chosen algorithm needs, at arena size)
        best = min(self.probe_ms, key=self.probe_ms.get)
        return best if self.probe_ms[best] != float("inf") else "ring"

    def describe(self) -> Dict:
        """what bench.py prints about the gradient exchange: algorithm, payload, bytes, the probe's timings"""
        n = self.flat_grad.numel()
        payload_bytes = n * (2 if self.payload == "bf16" else 4)
        sent = int(2 * (self.world - 1) / max(self.world, 1) * payload_bytes)
        return {"algo": self.algo, "chosen_by": self.chosen_by, "payload": self.payload, "gradient_bytes": payload_bytes, "bytes_sent_per_rank_per_step": sent,
                "probe_ms": {k: (v if v != float("inf") else None) for k, v in self.probe_ms.items()},
                **({"probe_error": self.probe_error} if getattr(self, "probe_error", None) else {})}

    def _buf(self, name: str, n: int, dtype) -> torch.Tensor:
        t = self._bufs.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = self._bufs[name] = torch.zeros(max(n, self.flat_grad.numel() + 8 * self.world), dtype=dtype, device=self.flat_grad.device)
        return t[:n]

    def reserve(self) -> None:
        """allocate the staging buffers up front (before a hipGraph capture pins the allocator pools)"""
        if not self.active or (self.algo == "ring" and self.payload == "fp32"):
            return
        dt = torch.bfloat16 if self.payload == "bf16" else torch.float32
        self._buf("stage", 1, dt)
        if self.algo == "direct":
            self._buf("recv", 1, dt)
            self._buf("mine", 1, dt)

    def _narrow(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        if self.on_gpu and dst.dtype == torch.bfloat16:
            kn.cast_f32_to_bf16(src, dst, src.numel())
        else:
            dst.copy_(src)

    def _widen(self, src: torch.Tensor, dst: torch.Tensor) -> None:
        if self.on_gpu and src.dtype == torch.bfloat16:
            kn.cast_bf16_to_f32(src, dst, src.numel())
        else:
            dst.copy_(src)

    def reduce(self, lo: int, hi: int, async_op: bool = False):
        """flat_grad[lo:hi] <- sum over ranks, on the current stream.  Returns a work handle only for async_op (CPU ring/fp32 path)."""
        view = self.flat_grad[lo:hi]
        n = hi - lo
        if n <= 0 or not self.active:
            return None
        if self.algo == "ring" and self.payload == "fp32":
            return dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
        dt = torch.bfloat16 if self.payload == "bf16" else torch.float32
        if self.algo == "ring":
            stage = self._buf("stage", n, dt)
            self._narrow(view, stage)
            dist.all_reduce(stage, op=dist.ReduceOp.SUM, group=self.group)
            self._widen(stage, view)
            return None
        W = self.world
        chunk = ((n + W - 1) // W + 7) // 8 * 8
        stage, recv, mine = self._buf("stage", chunk * W, dt), self._buf("recv", chunk * W, dt), self._buf("mine", chunk, dt)
        self._narrow(view, stage[:n])
        if chunk * W > n:
            stage[n:].zero_()
        dist.all_to_all_single(recv, stage, group=self.group)
        if self.on_gpu:
            kn.sum_chunks(recv, W, chunk, mine)
        else:
            mine.copy_(recv.view(W, chunk).float().sum(0))
        dist.all_gather_into_tensor(stage, mine, group=self.group)
        self._widen(stage[:n], view)
        return None


class GradBuckets:
    """Contiguous slices of the gradient arena reduced across ranks, overlapped with backward."""

    def __init__(self, params: List[torch.nn.Parameter], offsets: List[int], flat_grad: torch.Tensor, bucket_bytes: int,
                 group=None, overlap: bool = True, comm: Optional[GradComm] = None):
        self.group = group
        self.comm = comm if comm is not None else GradComm(flat_grad, group)
        self.world = self.comm.world
        self.active = self.comm.active
        self.flat_grad = flat_grad
        self.on_gpu = flat_grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if self.on_gpu else None
        # autograd finishes parameters roughly in reverse registration order: build buckets from the arena's tail
        self.buckets: List[Dict] = []
        cur = {"lo": None, "hi": None, "n": 0, "pending": 0}
        cap = max(bucket_bytes // 4, 1)
        self.param_bucket: Dict[int, int] = {}
        for p, off in reversed(list(zip(params, offsets))):
            n = p.numel()
            if cur["n"] and cur["n"] + n > cap:
                self.buckets.append(cur)
                cur = {"lo": None, "hi": None, "n": 0, "pending": 0}
            cur["lo"] = off
            cur["hi"] = cur["hi"] if cur["hi"] is not None else off + n
            cur["n"] += n
            self.param_bucket[id(p)] = len(self.buckets)
        if cur["n"]:
            self.buckets.append(cur)
        self.members = [0] * len(self.buckets)
        for b in self.param_bucket.values():
            self.members[b] += 1
        self.handles = []
        self.overlap = overlap
        self._hook_handles = []
        if self.active and overlap:
            for p in params:
                self._hook_handles.append(p.register_post_accumulate_grad_hook(self._hook))
        self.reset()

    def close(self) -> None:
        """remove the per-parameter hooks (a dead trainer's buckets must not keep launching all-reduces on its arena)"""
        for h in self._hook_handles:
            h.remove()
        self._hook_handles = []

    def reset(self):
        for i, b in enumerate(self.buckets):
            b["pending"] = self.members[i]
        self.handles = []

    def _launch(self, b):
        if self.on_gpu:
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm.reduce(b["lo"], b["hi"])
        else:
            h = self.comm.reduce(b["lo"], b["hi"], async_op=True)
            if h is not None:
                self.handles.append(h)

    def _hook(self, p):
        b = self.buckets[self.param_bucket[id(p)]]
        b["pending"] -= 1
        if b["pending"] == 0:
            self._launch(b)

    def finish(self):
        """Flush buckets whose parameters produced no gradient this step, then join the comm stream."""
        if self.active and not self.overlap:       # graph mode, eager step: one in-place all-reduce of the whole arena
            # Always on the comm stream, never on the caller's: ProcessGroupNCCL records the work's end event on the stream the collective
            # runs on, and its watchdog thread keeps querying that event for a while after the work is done.  capture() runs its warm-up
            # steps on the very stream it captures next — an event of that stream queried during capture is hipErrorCapturedEvent and the
            # watchdog aborts the process (found by the first RCCL run on hardware, round 2).
            if self.on_gpu:
                cur = torch.cuda.current_stream()
                self.comm_stream.wait_stream(cur)
                with torch.cuda.stream(self.comm_stream):
                    self.comm.reduce(0, self.flat_grad.numel())
                cur.wait_stream(self.comm_stream)
            else:
                self.comm.reduce(0, self.flat_grad.numel())
            return
        if self.active:
            for b in self.buckets:
                if b["pending"] > 0:
                    b["pending"] = 0
                    self._launch(b)
            for h in self.handles:
                h.wait()
            if self.on_gpu:
                torch.cuda.current_stream().wait_stream(self.comm_stream)
        self.reset()


class _LoadHook:
    """load_state_dict post-hook of a model with a trainer: re-derive the kernel-side copies (ArenaTrainer._after_model_load)"""

    def __init__(self, trainer):
        self.ref = weakref.ref(trainer)

    def __call__(self, module, incompatible_keys):
        tr = self.ref()
        if tr is not None and tr.model is module:
            tr._after_model_load(module, incompatible_keys)

    def __deepcopy__(self, memo):
        return self


class ArenaTrainer:
    def __init__(self, model: torch.nn.Module, lr: float = 2e-4, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 0.0,
                 bucket_mb: int = 32, group=None, overlap: bool = True, comm_algo: Optional[str] = None, grad_payload: Optional[str] = None,
                 force_comm: bool = False, shadows_only: bool = False, step_node: bool = False):
        """force_comm: run the multi-rank control flow (split graphs, comm stream, collectives) even with a single rank in the process
        group — how the RCCL path is exercised on a one-GPU box.
        shadows_only (round 4): keep the KERNEL-SIDE COPIES of the parameters only — the parameters move into the fp32 arena, every derived
        copy (bf16 shadow, transposed tiles, packed fragments, remainders, conv repacks) is allocated and registered, refresh_if_stale()
        re-derives all of them with five launches when an external optimizer has stepped (any parameter's version counter moved) — and leave gradients, optimizer state and
        communication to the caller (Lightning + torch.optim.Adam + torch DDP: hulc2/training.py:79-82).  Without it that loop re-derives every
        copy per parameter and layout through torch ops (~200 small launches per step).
        step_node (round 5, with shadows_only): the keeper also owns a GRADIENT arena and registers every parameter's slice as a gradient sink
        that is live only inside the model's step node (hulc2_amd/stepnode.py: the whole forward + backward of a training step as ONE autograd
        node, eager or as two replayed hipGraphs) — the backward kernels write weight gradients straight into the arena (grouped launch,
        first writer overwrites), the node hands the arena views to autograd as the parameters' gradients.  No Adam moments, no communication."""
        self.model = model
        self.shadows_only = bool(shadows_only)
        self.step_node = bool(step_node) and self.shadows_only
        prev = model.__dict__.get("_hulc_arena_trainer")
        prev = prev() if prev is not None else None
        if prev is not None and prev.model is not model:       # (a deep copy of a model carries the original's weak reference along)
            prev = None
        if prev is not None:                               # an earlier trainer of this model: its load hook would keep re-homing weights into a
            prev.close()                                   # dead arena (and keep that arena alive) — ADVICE r02
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.params = [p for p in model.parameters() if p.requires_grad]
        dev = self.params[0].device
        # arena order = registration order, except that modules may ask for groups of parameters to sit back to back
        # (fused_param_groups: the decoder's four heads become one (184, H) matrix view).  Fused views receive their gradient
        # through a sink only, so they exist only when sinks do (not with per-parameter all-reduce hooks).
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.multi = self.world > 1 or (force_comm and dist.is_initialized())
        use_sinks = dev.type == "cuda" and ((not self.multi or not overlap) and not self.shadows_only or self.step_node)
        groups = []
        for mod in model.modules():                     # views installed by an earlier trainer die with its arena
            if getattr(mod, "fused_param_groups", None) is not None:
                mod._fused = None
        if use_sinks:
            for mod in model.modules():
                fn = getattr(mod, "fused_param_groups", None)
                if fn is not None:
                    groups += [(mod, g) for g in fn() if all(p.requires_grad for p in g["params"])]
        member = {id(p): gi for gi, (_, g) in enumerate(groups) for p in g["params"]}
        order, placed, self.group_spans = [], set(), {}
        for p in self.params:
            gi = member.get(id(p))
            if gi is None:
                order.append((p, None))
            elif gi not in placed:
                placed.add(gi)
                order += [(q, gi) for q in groups[gi][1]["params"]]
        self.params = [p for p, _ in order]
        self.offsets, total = [], 0
        for i, (p, gi) in enumerate(order):
            self.offsets.append(total)
            last_of_group = gi is not None and (i + 1 == len(order) or order[i + 1][1] != gi)
            if gi is None:
                total += (p.numel() + 7) // 8 * 8        # 16-byte alignment in both the fp32 and the bf16 arena
            else:
                if gi not in self.group_spans:
                    self.group_spans[gi] = total
                total += p.numel()                        # members tightly packed ...
                if last_of_group:
                    total = (total + groups[gi][1]["pad"] + 7) // 8 * 8   # ... then the group's zero padding
        self._groups = groups
        self.total = total
        self.flat_p = torch.zeros(total, dtype=torch.float32, device=dev)
        _ARENAS[self.flat_p.untyped_storage().data_ptr()] = self    # (hulc2_amd.optim.Adam finds the arena behind a parameter list here)
        n_state = 0 if self.shadows_only else total            # (gradients and Adam moments belong to the caller's optimizer then)
        self.flat_g = torch.zeros(total if self.step_node else n_state, dtype=torch.float32, device=dev)
        self.grads_zeroed_at = None            # flat_g._version at which hulc2_amd.optim.Adam.zero_grad(set_to_none=False) left the arena all zeros
        self.exp_avg = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.exp_avg_sq = torch.zeros(n_state, dtype=torch.float32, device=dev)
        self.flat_bf16 = torch.zeros(total, dtype=torch.bfloat16, device=dev) if dev.type == "cuda" else None
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + n].view(p.shape)
                if not self.shadows_only:
                    p.grad = self.flat_g[off:off + n].view(p.shape)
                if self.flat_bf16 is not None and p.dim() == 2:
                    shadow.register_arena_view(p, self.flat_bf16[off:off + n].view(p.shape))
        # fused group views: parameter view for the kernels, gradient view as its sink, bf16 shadow like any 2-D weight
        self.fused = []
        for gi, (mod, g) in enumerate(groups):
            off, shape = self.group_spans[gi], tuple(g["shape"])
            n = 1
            for d in shape:
                n *= d
            pv, gv = self.flat_p[off:off + n].view(shape), self.flat_g[off:off + n].view(shape)
            if mod._fused is None:
                mod._fused = {}
            mod._fused[g["attr"]] = pv
            self.fused.append((pv, gv, off, shape))
            if len(shape) == 2:
                shadow.register_arena_view(pv, self.flat_bf16[off:off + n].view(shape))
        if self.flat_bf16 is not None:
            kn.cast_f32_to_bf16(self.flat_p, self.flat_bf16, total)
        # transposed bf16 shadows of the nn.Linear weights (data-gradient GEMMs read W^T k-major); RNN and conv weights are
        # consumed in place by their own kernels and stay out of the table
        self.flat_bf16_t = self.tiles_t = None
        if self.flat_bf16 is not None:
            names = {id(p): n for n, p in model.named_parameters()}
            mats = []                                     # (tensor the kernels see, arena offset): single weights and fused groups
            for p, off in zip(self.params, self.offsets):
                nm = names.get(id(p), "")
                if p.dim() != 2 or "rnn.weight_hh" in nm or "rnn.weight_ih_l1" in nm or min(p.shape) < 8 or id(p) in member:
                    continue
                mats.append((p, off))
            mats += [(pv, off) for pv, _, off, shape in self.fused if len(shape) == 2]
            tiles = []
            for t, off in mats:
                r, c = t.shape
                tiles += [(off, r, c, i, j) for i in range((r + 63) // 64) for j in range((c + 63) // 64)]
            if tiles:
                self.flat_bf16_t = torch.zeros(total, dtype=torch.bfloat16, device=dev)
                self.tiles_t = torch.tensor(tiles, dtype=torch.int64, device=dev)
                for t, off in mats:
                    shadow.register_arena_view_t(t, self.flat_bf16_t[off:off + t.numel()].view(t.shape[1], t.shape[0]))
                kn.transpose_bf16_tiles(self.flat_bf16, self.flat_bf16_t, self.tiles_t)
        # fragment-packed feed-forward weights of the transformer block launch (modules list them in frag_operands()): one gather launch per
        # step from the bf16 arena and its transposed shadow, 8-byte chunks
        self.frag_shadow = self.frag_idx = None
        if self.flat_bf16 is not None and self.flat_bf16_t is not None:
            off_of = {id(p): off for p, off in zip(self.params, self.offsets)}
            chunks, views, dst = [], [], 0
            for m in model.modules():
                if not hasattr(m, "frag_operands"):
                    continue
                for p, n in m.frag_operands():
                    off = off_of.get(id(p))
                    if off is None or off % 4 or id(p) in member:
                        continue
                    ff = p.shape[0] if n in (0, 3) else p.shape[1]
                    perm = kn.ffn_frag_perm(n, ff).astype("int64").reshape(-1, 4)
                    assert (perm[:, 0] % 4 == 0).all() and (perm[:, 1:] - perm[:, :1] == [1, 2, 3]).all(), "fragment permutations move runs of 4"
                    c = (perm[:, 0] + off) // 4
                    chunks.append(c | (1 << 31) if n in (2, 3) else c)
                    views.append((p, "ffn_p%d" % n, dst, p.numel()))
                    dst += p.numel()
            if chunks:
                import numpy as np
                self.frag_shadow = torch.zeros(dst, dtype=torch.bfloat16, device=dev)
                self.frag_idx = torch.from_numpy(np.concatenate(chunks).astype(np.uint32).view(np.int32)).to(dev)
                for p, name, d0, n_el in views:
                    shadow.register_layout_view(p, name, self.frag_shadow[d0:d0 + n_el])
                kn.gather_chunks(self.flat_bf16, self.flat_bf16_t, self.frag_shadow, self.frag_idx)
        # rounding remainders w - bf16(w) of the weights a split-operand forward reads (modules list them in lo_operands(): natural layout
        # "lo", packed "ffn_p0_lo" / "ffn_p1_lo").  Round 4: a second shadow ARENA (same offsets as the bf16 shadow; 94 MB of the 288 GB) that
        # the Adam kernel fills inside <= 8 element ranges while the new weights are in its registers — the separate residual launch
        # (66-73 us per step) is only used when weights are written from outside (refresh_shadows)
        self.flat_lo = self.lo_seg = self.lo_frag = self.lo_frag_idx = None
        self.lo_ranges = []
        if self.flat_bf16 is not None:
            off_of = {id(p): off for p, off in zip(self.params, self.offsets)}
            segs, nat, packed = [], {}, []
            for m in model.modules():
                if not hasattr(m, "lo_operands"):
                    continue
                for p, layout in m.lo_operands():
                    off = off_of.get(id(p))
                    if off is None or off % 4 or id(p) in member:
                        continue
                    if id(p) not in nat:
                        nat[id(p)] = (p, off)
                        segs.append((off, p.numel(), off))
                    if layout not in ("lo", "oihw_flat_lo"):
                        packed.append((p, layout))
            if segs:
                import numpy as np
                self.flat_lo = torch.zeros(total, dtype=torch.bfloat16, device=dev)
                self.lo_seg = torch.tensor(segs, dtype=torch.int64, device=dev)
                ranges = sorted([a, a + (n + 3) // 4 * 4] for a, n, _ in segs)      # (offsets are multiples of 8: the padding is the slice's own)
                merged = []
                for a, b in ranges:
                    if merged and a <= merged[-1][1]:
                        merged[-1][1] = max(merged[-1][1], b)
                    else:
                        merged.append([a, b])
                while len(merged) > 8:                       # the kernel takes 8: close the smallest gaps (remainders of the weights between
                    gi = min(range(len(merged) - 1), key=lambda i: merged[i + 1][0] - merged[i][1])   # them are written too: harmless)
                    merged[gi][1] = merged[gi + 1][1]
                    del merged[gi + 1]
                self.lo_ranges = [(a, min(b, total)) for a, b in merged]
                for p, d0 in nat.values():
                    shadow.register_layout_view(p, "lo", self.flat_lo[d0:d0 + p.numel()].view(p.shape))
                    if p.dim() == 4:                        # conv weight: its OIHW-flat remainder is the same memory
                        shadow.register_layout_view(p, "oihw_flat_lo", self.flat_lo[d0:d0 + p.numel()].view(p.shape[0], -1))
                chunks, views, fdst = [], [], 0
                for p, layout in packed:
                    n = int(layout[5])
                    ff = p.shape[0] if n in (0, 3) else p.shape[1]
                    perm = kn.ffn_frag_perm(n, ff).astype("int64").reshape(-1, 4)
                    chunks.append((perm[:, 0] + nat[id(p)][1]) // 4)
                    views.append((p, layout, fdst, p.numel()))
                    fdst += p.numel()
                if chunks:
                    self.lo_frag = torch.zeros(fdst, dtype=torch.bfloat16, device=dev)
                    self.lo_frag_idx = torch.from_numpy(np.concatenate(chunks).astype(np.uint32).view(np.int32)).to(dev)
                    for p, name, d0, n_el in views:
                        shadow.register_layout_view(p, name, self.lo_frag[d0:d0 + n_el])
                self._refresh_lo()
        # conv weights in their kernel layouts (OIHW flat for conv1, OHWI forward, IHWO data gradient): one repack launch per step
        self.conv_shadow = self.conv_table = None
        if self.flat_bf16 is not None:
            rows, views, dst = [], [], 0
            flat_lin = {id(w): chw for m in model.modules() if hasattr(m, "flatten_linears") for w, chw in m.flatten_linears()}
            for p, off in zip(self.params, self.offsets):
                chw = flat_lin.get(id(p))
                if p.dim() == 2 and chw is not None:          # Linear behind nn.Flatten of a (C, H, W) map (gripper encoder): NHWC column order
                    co, (ci, kh, kw) = p.shape[0], chw
                    # (+ the rounding remainder in the forward layout: the exact-forward site of the flatten-linear is TWO bf16 products on
                    #  the bf16 activation — a w_hi + a w_lo — instead of an fp32 GEMM on a cast copy; round 5)
                    modes = [("hwc", 1, (co, kh * kw * ci)), ("hwc_t", 3, (kh * kw * ci, co)), ("hwc_lo", 1 | 8, (co, kh * kw * ci))]
                elif p.dim() != 4:
                    continue
                else:
                    co, ci, kh, kw = p.shape
                    modes = [("oihw_flat", 0, (co, ci * kh * kw))] if ci < 8 else [("ohwi", 1, (co, kh * kw * ci)), ("ihwo", 2, (ci, kh, kw, co))]
                for name, mode, shape in modes:
                    rows.append((off, dst, co, ci, kh, kw, mode))
                    views.append((p, name, dst, shape))
                    dst += (p.numel() + 7) // 8 * 8
            if rows:
                self.conv_shadow = torch.zeros(dst, dtype=torch.bfloat16, device=dev)
                self.conv_table = torch.tensor(rows, dtype=torch.int64, device=dev)
                for p, name, d0, shape in views:
                    shadow.register_layout_view(p, name, self.conv_shadow[d0:d0 + p.numel()].view(shape))
                kn.repack_conv_weights(self.flat_p, self.conv_shadow, self.conv_table)
        self.comm = self.buckets = None
        if not self.shadows_only:
            self.comm = GradComm(self.flat_g, group, comm_algo, grad_payload, force=force_comm)
            self.buckets = GradBuckets(self.params, self.offsets, self.flat_g, bucket_mb << 20, group, overlap, comm=self.comm)
            # bucket all-reduces overlapped with backward share the GPU with the compute stream: barrier kernels are then off
            kn.set_concurrent_streams(dev.type == "cuda" and self.multi and overlap)
        self._autograd_written, self._zero_planned, self._acc_hooks, self._sink_keys = set(), None, [], []
        self._replan_pending = False
        if use_sinks:                                     # no per-parameter all-reduce hooks depend on AccumulateGrad
            owner = self if self.shadows_only else None       # (the keeper's sinks are live inside its model's step node only)
            for p, off in zip(self.params, self.offsets):
                self._sink_keys.append(gradsink.register(p, self.flat_g[off:off + p.numel()].view(p.shape), owner))
                if not self.shadows_only:                     # (the step node takes autograd-made gradients with autograd.grad: nothing accumulates into the arena)
                    self._acc_hooks.append(p.register_post_accumulate_grad_hook(self._saw_autograd_grad))
            for pv, gv, _, _ in self.fused:
                self._sink_keys.append(gradsink.register(pv, gv, owner))
        self.step_count = 0
        self.dev = dev
        if dev.type == "cuda" and not self.shadows_only:
            kn.step_state(dev)[1] = 0               # the device-resident Adam step count starts with this trainer (the RNG word keeps walking)
        self.graph_fb = self.graph_enc = self.graph_opt = None
        self.static_loss = None
        self._comm_events = []
        # (not a bound method: copy.deepcopy(model) copies the module's hook table, and a bound method would drag the trainer — arenas and all —
        # into the copy; the copy's hook finds that the trainer belongs to another module and does nothing)
        self._load_hook = model.register_load_state_dict_post_hook(_LoadHook(self))
        model.__dict__["_hulc_arena_trainer"] = weakref.ref(self)
        # Split point for overlapping the gradient all-reduce with the tail of backward in graph mode: the camera encoders
        # are registered first (arena head, 0.75 M parameters) but their backward (the conv stack) is the LAST ~2 ms of a step,
        # while everything else (98 % of the gradient bytes) is complete once backward reaches the encoder output.
        names = {id(p): n for n, p in model.named_parameters()}
        enc = [names.get(id(p), "").startswith("perceptual_encoder.") for p in self.params]
        idx = [i for i, e in enumerate(enc) if e]
        self.enc_lo = self.enc_hi = 0                      # arena slice [enc_lo, enc_hi) = the encoder gradients (one contiguous run)
        self.enc_params, self.rest_params = [], list(self.params)
        if idx and len(idx) < len(self.params) and idx[-1] - idx[0] + 1 == len(idx) and hasattr(model, "perceptual_encoder"):
            self.enc_lo = self.offsets[idx[0]]
            self.enc_hi = self.offsets[idx[-1] + 1] if idx[-1] + 1 < len(self.params) else total
            self.enc_params = [self.params[i] for i in idx]
            self.rest_params = [p for i, p in enumerate(self.params) if not enc[i]]
            if not self.shadows_only:                             # (the split backward belongs to this trainer's own step)
                model.perceptual_encoder.register_forward_hook(self._keep_encoder_output)
        self._emb = None

    def __deepcopy__(self, memo):
        """the shadows-only keeper is not copied with its model (copy.deepcopy(model) reaches it through the model's __dict__): the copy
        starts without one and builds its own on its first training step.  A full trainer copies like any object."""
        if self.shadows_only:
            return None
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, copy.deepcopy(v, memo))
        return new

    def refresh_if_stale(self) -> bool:
        """shadows_only mode: an external optimizer steps the parameters in place (views of the arena) and bumps their version counters; one
        look at three of them decides whether the derived copies are re-made (five launches for the whole model)"""
        ps = self.params
        sig = sum(p._version for p in ps)                      # (every in-place write to any parameter — optimizer, load, a test's nudge — moves it)
        if sig == getattr(self, "_fresh_sig", None):
            return False
        for p, off in zip(ps, self.offsets):                  # (an optimizer that REPLACED .data would have left the arena)
            if p.data_ptr() != self.flat_p.data_ptr() + off * 4:
                self._after_model_load(None, None)
                break
        else:
            self.refresh_shadows()
        self._fresh_sig = sig
        return True

    def _refresh_lo(self) -> None:
        """the remainders from scratch (weights written from outside; the optimizer step keeps them fresh inside the Adam kernel)"""
        if self.lo_seg is not None:
            kn.residual_bf16(self.flat_p, self.flat_bf16, self.flat_lo, self.lo_seg)
            if self.lo_frag_idx is not None:
                kn.gather_chunks(self.flat_lo, None, self.lo_frag, self.lo_frag_idx)

    # ---- weights written from outside (checkpoint restore) and optimizer state ----------------------------------------------------
    def refresh_shadows(self) -> None:
        """Re-derive every kernel-side copy of the parameters from the fp32 arena: the bf16 shadow, its transposed tiles and the conv
        repacks.  The Adam kernel keeps them fresh step by step; anything else that writes the parameters (model.load_state_dict —
        Lightning restores weights AFTER the optimizer exists, hulc2/training.py:41-53,82 —, an external p.data.copy_) must be followed by
        this call, else the next forward/backward runs on the old weights.  Installed as a load_state_dict post-hook on the model."""
        if self.flat_bf16 is not None:
            kn.cast_f32_to_bf16(self.flat_p, self.flat_bf16, self.total)
            if self.tiles_t is not None:
                kn.transpose_bf16_tiles(self.flat_bf16, self.flat_bf16_t, self.tiles_t)
            if self.frag_idx is not None:
                kn.gather_chunks(self.flat_bf16, self.flat_bf16_t, self.frag_shadow, self.frag_idx)
            self._refresh_lo()
            if self.conv_table is not None:
                kn.repack_conv_weights(self.flat_p, self.conv_shadow, self.conv_table)
        shadow.bump_epoch()

    def _after_model_load(self, module, incompatible_keys) -> None:
        for p, off in zip(self.params, self.offsets):         # a load that REPLACED .data (assign=True) would detach the arena: re-home it
            if p.data_ptr() != self.flat_p.data_ptr() + off * 4:
                with torch.no_grad():
                    self.flat_p[off:off + p.numel()].copy_(p.reshape(-1))
                    p.data = self.flat_p[off:off + p.numel()].view(p.shape)
        self.refresh_shadows()

    def state_dict(self) -> Dict:
        """optimizer state in the reference's terms (a Lightning checkpoint carries torch.optim.Adam's exp_avg / exp_avg_sq / step per
        parameter, hulc2.py:185-198): per-parameter tensors keyed by the model's parameter names, so the arena layout can change between
        save and load; plus the device step words (Adam step count, RNG word)."""
        names = {id(p): n for n, p in self.model.named_parameters()}
        st = {}
        for p, off in zip(self.params, self.offsets):
            n = p.numel()
            st[names[id(p)]] = {"exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                                "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        words = kn.step_state(self.dev).tolist() if self.dev.type == "cuda" else [0, self.step_count]
        return {"state": st, "step": int(self.step_count), "rng_word": int(words[0]), "device_step": int(words[1]),
                "comm": ({"algo": self.comm.algo, "payload": self.comm.payload, "chosen_by": self.comm.chosen_by} if self.comm is not None else None),
                "hparams": {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd}}

    def load_state_dict(self, sd: Dict) -> None:
        names = {n: p for n, p in self.model.named_parameters()}
        index = {id(p): off for p, off in zip(self.params, self.offsets)}
        missing = [n for n, p in names.items() if id(p) in index and n not in sd["state"]]
        if missing:
            raise KeyError(f"ArenaTrainer.load_state_dict: no optimizer state for {missing[:4]}{'...' if len(missing) > 4 else ''}")
        with torch.no_grad():
            for n, rec in sd["state"].items():
                p = names.get(n)
                if p is None or id(p) not in index:
                    continue
                off, k = index[id(p)], p.numel()
                self.exp_avg[off:off + k].copy_(rec["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + k].copy_(rec["exp_avg_sq"].reshape(-1))
        self.step_count = int(sd["step"])
        hp = sd.get("hparams", {})
        self._set_hparams(hp.get("lr", self.lr), hp.get("betas", self.betas), hp.get("eps", self.eps), hp.get("weight_decay", self.wd))
        algo = (sd.get("comm") or {}).get("algo")
        if algo in ("ring", "direct") and self.comm is not None and self.comm.active and not os.environ.get("HULC_ALLREDUCE"):
            self.comm.pin(algo)                                   # a resumed run keeps the summation order it started with
        if self.dev.type == "cuda":
            kn.reset_step_state(self.dev, seed=int(sd["rng_word"]), step=int(sd.get("device_step", sd["step"])))
        self.refresh_shadows()

    # ---- interchange with the reference's checkpoints: `optimizer_states[0]` of a Lightning checkpoint IS torch.optim.Adam.state_dict() ----------
    def to_torch_adam_state_dict(self) -> Dict:
        """This trainer's state as `torch.optim.Adam(model.parameters(), lr).state_dict()` would hold it (reference: hulc2.py:185-198,
        conf/model/optimizer/adam.yaml): `state` keyed by the INDEX of the parameter in `model.parameters()` order with a per-parameter `step`
        tensor, `param_groups[0]["params"]` = all indices.  Frozen / never-updated parameters have no entry, like in torch.  A reference
        Lightning run resumes from it with `optimizer.load_state_dict(...)`."""
        order = list(self.model.parameters())
        index = {id(p): off for p, off in zip(self.params, self.offsets)}
        state = {}
        if self.step_count > 0:
            for i, p in enumerate(order):
                off = index.get(id(p))
                if off is None:
                    continue
                n = p.numel()
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": self.wd, "amsgrad": False, "maximize": False,
                 "foreach": None, "capturable": False, "differentiable": False, "fused": None, "decoupled_weight_decay": False,
                 "params": list(range(len(order)))}
        return {"state": state, "param_groups": [group]}

    def from_torch_adam_state_dict(self, sd: Dict) -> None:
        """Resume from a reference checkpoint's optimizer state (`ckpt["optimizer_states"][0]`): the inverse of to_torch_adam_state_dict.
        The fused kernel keeps ONE step count, so the per-parameter steps must agree (they do for a torch.optim.Adam that stepped all its
        parameters together — every parameter of this model receives a gradient every step)."""
        order = list(self.model.parameters())
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(order):
            raise KeyError(f"optimizer state for {len(ids)} parameters, model.parameters() has {len(order)}")
        index = {id(p): off for p, off in zip(self.params, self.offsets)}
        steps = set()
        with torch.no_grad():
            for pos, key in enumerate(ids):
                p, rec = order[pos], sd["state"].get(key)
                off = index.get(id(p))
                if off is None:
                    continue
                n = p.numel()
                if rec is None:                                  # torch creates state lazily: a parameter that never saw a gradient
                    self.exp_avg[off:off + n].zero_()
                    self.exp_avg_sq[off:off + n].zero_()
                    continue
                if tuple(rec["exp_avg"].shape) != tuple(p.shape):
                    raise KeyError(f"optimizer state {key}: shape {tuple(rec['exp_avg'].shape)} vs parameter {tuple(p.shape)}")
                self.exp_avg[off:off + n].copy_(rec["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(rec["exp_avg_sq"].reshape(-1))
                steps.add(int(float(rec["step"])))
        if len(steps) > 1:
            raise ValueError(f"per-parameter Adam steps differ ({sorted(steps)}): the arena optimizer keeps one step count")
        self.step_count = steps.pop() if steps else 0
        g0 = groups[0]
        if g0.get("amsgrad") or g0.get("maximize"):
            raise NotImplementedError("amsgrad / maximize are not built (conf/model/optimizer/adam.yaml uses neither)")
        self._set_hparams(g0["lr"], g0["betas"], g0["eps"], g0.get("weight_decay", 0.0))
        if self.dev.type == "cuda":
            kn.step_state(self.dev)[1] = self.step_count
        self.refresh_shadows()

    def _set_hparams(self, lr, betas, eps, wd) -> None:
        """lr / betas / eps / weight decay are scalar kernel arguments: a captured optimizer graph has the OLD ones baked in, so a change
        after capture() drops the graphs — replay() then asks for a new capture() instead of silently stepping with the pre-load values
        (the lr in a Lightning checkpoint is the scheduler's current value, not the constructor's)."""
        new = (float(lr), tuple(float(b) for b in betas), float(eps), float(wd))
        old = (float(self.lr), tuple(float(b) for b in self.betas), float(self.eps), float(self.wd))
        self.lr, self.betas, self.eps, self.wd = lr, tuple(betas), eps, wd
        if new != old and getattr(self, "graph_opt", None) is not None:
            self.graph_fb = self.graph_enc = self.graph_opt = None

    def close(self) -> None:
        """Detach this trainer from the model: the load_state_dict post-hook (which keeps the trainer, hence its four arenas, alive through the
        model) is removed and the gradient sinks are dropped.  Parameters keep living in the arena until another trainer re-homes them."""
        h, self._load_hook = getattr(self, "_load_hook", None), None
        if h is not None:
            h.remove()
        for h in getattr(self, "_acc_hooks", ()):
            h.remove()
        self._acc_hooks = []
        b = getattr(self, "buckets", None)
        if b is not None:
            b.close()
        gradsink.unregister(getattr(self, "_sink_keys", ()))       # this trainer's sinks only (another live trainer keeps its own)
        self._sink_keys = []

    def _saw_autograd_grad(self, p) -> None:
        """post-accumulate hook: this parameter's gradient arrives through autograd's `grad +=` (not a kernel-side sink), so its arena slice must
        be zeroed before every step.  One that shows up only after the zeroing plan was made asks for a new plan — through a flag that
        optimizer_step() acts on AFTER its clean-up of this step: the backward that is running was started in overwrite mode and must be
        finished in it (clearing the plan here skipped the clean-up of planned-but-unwritten sinks for this very step)."""
        k = id(p)
        if k not in self._autograd_written:
            self._autograd_written.add(k)
            if self._zero_planned is not None and k not in self._zero_planned:
                self._replan_pending = True

    def _keep_encoder_output(self, module, inputs, output):
        self._emb = output if (self._split_active and torch.is_tensor(output) and output.requires_grad) else None

    _split_active = False

    def _forward_backward_head(self, batch, batch_idx: int) -> torch.Tensor:
        """forward + backward down to the encoder output: every gradient outside the camera encoders is final afterwards"""
        shadow.bump_epoch()
        kn.advance_step_state(self.dev)
        kn.wgrad_reset(self.dev)
        self.zero_grad()
        self._split_active = True
        try:
            loss = self.model.training_step(batch, batch_idx)
        finally:
            self._split_active = False
        emb = self._emb
        if emb is None:
            raise RuntimeError("split backward: the perceptual encoder was not called exactly through its module (no output captured)")
        torch.autograd.backward(loss, grad_tensors=self._one_like(loss), inputs=[emb] + self.rest_params)
        kn.wgrad_flush(self.dev)                     # (autograd's end-of-pass callback already issued it: no-op unless the pass was cut short)
        return loss.detach()

    def _backward_encoder(self) -> None:
        emb, self._emb = self._emb, None
        g, emb.grad = emb.grad, None
        torch.autograd.backward(emb, grad_tensors=g, inputs=self.enc_params)
        kn.wgrad_flush(self.dev)

    def _plan_partial_zero(self) -> None:
        """After a fully zeroed step: the arena slices NOT written through a gradient sink are the only ones the next steps need zeroed
        (autograd's `param.grad += g` lands there); sinks written by the backward kernels are overwritten by their first writer
        (gradsink.first_write).  Neighbouring must-zero slices are merged across small written ones — zeroing a slice that is overwritten
        later is harmless, one fill launch per slice is not."""
        spans = []                                                 # arena ranges of fused groups whose single sink was written
        for pv, gv, off, shape in self.fused:
            if gradsink.written(pv):
                spans.append((off, off + gv.numel()))
        inside = lambda a, b: any(lo <= a and b <= hi for lo, hi in spans)
        # ... and of those only the ones autograd really accumulates into (seen by the post-accumulate hooks during this first, fully zeroed
        # step): a parameter nobody writes (an unused module of the reference's constructor, e.g. plan_recognition.layernorm) stays zero
        need = [(off, off + p.numel()) for p, off in zip(self.params, self.offsets)
                if not (gradsink.written(p) or inside(off, off + p.numel())) and id(p) in self._autograd_written]
        self._zero_planned = {id(p) for p in self.params if id(p) in self._autograd_written}
        merged = []
        for a, b in sorted(need):
            if merged and a - merged[-1][1] <= (1 << 18):          # gaps up to 1 MB of fp32 are cheaper to zero than another launch
                merged[-1][1] = max(merged[-1][1], b)
            else:
                merged.append([a, b])
        self._zero_ranges = [(a, b) for a, b in merged]
        self._planned_written = gradsink.written_ids()
        self._sink_slices = {id(p): (off, off + p.numel()) for p, off in zip(self.params, self.offsets)}
        self._sink_slices.update({id(pv): (off, off + gv.numel()) for pv, gv, off, _ in self.fused})

    _zero_ranges = None

    def _zero_arena(self) -> None:
        """what a backward pass needs of the gradient arena before it starts: everything zeroed (first pass: sinks accumulate) or only the slices
        autograd accumulates into (later passes: the first sink writer of a slice overwrites it)"""
        if self._zero_ranges is None or os.environ.get("HULC_FULL_ZERO_GRAD"):
            self.flat_g.zero_()
            gradsink.begin_step(False)
        else:
            for a, b in self._zero_ranges:
                self.flat_g[a:b].zero_()
            gradsink.begin_step(True)

    def zero_grad(self):
        if self.shadows_only:
            raise RuntimeError("ArenaTrainer(shadows_only=True) keeps the kernel-side weight copies only: gradients and the optimizer are the caller's")
        self.grads_zeroed_at = None                        # (optim.Adam.zero_grad's mark: this pass writes the arena with kernels of its own)
        self._zero_arena()
        for p, off in zip(self.params, self.offsets):      # autograd may have replaced .grad; re-point at the arena
            if p.grad is None or p.grad.data_ptr() != self.flat_g.data_ptr() + off * 4:
                p.grad = self.flat_g[off:off + p.numel()].view(p.shape)

    def _settle_sinks(self) -> None:
        """bookkeeping behind a finished backward pass: the first (fully zeroed) pass makes the zeroing plan; later passes clean up after it"""
        if self._zero_ranges is None and gradsink._sinks and not os.environ.get("HULC_FULL_ZERO_GRAD"):
            self._plan_partial_zero()                              # the backward that just finished ran on a fully zeroed arena
        elif self._zero_ranges is not None:
            # a sink the plan expects to be overwritten was not written by this backward (a branch of the model did not run): its slice
            # still holds the previous step's gradient — the true gradient is zero
            now = gradsink.written_ids()
            for k in self._planned_written - now:
                a, b = self._sink_slices[k]
                self.flat_g[a:b].zero_()
            # a sink written for the first time AFTER the plan was made (untouched in the planning step, so neither zeroed per step nor
            # expected to be overwritten): from now on it is "expected" — a later step that does not write it gets the clean-up above
            # instead of Adam applying the stale slice (second case)
            new = {k for k in now if k in self._sink_slices} - self._planned_written
            if new:
                self._planned_written = self._planned_written | new
            if self._replan_pending:
                # a parameter autograd accumulates into showed up after the plan: its slice was zero when this backward started (unplanned
                # slices are only ever written through sinks, and those are cleaned above), so THIS step's sum is right; the next step
                # runs on a fully zeroed arena and the plan is re-made behind it
                self._zero_ranges, self._replan_pending = None, False

    def optimizer_step(self):
        if self.shadows_only:
            raise RuntimeError("ArenaTrainer(shadows_only=True) keeps the kernel-side weight copies only: gradients and the optimizer are the caller's")
        self._settle_sinks()
        self.step_count += 1
        kn.adam_step(self.flat_p, self.flat_g, self.exp_avg, self.exp_avg_sq, self.flat_bf16, self.total, self.lr, self.betas[0],
                     self.betas[1], self.eps, self.wd, self.step_count, grad_scale=1.0 / self.world,
                     step_state_dev=kn.step_state(self.dev),       # step count lives on the device (graph replay)
                     lo=self.flat_lo, lo_ranges=self.lo_ranges)    # (the split operands' remainders come out of the same pass)
        # the derived copies: two launches behind Adam (were five: transposed tiles, fragment gather, residual, remainder gather, conv repack)
        if self.tiles_t is not None or self.conv_table is not None:
            kn.derive_copies(self.flat_bf16, self.flat_bf16_t, self.tiles_t, self.flat_p, self.conv_shadow, self.conv_table)
        if self.frag_idx is not None or self.lo_frag_idx is not None:
            kn.gather_chunks2(self.flat_bf16, self.flat_bf16_t, self.frag_shadow, self.frag_idx, self.flat_lo, self.lo_frag, self.lo_frag_idx)
        shadow.bump_epoch()

    def _forward_backward(self, batch, batch_idx: int) -> torch.Tensor:
        shadow.bump_epoch()                    # every repack/shadow is re-made inside this step (and inside a capture)
        kn.advance_step_state(self.dev)        # fresh dropout / plan-sample stream, step count + 1
        kn.wgrad_reset(self.dev)
        self.zero_grad()
        loss = self.model.training_step(batch, batch_idx)
        with kn.wgrad_branch_scope():          # (the recurrent weight gradients as a third branch of this graph: functional.DecoderRNNFn.backward)
            torch.autograd.backward(loss, grad_tensors=self._one_like(loss))
            kn.wgrad_flush(self.dev)
        return loss.detach()

    def _one_like(self, loss: torch.Tensor) -> torch.Tensor:
        """d loss / d loss = 1, kept per (shape, device): backward() would fill a fresh ones tensor every step"""
        one = self.__dict__.get("_one")
        if one is None or one.shape != loss.shape or one.device != loss.device or one.dtype != loss.dtype:
            one = self.__dict__["_one"] = torch.ones_like(loss)
        return one

    # Hook for processes that SHARE one GPU (tests: two gloo ranks on a one-GPU box): a context-manager factory entered around every stretch of GPU
    # work that holds no collective.  The device-wide-barrier kernels need the GPU to themselves, so such ranks take turns (a file lock that
    # synchronises the device before it is released).  Default: nothing.
    gpu_section = staticmethod(contextlib.nullcontext)
    time_comm = False              # bench.py: record an event pair around the wait for the gradient exchange (exposed communication)

    def _comm_wait_begin(self):
        if self.time_comm and self.dev.type == "cuda":
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            return ev
        return None

    def _comm_wait_end(self, ev) -> None:
        if ev is not None:
            ev[1].record()
            self._comm_events.append(ev)

    def comm_exposed_ms(self) -> float:
        """mean time per step the compute stream spent waiting for the gradient exchange since the last call (synchronises)"""
        evs, self._comm_events = self._comm_events, []
        if not evs:
            return 0.0
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in evs) / len(evs)

    def step(self, batch, batch_idx: int = 0) -> torch.Tensor:
        """zero grads -> training_step -> backward (+ overlapped all-reduce) -> fused Adam.  Returns the detached loss."""
        with (self.gpu_section() if not (self.multi and self.buckets.overlap) else contextlib.nullcontext()):
            loss = self._forward_backward(batch, batch_idx)
        ev = self._comm_wait_begin()
        self.buckets.finish()
        self._comm_wait_end(ev)
        with self.gpu_section():
            self.optimizer_step()
            self._poll_faults()
        return loss

    _fault_every = int(os.environ.get("HULC_FAULT_CHECK_EVERY", "64"))
    _since_check = 0

    def _poll_faults(self) -> None:
        """every N steps (one device sync): a barrier-kernel timeout becomes an exception; the Adam kernel has skipped the faulty steps"""
        self._since_check += 1
        if self.dev.type == "cuda" and self._since_check >= self._fault_every:
            self._since_check = 0
            try:
                kn.check_faults(self.dev)
            except Exception:
                # a timed-out chain kernel leaves its barrier counters non-zero in a workspace whose address is baked into the captured
                # graphs (or lives in their private pool): replaying them again would mis-count every barrier.  The graphs are dropped —
                # step() keeps working eagerly on fresh workspaces, replay() asks for a new capture()  
                self.graph_fb = self.graph_enc = self.graph_opt = None
                raise

    # ---- hipGraph mode: the ~900 launches of a step are captured once and replayed ----------------------
    def capture(self, batch) -> None:
        """Capture forward+backward and the optimizer as two HIP graphs over the (static, device-resident) batch.
        Between them the gradient arena is all-reduced eagerly when world > 1 (construct with overlap=False).
        Call after a few eager warm-up steps (allocator pools, lazy scratch buffers and kernels are then live)."""
        assert self.dev.type == "cuda"
        if self.multi and self.buckets.overlap:
            raise RuntimeError("graph mode needs ArenaTrainer(overlap=False): bucket hooks cannot run inside a replayed graph")
        self.comm.reserve()
        torch.autograd.graph.set_warn_on_accumulate_grad_stream_mismatch(False)
        torch.cuda.synchronize()
        side = kn.capture_stream(self.dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                # AccumulateGrad nodes must be born on the capture stream
            for i in range(2):
                self.step(batch, i)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph_fb = torch.cuda.CUDAGraph()
        # world > 1: the process group's watchdog thread may touch the runtime while this thread captures; only this thread's calls are policed
        mode = {"capture_error_mode": os.environ.get("HULC_CAPTURE_MODE", "thread_local")} if self.multi else {}
        self.graph_opt = torch.cuda.CUDAGraph()
        with kn.no_gc():                             # (no garbage collection inside a capture: see kernels.no_gc)
            if self.multi and self.enc_hi > self.enc_lo and not os.environ.get("HULC_NO_SPLIT_GRAPH"):
                # two graphs around the split point; replay() launches the big all-reduce between them on the comm stream
                with torch.cuda.graph(self.graph_fb, stream=side, **mode):
                    self.static_loss = self._forward_backward_head(batch, 0)
                self.graph_enc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(self.graph_enc, pool=self.graph_fb.pool(), stream=side, **mode):
                    self._backward_encoder()
            else:
                with torch.cuda.graph(self.graph_fb, stream=side, **mode):
                    self.static_loss = self._forward_backward(batch, 0)
            with torch.cuda.graph(self.graph_opt, pool=self.graph_fb.pool(), stream=side, **mode):
                self.optimizer_step()
        torch.cuda.synchronize()

    def replay(self) -> torch.Tensor:
        if self.graph_fb is None:
            raise RuntimeError("ArenaTrainer.replay: no captured graphs (never captured, or dropped after a barrier-kernel fault): call capture(batch)")
        with self.gpu_section():
            self.graph_fb.replay()
        if self.graph_enc is not None:
            # everything but the encoder gradients is final: reduce it on the comm stream while the conv backward graph runs
            cur, comm = torch.cuda.current_stream(), self.buckets.comm_stream
            comm.wait_stream(cur)
            with torch.cuda.stream(comm):
                for lo, hi in ((0, self.enc_lo), (self.enc_hi, self.total)):
                    self.comm.reduce(lo, hi)
            with self.gpu_section():
                self.graph_enc.replay()
            comm.wait_stream(cur)                    # the small encoder slice follows on the SAME stream: the staging buffers of the
            with torch.cuda.stream(comm):            # bf16 / direct modes are shared, two reduces must never run concurrently
                self.comm.reduce(self.enc_lo, self.enc_hi)
            ev = self._comm_wait_begin()             # (recorded behind the conv backward graph: what follows is exposed communication)
            cur.wait_stream(comm)
            self._comm_wait_end(ev)
        else:
            ev = self._comm_wait_begin()
            self.buckets.finish()
            self._comm_wait_end(ev)
        with self.gpu_section():
            self.graph_opt.replay()
            self._poll_faults()
        return self.static_loss
