"""hulc2_amd.optim.Adam — torch.optim.Adam for a model whose parameters live in the fp32 arena of hulc2_amd's weight keeper.

reference: hulc2/models/hulc2.py:185-198 (`configure_optimizers` instantiates `optimizer._target_`, conf/model/optimizer/adam.yaml:
`torch.optim.Adam`, lr 2e-4).  Swapping that one class path keeps the update rule, the hyper-parameters and the `state_dict()` layout
(per parameter `step` / `exp_avg` / `exp_avg_sq`: a Lightning checkpoint of either optimizer loads into the other) and replaces the step
itself: torch's multi-tensor Adam walks 212 tensors (~1.5 ms of host time per step, the largest single item of the eager loop's host
budget, tools/eager_profile.py) and leaves the kernel-side weight copies stale, so the keeper re-derives them before the next forward
(five launches over the whole arena); here the gradients are gathered into one flat buffer (one `_foreach_copy_`) and ONE launch of the
arena Adam kernel updates parameters and moments and writes the bf16 shadow and the split operands' remainders, two more derive the
transposed / repacked copies — exactly what ArenaTrainer.optimizer_step launches.

Whenever the fused form does not apply — parameters not (yet) in an arena, a parameter without a gradient, amsgrad / maximize / decoupled weight decay,
several parameter groups, CPU — `step()` is torch.optim.Adam.step(), on the same state tensors."""
from typing import Optional

import torch

from . import kernels as kn
from . import shadow
from .trainer import arena_of


class Adam(torch.optim.Adam):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False, **kw):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=amsgrad, **kw)
        self._arena = None           # (tr, flat_g, grad views, exp_avg, exp_avg_sq) once the parameters are found in a keeper's arena
        self._live, self._mv = set(), ([], [])
        self._fused_steps = 0        # the step count of the fused path (torch keeps one `step` tensor per parameter: written back on demand)
        self.fused_launches = 0      # (tests / logging: steps taken by the fused path)

    # ---- arena binding -----------------------------------------------------------------------------------------------------------------
    def _bind(self) -> Optional[tuple]:
        if len(self.param_groups) != 1:
            return None
        g = self.param_groups[0]
        if g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("capturable") or g.get("decoupled_weight_decay"):
            return None
        params = [p for p in g["params"] if p.requires_grad]     # (frozen parameters never get a gradient: torch skips them, the arena does not hold them)
        tr = arena_of(params)
        if tr is None:
            self._arena = None
            return None
        if self._arena is not None and self._arena[0] is tr:
            return self._arena
        dev, total = tr.dev, tr.total
        flat_g = torch.zeros(total, dtype=torch.float32, device=dev)
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        views, mviews, vviews, steps = [], [], [], []
        self._live = set()                                        # indices (arena order) of the parameters that have had a gradient = have state, as in torch
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(tr.params, tr.offsets)):
                n = p.numel()
                mv, vv = m[off:off + n].view(p.shape), v[off:off + n].view(p.shape)
                st = self.state.get(p)
                if st is not None and "exp_avg" in st:            # state made by torch's path / a loaded checkpoint moves into the arenas
                    mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                    steps.append(int(float(st["step"])))
                    st["exp_avg"], st["exp_avg_sq"] = mv, vv
                    self._live.add(i)
                views.append(flat_g[off:off + n].view(p.shape)); mviews.append(mv); vviews.append(vv)
        self._mv = (mviews, vviews)
        if steps:
            if min(steps) != max(steps):
                return None                                       # parameters with different histories: torch's per-tensor path
            self._fused_steps = steps[0]
        self._arena = (tr, flat_g, views, m, v)
        return self._arena

    def _sync_steps(self) -> None:
        if self._arena is not None:
            ps = self._arena[0].params
            for i in self._live:
                self.state[ps[i]]["step"] = torch.tensor(float(self._fused_steps))

    # ---- torch.optim.Optimizer interface --------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        arena = self._bind()
        grads = None
        idx = None
        if arena is not None:
            tr = arena[0]
            idx = [i for i, p in enumerate(tr.params) if p.grad is not None]
            grads = [tr.params[i].grad for i in idx]
            # torch SKIPS a parameter without a gradient (no state, no update).  The arena launch gives the same result for a parameter that has
            # never had one — its gradient slice and its moments are zero, the update is 0 / (0 + eps) — as long as there is no weight decay; a
            # parameter that HAS moments and misses a gradient would get an update from them here and none from torch: the per-tensor path then
            if (any(g.is_sparse or g.dtype != torch.float32 or g.device != tr.dev for g in grads)
                    or (len(idx) < len(tr.params) and (float(self.param_groups[0]["weight_decay"]) != 0.0 or not self._live.issubset(idx)))):
                arena = None
        if arena is None:
            self._sync_steps()
            super().step()
            # torch's path may have made state of its own (a parameter's first gradient) and has moved the step counters: the next fused step
            # re-homes whatever the state holds now into fresh arenas (_bind)
            self._arena = None
            return loss
        tr, flat_g, views, m, v = arena
        g = self.param_groups[0]
        torch._foreach_copy_([views[i] for i in idx] if len(idx) < len(views) else views, grads)
        for i in idx:
            if i not in self._live:                               # first gradient of this parameter: it gets its state entry, as torch would make it
                self._live.add(i)
                st = self.state[tr.params[i]]
                st["step"], st["exp_avg"], st["exp_avg_sq"] = torch.tensor(float(self._fused_steps)), self._mv[0][i], self._mv[1][i]
        self._fused_steps += 1
        kn.adam_step(tr.flat_p, flat_g, m, v, tr.flat_bf16, tr.total, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                     float(g["eps"]), float(g["weight_decay"]), self._fused_steps, grad_scale=1.0, step_state_dev=None,
                     lo=tr.flat_lo, lo_ranges=tr.lo_ranges)
        if tr.tiles_t is not None or tr.conv_table is not None:
            kn.derive_copies(tr.flat_bf16, tr.flat_bf16_t, tr.tiles_t, tr.flat_p, tr.conv_shadow, tr.conv_table)
        if tr.frag_idx is not None or tr.lo_frag_idx is not None:
            kn.gather_chunks2(tr.flat_bf16, tr.flat_bf16_t, tr.frag_shadow, tr.frag_idx, tr.flat_lo, tr.lo_frag, tr.lo_frag_idx)
        shadow.bump_epoch()
        # the kernel wrote the arena directly: the parameters' version counters did not move, so the keeper's staleness check (sum of the
        # versions) sees nothing to refresh — which is right, its copies came out of the same launches
        self.fused_launches += 1
        return loss

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._arena = None                                        # the loaded tensors are re-homed into the arenas by the next step()
