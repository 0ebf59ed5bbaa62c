"""hulc2_amd.optim.Adam — torch.optim.Adam for a model whose parameters live in the fp32 arena of hulc2_amd's weight keeper.

reference: hulc2/models/hulc2.py:185-198 (`configure_optimizers` instantiates `optimizer._target_`, conf/model/optimizer/adam.yaml:
`torch.optim.Adam`, lr 2e-4).  Swapping that one class path keeps the update rule, the hyper-parameters and the `state_dict()` layout
(per parameter `step` / `exp_avg` / `exp_avg_sq`: a Lightning checkpoint of either optimizer loads into the other) and replaces the step
itself: torch's multi-tensor Adam walks 212 tensors (~1.5 ms of host time per step, the largest single item of the eager loop's host
budget, tools/eager_profile.py) and leaves the kernel-side weight copies stale, so the keeper re-derives them before the next forward
(five launches over the whole arena); here ONE launch of the arena Adam kernel updates parameters and moments and writes the bf16 shadow
and the split operands' remainders, two more derive the transposed / repacked copies — exactly what ArenaTrainer.optimizer_step launches.
The gradients are read where the step node left them (hulc2_amd/stepnode.py: the keeper's gradient arena, `p.grad` = its views); a
gradient that lives elsewhere is copied into its slice first (one `_foreach_copy_`).

Under `torch.amp.GradScaler` (the reference trains with `precision: 16`, conf/trainer/play_trainer.yaml:3) the optimizer declares
`_step_supports_amp_scaling`: `scaler.step(optimizer)` then hands over its device scalars (`grad_scale`, `found_inf`) instead of
synchronising on `found_inf.item()`, and the Adam kernel applies them itself — gradients multiplied by 1 / scale, the whole step skipped
(parameters, moments, step count) when an inf / NaN was found, as torch's fused Adam does.  The host never waits for the GPU inside a step.

Whenever the fused form does not apply — parameters not (yet) in an arena, a parameter that has state and no gradient, a parameter whose
FIRST gradient arrives after the others have stepped (torch starts its step count at 1 then), amsgrad / maximize / decoupled weight decay /
`fused=True`, several parameter groups, CPU — `step()` is torch.optim.Adam.step(), on the same state tensors."""
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
        self._fused_steps = 0        # the step count of the fused path as the HOST knows it (a GradScaler may skip steps on the device: _steps())
        self._dev_steps = None       # device words {unused, step count}: the kernel's bias correction reads the count from here
        self._skippable = False      # a GradScaler's found_inf has been handed in since the host count was last read back
        self.fused_launches = 0      # (tests / logging: steps taken by the fused path)
        # torch.amp.GradScaler.step: hand `grad_scale` / `found_inf` over as attributes instead of unscaling and synchronising itself
        self._step_supports_amp_scaling = True

    # ---- arena binding -----------------------------------------------------------------------------------------------------------------
    def _release(self) -> None:
        """leave the arenas: the fused step count goes back into the per-parameter `step` tensors FIRST (torch's path and a later re-bind
        read them), the moments stay where they are (state entries hold the views)"""
        if self._arena is not None:
            self._sync_steps()
            self._arena = None

    def _bind(self) -> Optional[tuple]:
        if len(self.param_groups) != 1:
            self._release()
            return None
        g = self.param_groups[0]
        if (g.get("amsgrad") or g.get("maximize") or g.get("differentiable") or g.get("capturable") or g.get("decoupled_weight_decay")
                or g.get("fused")):
            self._release()
            return None
        params = [p for p in g["params"] if p.requires_grad]     # (frozen parameters never get a gradient: torch skips them, the arena does not hold them)
        tr = arena_of(params)
        if tr is None:
            self._release()
            return None
        if self._arena is not None and self._arena[0] is tr:
            return self._arena
        self._release()                                           # (another arena took the parameters over: the old one's step count is written back)
        steps = []
        for p in tr.params:
            st = self.state.get(p)
            if st is not None and "exp_avg" in st:
                steps.append(int(float(st["step"])))
        if steps and min(steps) != max(steps):
            # parameters with different histories: torch's per-tensor path.  Remembered by the histories themselves, so that the (arena-sized)
            # re-homing below is not attempted again on every step while nothing has changed
            return None
        dev, total = tr.dev, tr.total
        # gradients: the keeper's own gradient arena when it has one (the step node writes them there: no copy), else a buffer of this optimizer
        flat_g = tr.flat_g if tr.flat_g.numel() == total else torch.zeros(total, dtype=torch.float32, device=dev)
        m = torch.zeros(total, dtype=torch.float32, device=dev)
        v = torch.zeros(total, dtype=torch.float32, device=dev)
        views, mviews, vviews = [], [], []
        self._live = set()                                        # indices (arena order) of the parameters that have had a gradient = have state, as in torch
        with torch.no_grad():
            for i, (p, off) in enumerate(zip(tr.params, tr.offsets)):
                n = p.numel()
                mv, vv = m[off:off + n].view(p.shape), v[off:off + n].view(p.shape)
                st = self.state.get(p)
                if st is not None and "exp_avg" in st:            # state made by torch's path / a loaded checkpoint moves into the arenas
                    mv.copy_(st["exp_avg"]); vv.copy_(st["exp_avg_sq"])
                    st["exp_avg"], st["exp_avg_sq"] = mv, vv
                    self._live.add(i)
                views.append(flat_g[off:off + n].view(p.shape)); mviews.append(mv); vviews.append(vv)
        self._mv = (mviews, vviews)
        self._fused_steps = steps[0] if steps else 0
        self._dev_steps = torch.tensor([0, self._fused_steps], dtype=torch.int64, device=dev)
        self._skippable = False
        self._arena = (tr, flat_g, views, m, v)
        return self._arena

    def _steps(self) -> int:
        """the fused path's step count; read back from the device when a GradScaler may have skipped steps there (one synchronisation, only
        where somebody asks: state_dict(), leaving the fused path)"""
        if self._skippable and self._dev_steps is not None:
            self._fused_steps = int(self._dev_steps[1].item())
            self._skippable = False
        return self._fused_steps

    def _sync_steps(self) -> None:
        if self._arena is not None:
            ps = self._arena[0].params
            n = self._steps()
            for i in self._live:
                self.state[ps[i]]["step"] = torch.tensor(float(n))

    # ---- torch.optim.Optimizer interface --------------------------------------------------------------------------------------------------
    def _torch_step(self):
        """torch.optim.Adam.step() on the same state.  A GradScaler's device scalars (handed over because this class supports them) are applied
        the way the scaler itself would have: unscale + inf check over the gradients, one synchronisation, skip on inf."""
        found_inf, grad_scale = getattr(self, "found_inf", None), getattr(self, "grad_scale", None)
        if any(g.get("fused") for g in self.param_groups):       # torch's own fused kernels take the scaler's scalars themselves
            return super().step()
        if found_inf is not None or grad_scale is not None:
            grads = [p.grad for g in self.param_groups for p in g["params"] if p.grad is not None]
            if grad_scale is not None and grads:
                inv = grad_scale.double().reciprocal().float()
                fi = found_inf if found_inf is not None else torch.zeros(1, dtype=torch.float32, device=inv.device)
                by_dev = {}
                for t in grads:
                    by_dev.setdefault((t.device, t.dtype), []).append(t)
                for (d, _), ts in by_dev.items():
                    torch._amp_foreach_non_finite_check_and_unscale_(ts, fi.to(d), inv.to(d))
            self.found_inf = self.grad_scale = None             # (torch's foreach path refuses the attributes; GradScaler deletes them afterwards)
            if found_inf is not None and float(found_inf.item()) != 0.0:
                return
        super().step()

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
            # parameter that HAS moments and misses a gradient would get an update from them here and none from torch, and one whose FIRST
            # gradient arrives after the others have stepped starts at step 1 in torch (its own bias correction): the per-tensor path then
            late = self._fused_steps > 0 and any(i not in self._live for i in idx)
            if (late or any(g.is_sparse or g.dtype != torch.float32 or g.device != tr.dev for g in grads)
                    or (len(idx) < len(tr.params) and (float(self.param_groups[0]["weight_decay"]) != 0.0 or not self._live.issubset(idx)))):
                arena = None
        if arena is None:
            self._release()
            self._torch_step()
            # torch's path may have made state of its own (a parameter's first gradient) and has moved the step counters: the next fused step
            # re-homes whatever the state holds now into fresh arenas (_bind)
            return loss
        tr, flat_g, views, m, v = arena
        g = self.param_groups[0]
        base = flat_g.data_ptr()
        away = [k for k, i in enumerate(idx) if grads[k].data_ptr() != base + 4 * tr.offsets[i] or not grads[k].is_contiguous()]
        if away:                                                  # gradients that are not already views of the gradient arena
            torch._foreach_copy_([views[idx[k]] for k in away], [grads[k] for k in away])
        if len(idx) < len(views) and flat_g is tr.flat_g:         # (a shared arena may hold an older pass's values in a slice nobody wrote this time)
            have = set(idx)
            for i in range(len(views)):
                if i not in have:
                    views[i].zero_()
        for i in idx:
            if i not in self._live:                               # first gradient of this parameter: it gets its state entry, as torch would make it
                self._live.add(i)
                st = self.state[tr.params[i]]
                st["step"], st["exp_avg"], st["exp_avg_sq"] = torch.tensor(float(self._fused_steps)), self._mv[0][i], self._mv[1][i]
        found_inf, grad_scale = getattr(self, "found_inf", None), getattr(self, "grad_scale", None)
        if found_inf is not None:
            found_inf = found_inf.reshape(1).to(device=tr.dev, dtype=torch.float32)
            self._skippable = True
        if grad_scale is not None:
            grad_scale = grad_scale.reshape(1).to(device=tr.dev, dtype=torch.float32)
        self._fused_steps += 1
        kn.step_count_advance_if(self._dev_steps, found_inf)
        kn.adam_step(tr.flat_p, flat_g, m, v, tr.flat_bf16, tr.total, float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]),
                     float(g["eps"]), float(g["weight_decay"]), self._fused_steps, grad_scale=1.0, step_state_dev=self._dev_steps,
                     lo=tr.flat_lo, lo_ranges=tr.lo_ranges, loss_scale_dev=grad_scale, found_inf_dev=found_inf)
        if tr.tiles_t is not None or tr.conv_table is not None:
            kn.derive_copies(tr.flat_bf16, tr.flat_bf16_t, tr.tiles_t, tr.flat_p, tr.conv_shadow, tr.conv_table)
        if tr.frag_idx is not None or tr.lo_frag_idx is not None:
            kn.gather_chunks2(tr.flat_bf16, tr.flat_bf16_t, tr.frag_shadow, tr.frag_idx, tr.flat_lo, tr.lo_frag, tr.lo_frag_idx)
        shadow.bump_epoch()
        # the kernel wrote the arena directly: the parameters' version counters did not move, so the keeper's staleness check (sum of the
        # versions) sees nothing to refresh — which is right, its copies came out of the same launches
        self.fused_launches += 1
        return loss

    def zero_grad(self, set_to_none: bool = True) -> None:
        """torch.optim.Optimizer.zero_grad.  `set_to_none=False` — the default of the torch 1.12 the reference pins, and what its Lightning calls
        between training_step and backward — is one launch per gradient in torch (106 launches, 0.38 ms of GPU time and 0.47 ms of host time
        per step, tools/study/zero_in_place_cost.py); when the gradients are the views of the keeper's gradient arena it is ONE fill of the
        arena here.  The keeper is told which version of the arena is known to be all zeros: the step node's backward then has nothing to keep
        and add back (stepnode._take_live_grads) as long as no torch operation has written the arena in between."""
        arena = self._arena
        if set_to_none or arena is None or arena[1] is not arena[0].flat_g:
            return super().zero_grad(set_to_none=set_to_none)
        tr, flat_g = arena[0], arena[1]
        lo = flat_g.data_ptr()
        hi = lo + 4 * flat_g.numel()
        inside = False
        for grp in self.param_groups:
            for p in grp["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.grad_fn is not None:
                    g.detach_()
                else:
                    g.requires_grad_(False)
                if not g.is_sparse and lo <= g.data_ptr() < hi:
                    inside = True
                else:
                    g.zero_()                                     # a gradient that lives elsewhere (DDP's bucket views, a user's tensor): as torch does
        if inside:
            flat_g.zero_()                                        # (slices nobody's `.grad` points at are scratch: step() fills or zeroes them itself)
            tr.grads_zeroed_at = flat_g._version                  # views share the arena's version counter: any in-place torch op on one moves it

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._arena = None                                        # the loaded tensors are re-homed into the arenas by the next step()
