"""Direct-to-arena weight gradients.

torch autograd hands every weight gradient to an AccumulateGrad node that does `param.grad += g`: one extra
elementwise kernel and a 3x (read, read, write) pass over each of the 47 M gradient elements per contribution.
When the native trainer owns a flat gradient arena it registers each parameter's arena slice here; the backward
kernels (wgrad GEMM epilogue `accumulate`, column-sum kernels) then add straight into the slice and the autograd
Function returns None for that parameter.  Only enabled when no per-parameter all-reduce hooks depend on
AccumulateGrad (single GPU, or graph mode where the arena is reduced in one piece).
"""
import weakref
from typing import Dict, Optional, Tuple

import torch

_sinks: Dict[int, Tuple["weakref.ref", torch.Tensor, object]] = {}
# Sinks registered with an `owner` are visible only while that owner is active (`with gradsink.active(owner):`).  The weight keeper of an
# externally optimized model (ArenaTrainer(shadows_only=True, step_node=True)) registers its sinks that way: they exist for the forward and
# backward the step node runs (hulc2_amd/stepnode.py) and for nothing else — a user's own forward / backward through parts of the model
# (validation with gradients, a probe of one encoder) gets its gradients from autograd as always.
_active_owners: Dict[int, int] = {}


def clear() -> None:
    _sinks.clear()


def unregister(keys) -> None:
    """drop the sinks a trainer registered (its own keys only: another live trainer's sinks — an affordance model next to the policy — stay)"""
    for k in keys:
        _sinks.pop(k, None)


# Aliases: while the step node captures its graphs the modules hold detached leaf aliases of the parameters (same memory, their own autograd
# leaves: the AccumulateGrad nodes of the REAL parameters live on the caller's stream and would pull it into the capture).  Everything keyed
# by the identity of a parameter — these sinks, shadow.weight_operand — resolves an alias to its parameter first.
_alias: Dict[int, Tuple[torch.Tensor, torch.Tensor]] = {}


def set_aliases(aliases, params) -> None:
    _alias.clear()
    for a, p in zip(aliases, params):
        _alias[id(a)] = (a, p)


def clear_aliases() -> None:
    _alias.clear()


def resolve(t: torch.Tensor) -> torch.Tensor:
    if _alias:
        hit = _alias.get(id(t))
        if hit is not None and hit[0] is t:
            return hit[1]
    return t


def register(param: torch.Tensor, grad_view: torch.Tensor, owner=None) -> int:
    key = id(param)
    _sinks[key] = (weakref.ref(param, lambda _r, k=key: _sinks.pop(k, None)), grad_view, None if owner is None else id(owner))
    return key


def get(param: torch.Tensor) -> Optional[torch.Tensor]:
    param = resolve(param)
    hit = _sinks.get(id(param))
    if hit is not None and hit[0]() is param and (hit[2] is None or hit[2] in _active_owners):
        return hit[1]
    return None


class active:
    """`with gradsink.active(owner):` — the sinks registered with this owner take gradients inside the block (re-entrant; the backward kernels
    of one pass run on autograd's worker thread, strictly one after the other, so a process-wide table is enough)"""

    def __init__(self, owner):
        self.key = id(owner)

    def __enter__(self):
        _active_owners[self.key] = _active_owners.get(self.key, 0) + 1
        return self

    def __exit__(self, *exc):
        n = _active_owners.get(self.key, 0) - 1
        if n <= 0:
            _active_owners.pop(self.key, None)
        else:
            _active_owners[self.key] = n
        return False


# ---- first write of a step overwrites --------------------------------------------------------------------------------------------
# A sink that is written exactly once per step does not need a zeroed arena slice nor a read-modify-write epilogue: the trainer turns
# `overwrite mode` on once it has seen (in a first, fully zeroed step) which sinks the backward kernels write, stops zeroing those slices
# and the writers ask first_write() whether they are the first contribution of the step.
_written = set()
_overwrite = False


def begin_step(overwrite: bool) -> None:
    global _overwrite
    _written.clear()
    _overwrite = bool(overwrite)


def first_write(param: torch.Tensor) -> bool:
    """marks the sink of `param` as written in this step; True when the caller may OVERWRITE it (first write, overwrite mode on)"""
    k = id(resolve(param))
    first = k not in _written
    _written.add(k)
    return first and _overwrite


def written(param: torch.Tensor) -> bool:
    return id(resolve(param)) in _written


def written_ids() -> frozenset:
    return frozenset(_written)
