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

_sinks: Dict[int, Tuple["weakref.ref", torch.Tensor]] = {}


def clear() -> None:
    _sinks.clear()


def unregister(keys) -> None:
    """drop the sinks a trainer registered (its own keys only: another live trainer's sinks — an affordance model next to the policy — stay)"""
    for k in keys:
        _sinks.pop(k, None)


def register(param: torch.Tensor, grad_view: torch.Tensor) -> int:
    key = id(param)
    _sinks[key] = (weakref.ref(param, lambda _r, k=key: _sinks.pop(k, None)), grad_view)
    return key


def get(param: torch.Tensor) -> Optional[torch.Tensor]:
    hit = _sinks.get(id(param))
    if hit is not None and hit[0]() is param:
        return hit[1]
    return None


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
    k = id(param)
    first = k not in _written
    _written.add(k)
    return first and _overwrite


def written(param: torch.Tensor) -> bool:
    return id(param) in _written


def written_ids() -> frozenset:
    return frozenset(_written)
