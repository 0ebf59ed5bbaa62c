"""Kernel-side views of the parameters: layout repacks and bf16 shadows.

The checkpoint contract keeps parameters fp32 in the reference's layouts (OIHW convs, [out][in] linears).
The MFMA kernels want (a) conv weights with the k axis ordered like the NHWC gather and (b) in bf16
compute mode, bf16 operands so weight panels stream at half the HBM/L2 bytes.  Shadows are refreshed
lazily when a parameter's version counter changes (any optimizer works); the fused Adam kernel writes
the bf16 arena directly and marks shadows fresh (hulc2_amd/trainer.py).
"""
import weakref
from typing import Dict, Optional, Tuple

import torch

from . import kernels as kn
from .gradsink import resolve as _resolve

_cache: Dict[Tuple[int, Optional[str], int], Tuple[int, torch.Tensor, "weakref.ref"]] = {}
_arena: Dict[int, Tuple["weakref.ref", torch.Tensor]] = {}   # id(param) -> (weakref(param), bf16 view into the shadow arena)
_arena_t: Dict[int, Tuple["weakref.ref", torch.Tensor]] = {}  # id(param) -> (weakref(param), bf16 W^T view, refreshed by the trainer)
_epoch = 0                                # bumped by optimizers that update parameters through raw pointers


def clear() -> None:
    _cache.clear()
    _arena.clear()
    _arena_t.clear()
    _arena_layout.clear()


def bump_epoch() -> None:
    """Invalidate every cached repack/shadow (called after an in-place kernel update of the parameters,
    which does not touch torch's version counters)."""
    global _epoch
    _epoch += 1


def epoch() -> int:
    """the counter bump_epoch() moves: part of the key of caches that hold copies of TRAINABLE tensors outside this module"""
    return _epoch


def register_arena_view(param: torch.Tensor, view_bf16: torch.Tensor) -> None:
    """The native trainer keeps one flat bf16 arena that the Adam kernel refreshes in place."""
    key = id(param)
    _arena[key] = (weakref.ref(param, lambda _r, k=key: _arena.pop(k, None)), view_bf16)


def register_arena_view_t(param: torch.Tensor, view_bf16_t: torch.Tensor) -> None:
    """transposed bf16 shadow (cols, rows) kept fresh by the trainer (one hulc_transpose_bf16_tiles launch per step)"""
    key = id(param)
    _arena_t[key] = (weakref.ref(param, lambda _r, k=key: _arena_t.pop(k, None)), view_bf16_t)


_arena_layout: Dict[Tuple[int, str], Tuple["weakref.ref", torch.Tensor]] = {}   # (id(param), layout) -> bf16 repack kept fresh by the trainer


def register_layout_view(param: torch.Tensor, layout: str, view_bf16: torch.Tensor) -> None:
    """conv-weight repack (oihw_flat / ohwi / ihwo) refreshed by the trainer's one-launch hulc_repack_conv_weights"""
    key = (id(param), layout)
    _arena_layout[key] = (weakref.ref(param, lambda _r, k=key: _arena_layout.pop(k, None)), view_bf16)


_ffn_perm: Dict[Tuple[int, int, str], torch.Tensor] = {}


def ffn_frag_perm(n: int, ff: int, device) -> torch.Tensor:
    """int64 device tensor: source element of every element of the fragment-packed array (include/hulc2_amd.h hulc_ffn_frag_perm)"""
    key = (n, ff, str(device))
    t = _ffn_perm.get(key)
    if t is None:
        t = _ffn_perm[key] = torch.from_numpy(kn.ffn_frag_perm(n, ff)).to(device=device, dtype=torch.long)
    return t


def _layout(w: torch.Tensor, layout: Optional[str], chw=None) -> torch.Tensor:
    if layout is None:
        return w
    if layout == "oihw_flat":          # conv1: k = (c, kh, kw) — the parameter itself, flattened
        return w.reshape(w.shape[0], -1)
    if layout == "ohwi":               # NHWC forward: k = (kh, kw, c)
        return w.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    if layout == "ohwi_c8":            # the ResNet stem on NHWC-8 frames: input channels zero-padded to 8, k = (kh, kw, c)
        w8 = w.new_zeros((w.shape[0], 8) + tuple(w.shape[2:]))
        w8[:, :w.shape[1]] = w
        return w8.permute(0, 2, 3, 1).reshape(w.shape[0], -1)
    if layout == "ihwo":               # data gradient: rows = input channel, k = (kh, kw, cout)
        return w.permute(1, 2, 3, 0)
    if layout == "t":                  # transposed 2-D weight: lets dX = dY W stream W k-major
        return w.t()
    if layout == "lo" or layout.endswith("_lo"):   # rounding remainder w - bf16(w) (the caller casts it to bf16): second half of a split operand
        rem = w - w.to(torch.bfloat16).to(torch.float32)
        return rem if layout == "lo" else _layout(rem, layout[:-3], chw)
    if layout.startswith("ffn_p"):     # fragment-packed feed-forward weights of the transformer block launch (hulc_ffn_frag_perm layouts 0..3)
        n = int(layout[-1])
        src = w if n in (0, 1) else w.t()                   # 2: W2^T [FF][128], 3: W1^T [128][FF]
        ff = w.shape[0] if n in (0, 3) else w.shape[1]
        return src.contiguous().reshape(-1)[ffn_frag_perm(n, ff, w.device)]
    if layout in ("hwc", "hwc_t"):     # Linear behind nn.Flatten of a (C, H, W) map: columns reordered to the NHWC activation's (h, w, c)
        c, h, w_ = chw
        m = w.view(w.shape[0], c, h, w_).permute(0, 2, 3, 1).reshape(w.shape[0], -1)
        return m if layout == "hwc" else m.t()
    raise ValueError(layout)


@torch.no_grad()
def weight_operand(w: torch.Tensor, layout: Optional[str] = None, chw: Optional[Tuple[int, int, int]] = None) -> torch.Tensor:
    """Tensor handed to the kernels for parameter `w`: fp32 (repacked if asked) in fp32 compute mode, a
    cached bf16 copy in bf16 mode."""
    w = _resolve(w)                      # (a detached leaf alias of a parameter, installed while the step node captures: hulc2_amd/gradsink.py)
    bf16 = kn.get_compute() == "bf16"
    base = w.detach()
    if not bf16:
        if layout in (None, "oihw_flat"):
            return _layout(base, layout) if layout else base
    elif layout is None or layout == "t":
        hit = (_arena if layout is None else _arena_t).get(id(w))
        if hit is not None and hit[0]() is w:       # ids are recycled: trust the entry only for the very same tensor object
            return hit[1]
    if bf16 and layout is not None:
        hit = _arena_layout.get((id(w), layout))
        if hit is not None and hit[0]() is w:
            return hit[1]
    key = (id(w), layout, int(bf16))
    ver = (w._version, _epoch)
    hit = _cache.get(key)
    if hit is not None and hit[0] == ver and hit[2]() is w and hit[1].device == w.device:
        return hit[1]
    src = _layout(base, layout, chw).contiguous()
    if bf16:
        out = torch.empty(src.shape, dtype=torch.bfloat16, device=src.device)
        kn.cast_f32_to_bf16(src, out, src.numel())
    else:
        out = src
    _cache[key] = (ver, out, weakref.ref(w, lambda _r, k=key: _cache.pop(k, None)))
    return out
