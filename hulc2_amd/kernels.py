"""Thin typed wrappers over the C ABI (include/hulc2_amd.h): tensors in, raw pointers out.

Everything here takes device tensors, extracts `data_ptr()` / strides and calls libhulc2_amd.so on
the current torch stream.  No arithmetic happens in Python.
"""
import ctypes
from typing import Optional

import torch

from . import lib as _L

F32, BF16, F16 = 0, 1, 2
_COMPUTE = {"bf16": BF16, "fp32": F32, "f32": F32}
_compute_mode = BF16


_base_mode = "bf16"         # what set_compute selected: "bf16" | "fp32" | "mixed"
_bwd_mode = None            # compute mode autograd Functions created now will run their BACKWARD in (None: the current mode)


def set_compute(mode: str) -> None:
    """Select the arithmetic of a step:
      'bf16'  v_mfma_f32_32x32x16_bf16, fp32 accumulate (the benchmarked mode; the contrastive head alone runs exact fp32, see fp32_sites)
      'fp32'  exact fp32 MFMA everywhere (parity / debug mode)
      'mixed' FORWARD exact fp32 for everything upstream of the contrastive head (camera encoders, goal encoders, prior, posterior), BACKWARD
              and the recurrent decoder bf16 — the gradients of a bf16 step are only as good as its forward activations (DESIGN §5:
              backward rounding costs < 1 %, forward rounding 5-25 % on this cancellation-prone loss), so the precision goes where it pays."""
    global _compute_mode, _base_mode, _bwd_mode
    if mode not in ("bf16", "fp32", "mixed"):
        raise ValueError(mode)
    _base_mode, _bwd_mode = mode, None
    _compute_mode = _COMPUTE["bf16" if mode == "mixed" else mode]


def get_compute() -> str:
    """the MFMA arithmetic of launches issued now: 'bf16' or 'fp32'"""
    return "bf16" if _compute_mode == BF16 else "fp32"


def base_mode() -> str:
    return _base_mode


def backward_compute() -> str:
    """the mode an autograd Function created now runs its backward in"""
    return _bwd_mode if _bwd_mode is not None else get_compute()


class compute_scope:
    """`with compute_scope("fp32"):` — the MFMA arithmetic of the launches issued inside (selective precision).  Every autograd
    Function of functional.py records backward_compute() in its forward and re-enters that mode for its backward: the scope's own mode, or —
    `fwd_only=True` — the mode that was current outside the scope (exact forward, bf16 backward)."""

    def __init__(self, mode: str, fwd_only: bool = False):
        self.mode, self.fwd_only = _COMPUTE[mode], fwd_only

    def __enter__(self):
        global _compute_mode, _bwd_mode
        self.prev = (_compute_mode, _bwd_mode)
        if self.fwd_only and self.mode != _compute_mode:
            _bwd_mode = backward_compute()
        elif not self.fwd_only:
            _bwd_mode = None
        _compute_mode = self.mode
        return self

    def __exit__(self, *exc):
        global _compute_mode, _bwd_mode
        _compute_mode, _bwd_mode = self.prev
        return False


def fp32_sites() -> frozenset:
    """Which parts of a bf16-mode step run their FORWARD in exact fp32 (HULC_FP32_SITES, comma separated).  Sites: `head` = plan
    recognition's 128 -> 4096 projection of the pooled feature, ProjVisLang, the CLIP loss;  `goal` = the language goal encoder (in a bf16 step; the visual one is not upstream of the contrastive head and stays bf16);  `encfc` = the fc
    tails of the camera encoders (flatten-linear, fc1, fc2);  `pool` = the sequence mean;  `txl` = the posterior's transformer layers;
    `conv1` = conv1 of the camera encoders from split bf16 operands (fp32 frames only);  `a3` = the conv stacks' output map kept in fp32;
    `enc` = the whole camera encoders (exact-fp32 MFMA);  `prior`;  `none`.
    `txl` inside the whole-trunk launch (csrc/txl_block.hip) means split operands — three bf16 MFMAs per product, fp32-class values — not
    the fp32 matrix instruction.
    Default `head,goal,encfc,txl` (measured at the benchmark's size against the fp32 oracle): every gradient within 9.2 %,
    median 4.8 % — closer than the reference's own fp16 autocast (26 % / 6.7 %) — for +0.06-0.10 ms per step over the head alone; round 6:
    `encfc` also makes the gripper camera's conv3 store the EXACT map next to the bf16 one (hulc_conv_desc.y_bf16) for its exact-fp32
    flatten-linear — that operand's bf16 rounding was what held the FORWARD perceptual embeddings at 1.06e-3; with it 9.0e-4, inside
    north_star's 1e-3, at no cost (the cast launch it replaces was dearer);  `a3` adds the static camera's map (an fp16 twin for the spatial
    softmax: + 0.01-0.03 ms per step, the embeddings' maximum error does not move);
    `head,goal,encfc`: 13 % / 8.0 % for +0.09 ms; `head` alone: 23 % / 13 % at no cost; `head,goal,encfc,txl,conv1,a3`: median
    0.84 % (worst 10 %: the conv stacks' own parameters) for +0.45 ms."""
    import os
    v = os.environ.get("HULC_FP32_SITES", "head,goal,encfc,txl")
    sites = frozenset(x for x in v.replace(" ", "").split(",") if x and x != "none")
    unknown = sites - _KNOWN_SITES
    if unknown:
        raise ValueError(f"HULC_FP32_SITES: unknown site(s) {sorted(unknown)}; known: {sorted(_KNOWN_SITES)}")
    return sites


_KNOWN_SITES = frozenset(("head", "goal", "encfc", "txl", "pool", "enc", "prior", "conv1", "a3"))


def site_scope(site: str):
    """the scope a part of the step runs in: exact fp32 when selected by HULC_FP32_SITES (bf16 / mixed modes), exact-fp32 FORWARD for every
    site in 'mixed' mode, else the current mode"""
    if _base_mode != "fp32" and _compute_mode == BF16:
        if site in fp32_sites() or _base_mode == "mixed":
            # forward only: rounding in the BACKWARD products moves no gradient of this model by more than 1 % (tools/study/bf16_emulation.py,
            # `only_bwd`), and the backward then stays on the grouped / chained bf16 launches.  HULC_FP32_SITES_BWD=1: both directions.
            import os
            return compute_scope("fp32", fwd_only=not os.environ.get("HULC_FP32_SITES_BWD"))
    return compute_scope(get_compute(), fwd_only=_bwd_mode is not None)



# ------------------------------------------------------------------------------------------------
# optional per-launch timing (bench.py's roofline leg): HIP events recorded on the stream the kernels are
# launched on (torch's current stream), collected per (entry point, shape key)
# ------------------------------------------------------------------------------------------------
_timing = None


def start_timing() -> None:
    global _timing
    _timing = []


def stop_timing():
    """-> {key: (launches, total_ms, flops_per_launch, bytes_per_launch)}; synchronises the device.  flops / bytes are the
    ALGORITHMIC work of one launch (2 x MACs; each operand and the result touched once), 0 where not annotated."""
    global _timing
    rec, _timing = _timing or [], None
    torch.cuda.synchronize()
    out = {}
    for key, e0, e1, fl, by in rec:
        n, t, _, _ = out.get(key, (0, 0.0, 0.0, 0.0))
        out[key] = (n + 1, t + e0.elapsed_time(e1), fl, by)
    return out


def _nbytes(*ts) -> float:
    return float(sum(t.numel() * t.element_size() for t in ts if t is not None))


class _Timed:
    __slots__ = ("key", "e0", "flops", "bytes")

    def __init__(self, key, flops=0.0, nbytes=0.0):
        self.key, self.flops, self.bytes = key, flops, nbytes

    def __enter__(self):
        if _timing is not None:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()

    def __exit__(self, *exc):
        if _timing is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            _timing.append((self.key, self.e0, e1, self.flops, self.bytes))


def _dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:       # (ABI 7: the finer twin of a bf16 map — conv forward output next to y_bf16, spatial softmax input)
        return F16
    raise TypeError(f"unsupported dtype {t.dtype}")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream() -> int:
    """the current HIP stream's handle.  Through torch's raw getter (what its own compiled-kernel launchers use): `torch.cuda.current_stream()`
    builds a Stream object per call, ~8 us — times ~130 launches it was 0.5 ms of the eager step's host time (tools/eager_profile.py)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


def _stream_of(dev) -> int:
    """the raw handle of `dev`'s current stream (no device-context switch, no Stream object)"""
    if _raw_stream is not None:
        return _raw_stream(dev.index if dev.index is not None else _raw_device())
    return torch.cuda.current_stream(dev).cuda_stream


def _require_contiguous(**ts) -> None:
    for name, t in ts.items():
        if t is not None and not t.is_contiguous():
            raise _L.HulcKernelError(f"{name} must be contiguous (the kernels take dense layouts; got strides {tuple(t.stride())})")


def _require_cuda(*ts) -> None:
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _L.HulcKernelError(
                "hulc2_amd kernels run on MI355X device memory only; got a CPU tensor "
                "(there is no CPU fallback — use oracle/ for CPU reference values in tests)"
            )


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_scratch = {}
_step_state = {}


def step_state(device) -> torch.Tensor:
    """Device-resident {rng word, optimizer step count} (two uint64 stored as int64).  Every RNG kernel xors word 0
    into its site seed and the fused Adam reads word 1, so a captured hipGraph replays with fresh randomness and
    the right bias correction; `advance_step_state` is the one-thread kernel that moves both forward."""
    t = _step_state.get(device)
    if t is None:
        t = _step_state[device] = torch.tensor([0x243F6A8885A308D3 >> 1, 0], dtype=torch.int64, device=device)
    return t


_rng_fresh = {}
_fault = {}


def fault_word(device) -> torch.Tensor:
    """Sticky device word the barrier kernel sets on a timeout (hulc_rnn_wave_desc.err_sticky).  While it is set the Adam kernel leaves the
    weights alone; `check_faults` turns it into an exception."""
    t = _fault.get(device)
    if t is None:
        t = _fault[device] = torch.zeros(1, dtype=torch.int32, device=device)
    return t


def check_faults(device) -> None:
    """Host check of the fault word (synchronises the device: call it where a sync happens anyway — after a timed region, every N steps).
    Raises HulcKernelError instead of letting a NaN loss propagate."""
    t = _fault.get(device)
    w = int(t.item()) if t is not None else 0
    if w != 0:
        t.zero_()
        for ws in _chain_ws.values():     # a timed-out chain leaves its barrier counters non-zero: clear them where they are (the cached
            ws.zero_()                    # buffers stay valid for eager launches; captured graphs are re-captured by ArenaTrainer._poll_faults)
        who = [n for b, n in ((1, "rnn_wavefront (HULC_NO_RNN_WAVEFRONT=1 selects the per-step GEMM path)"),
                              (2, "mlp_chain (HULC_NO_MLP_CHAIN=1 selects the per-layer GEMM path)"),
                              (4, "txl_block (HULC_TXL_NO_SHARE=1 keeps one workgroup per sequence, HULC_NO_TXL_BLOCK=1 the per-layer launches)")) if w & b]
        raise _L.HulcKernelError(
            "device-wide barrier timed out in " + " and ".join(who) + " — the kernel's 256 workgroups were not all resident (another "
            "kernel / process shared the GPU, or the device is partitioned).  The optimizer update of that step was skipped.")


_cu_count = {}


def device_cu_count(device) -> int:
    n = _cu_count.get(device)
    if n is None:
        n = _cu_count[device] = torch.cuda.get_device_properties(device).multi_processor_count
    return n


def advance_step_state(device, rng: bool = True, step: bool = True) -> None:
    """Walk the RNG word and / or bump the optimizer step count (one 1-thread kernel).  The native trainer advances both at the top of a
    step and marks the RNG word fresh, so the `ensure_fresh_rng` inside Hulc2.training_step is then free."""
    _call("hulc_step_state_advance_words", step_state(device), _i(rng), _i(step))
    if rng:
        _rng_fresh[device] = True


def ensure_fresh_rng(device) -> None:
    """Called by Hulc2.training_step in training mode: every call of the step draws new dropout masks and a new latent-plan sample, whoever
    drives the loop (Lightning + any torch optimizer, or ArenaTrainer).  A trainer that already advanced the word for this step is honoured
    once; the optimizer step count (word 1) is never touched here."""
    if _rng_fresh.pop(device, False):
        return
    _call("hulc_step_state_advance_words", step_state(device), _i(1), _i(0))


def reset_step_state(device, seed: int = 0x243F6A8885A308D3 >> 1, step: int = 0) -> None:
    step_state(device).copy_(torch.tensor([seed, step], dtype=torch.int64))
    _rng_fresh.pop(device, None)



def _gemm_scratch(device) -> torch.Tensor:
    """split-K slab buffer, one per (device, stream): reuse is stream-ordered, concurrent streams never share it"""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    t = _scratch.get(key)
    if t is None:
        t = _scratch[key] = torch.zeros(8 << 20, dtype=torch.float32, device=device)   # 32 MiB; the head holds hulc_gemm's tile counters (zero between launches)
    return t


_side = {}


def side_stream(device) -> "torch.cuda.Stream":
    """second HIP stream used to run independent kernel chains concurrently (the two RNN layers as a wavefront)"""
    import os
    if not os.environ.get("HULC_WAVEFRONT"):            # measured (tools/decoder_bench.py): the two recurrent chains do not overlap
        return torch.cuda.current_stream(device)        # usefully (each launch already fills the chip) -> default: one stream
    s = _side.get(device)
    if s is None:
        s = _side[device] = torch.cuda.Stream(device=device)
    return s


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor=True, b_kmajor=True, bias=None, add=None, ld_add=0,
         mask=None, ld_mask=0, mask_scale=1.0, relu=False, accumulate=False, alpha=1.0, drop_p=0.0,
         drop_seed=0, compute=None, rowsum=None, rowsum_accumulate=False):
    """C[M,N] = epi(alpha * A·B^T); see hulc_gemm in include/hulc2_amd.h."""
    _require_cuda(A, B, C, bias, add, mask)
    d = _L.GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.add = add.data_ptr() if add is not None else None
    d.mask = mask.data_ptr() if mask is not None else None
    d.M, d.N, d.K = M, N, K
    d.lda, d.ldb, d.ldc, d.ld_add, d.ld_mask = lda, ldb, ldc, ld_add, ld_mask
    d.a_dtype, d.b_dtype, d.c_dtype = _dt(A), _dt(B), _dt(C)
    d.add_dtype = _dt(add) if add is not None else F32
    d.mask_dtype = _dt(mask) if mask is not None else F32
    d.a_kmajor, d.b_kmajor = int(a_kmajor), int(b_kmajor)
    d.relu, d.accumulate = int(relu), int(accumulate)
    d.alpha, d.mask_scale, d.drop_p, d.drop_seed = alpha, mask_scale, drop_p, drop_seed
    d.compute = _compute_mode if compute is None else compute
    if bias is not None and bias.dtype != torch.float32:
        raise TypeError("bias must be float32")
    ws = _gemm_scratch(C.device)      # split-K slabs (stream-ordered reuse of one scratch buffer)
    d.ws, d.ws_bytes = ws.data_ptr(), ws.numel() * 4
    d.seed_dev = step_state(C.device).data_ptr() if drop_p > 0.0 else None
    d.rowsum_a, d.rowsum_accumulate = (rowsum.data_ptr() if rowsum is not None else None), int(rowsum_accumulate)
    esz = lambda t: t.element_size()
    gbytes = M * K * esz(A) + N * K * esz(B) + M * N * esz(C) * (2 if accumulate else 1) \
        + (M * N * esz(add) if add is not None and ld_add else 0) + (M * N * esz(mask) if mask is not None else 0)
    with _Timed(("gemm", M, N, K, int(a_kmajor), int(b_kmajor)), 2.0 * M * N * K, float(gbytes)):
        _L.check(_L.load().hulc_gemm(ctypes.byref(d), ctypes.c_void_p(_stream())), "hulc_gemm")
    return C


# ------------------------------------------------------------------------------------------------
# convolutions
# ------------------------------------------------------------------------------------------------
def _conv_desc(N, H, W, Cin, Cout, KH, KW, stride, x_nchw, x_dt, y_dt, w_dt, relu, compute=None):
    d = _L.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.KH, d.KW, d.stride = N, H, W, Cin, Cout, KH, KW, stride
    d.x_nchw, d.x_dtype, d.y_dtype, d.w_dtype, d.relu = int(x_nchw), x_dt, y_dt, w_dt, int(relu)
    d.compute = _compute_mode if compute is None else compute
    return d


def conv_out_hw(H, W, KH, KW, stride):
    return (H - KH) // stride + 1, (W - KW) // stride + 1


def _u8_frames(d, x, aug_shift, aug_pad, frame_index=None):
    """conv1 fed by uint8 NHWC frames: shift / scale / normalise happen while the kernel stages the band.
    frame_index (N,) int32: x is the episode store and batch frame n is store frame frame_index[n]."""
    if x.dtype != torch.uint8:
        if frame_index is not None:
            raise TypeError("frame_index addresses a uint8 NHWC episode store")
        return
    if frame_index is not None:
        if frame_index.dtype != torch.int32 or not frame_index.is_contiguous() or frame_index.numel() != d.N:
            raise TypeError("frame_index must be a contiguous int32 tensor with one store frame number per batch frame")
        d.frame_index = frame_index.data_ptr()
    if aug_shift is not None and (aug_shift.dtype != torch.int32 or not aug_shift.is_contiguous() or aug_shift.numel() != 2 * d.N):
        raise TypeError("aug_shift must be a contiguous int32 (N, 2) tensor of {sx, sy}")
    d.x_u8_nhwc, d.aug_pad = 1, int(aug_pad)
    d.aug_shift = aug_shift.data_ptr() if aug_shift is not None else None


def _second_frames(d, x, x2, N, Cin, H, W):
    """hulc_conv_desc.x2 / n_split: conv1 over two frame tensors (the modalities of a step; fp32 NCHW or uint8 NHWC, both alike) as one launch"""
    if x2 is None:
        return
    _require_cuda(x2)
    if (x.dtype not in (torch.float32, torch.uint8) or x2.dtype != x.dtype or not x2.is_contiguous() or x2.shape[1:] != x.shape[1:]
            or x.shape[0] + x2.shape[0] != N):
        raise TypeError("x2: a second contiguous frame tensor of x's type and geometry (fp32 NCHW or uint8 NHWC); N = frames of x + frames of x2")
    d.x2, d.n_split = x2.data_ptr(), int(x.shape[0])


# hulc_conv_desc.x_slot / x2_slot (ABI 5): while the step node captures its graphs, {address of a graph-input frame tensor: device address of the
# slot that holds its current pointer}.  The conv1 band launches of the capture then read the frames through the slots, and a batch at new
# addresses costs two pointer updates instead of a copy (hulc2_amd/stepnode.py).  None outside a capture: the launches take plain addresses.
_frame_slots = None
_frame_slots_used = set()
_frame_slots_probe = False          # record which frame tensors WOULD be read through slots (the node's warm-up pass), change nothing


def _slot_fields(d, x, x2, H, W, Cin, Cout, KH, stride, x_nchw) -> None:
    if not _frame_slots or not x_nchw or x.dtype != torch.float32 or (Cin, Cout, KH, stride) != (3, 32, 8, 4):
        return
    if W % 4 or (H - 8) % 4 or (W - 8) % 4 or W < 72 or H < 72 or _compute_mode != BF16:      # (what the band kernels take; smaller frames keep plain addresses)
        return
    if _frame_slots_probe:
        if x.data_ptr() in _frame_slots and (x2 is None or x2.data_ptr() in _frame_slots):
            _frame_slots_used.add(x.data_ptr())
            if x2 is not None:
                _frame_slots_used.add(x2.data_ptr())
        return
    a = _frame_slots.get(x.data_ptr())
    b = _frame_slots.get(x2.data_ptr()) if x2 is not None else None
    if a is None or (x2 is not None and b is None):
        return
    d.x_slot = a
    _frame_slots_used.add(x.data_ptr())
    if x2 is not None:
        d.x2_slot = b
        _frame_slots_used.add(x2.data_ptr())


def conv2d_fwd(x, w2d, bias, y, N, H, W, Cin, Cout, KH, KW, stride, x_nchw, relu=True, compute=None, aug_shift=None, aug_pad=0,
               frame_index=None, relu_bits=None, w_lo=None, x2=None, y_bf16=None):
    """y (NHWC) = relu(conv(x, w) + b); w2d is [Cout][K] in the layout's k order (see hulc_conv_desc).  x may be uint8 NHWC frames
    for conv1 (aug_shift (N, 2) int32 {sx, sy} or None, aug_pad: RandomShiftsAug's pad).  y_bf16 (fp32 y only): a bf16 copy of y from the
    same accumulators (hulc_conv_desc.y_bf16)."""
    _require_cuda(x, w2d, bias, y, aug_shift, frame_index)
    _require_contiguous(x=x, w2d=w2d, y=y)
    d = _conv_desc(N, H, W, Cin, Cout, KH, KW, stride, x_nchw, F32 if x.dtype == torch.uint8 else _dt(x), _dt(y), _dt(w2d), relu, compute)
    _u8_frames(d, x, aug_shift, aug_pad, frame_index)
    _second_frames(d, x, x2, N, Cin, H, W)
    if compute is None and frame_index is None:
        _slot_fields(d, x, x2, H, W, Cin, Cout, KH, stride, x_nchw)
    oh, ow = conv_out_hw(H, W, KH, KW, stride)
    if relu_bits is not None:              # ReLU sign planes of y: int32 (N * OH * OW * Cout / 32,), written next to y (hulc_conv_desc.relu_bits)
        _require_cuda(relu_bits)
        if relu_bits.dtype != torch.int32 or not relu_bits.is_contiguous() or relu_bits.numel() != N * oh * ow * (Cout // 32):
            raise TypeError("relu_bits: contiguous int32 tensor of N * OH * OW * Cout / 32 words")
        d.relu_bits = relu_bits.data_ptr()
    if w_lo is not None:                   # conv1, fp32 frames: split-operand products (hulc_conv_desc.w_lo)
        _require_cuda(w_lo)
        if w_lo.dtype != torch.bfloat16 or w_lo.shape != w2d.shape or not w_lo.is_contiguous():
            raise TypeError("w_lo: the bf16 remainders of w2d, same shape")
        d.w_lo = w_lo.data_ptr()
    if y_bf16 is not None:
        _require_cuda(y_bf16)
        if y.dtype not in (torch.float32, torch.float16) or y_bf16.dtype != torch.bfloat16 or y_bf16.shape != y.shape or not y_bf16.is_contiguous():
            raise TypeError("y_bf16: a contiguous bf16 tensor shaped like the fp32 / fp16 output y")
        d.y_bf16 = y_bf16.data_ptr()
    macs = float(N) * oh * ow * Cout * Cin * KH * KW * (3 if w_lo is not None else 1)
    with _Timed(("conv2d_fwd", N, H, W, Cin, Cout, KH, stride), 2 * macs, _nbytes(x, w2d, y, y_bf16) + (_nbytes(x2) if x2 is not None else 0)):
        _L.check(_L.load().hulc_conv2d_fwd(ctypes.byref(d), _p(x), _p(w2d), _p(bias), _p(y), ctypes.c_void_p(_stream())),
                 "hulc_conv2d_fwd")
    return y


def conv2d_padded_fwd(x, w2d, bias, y, N, H, W, Cin, Cout, KH, KW, stride, pad, relu=True, add=None, compute=None):
    """y (NHWC) = [relu](conv_pad(x, w) + bias [+ add]): the convolutions of the frozen ResNet trunk (VisionR3M), BatchNorm folded into
    w2d [Cout][KH*KW*Cin] / bias; add = the residual branch, shaped and typed like y."""
    _require_cuda(x, w2d, bias, y, add)
    _require_contiguous(x=x, w2d=w2d, y=y)
    if add is not None and (add.dtype != y.dtype or add.shape != y.shape or not add.is_contiguous()):
        raise TypeError("conv2d_padded_fwd: add must match y (shape, dtype, contiguous)")
    d = _conv_desc(N, H, W, Cin, Cout, KH, KW, stride, False, _dt(x), _dt(y), _dt(w2d), relu, compute)
    oh, ow = (H + 2 * pad - KH) // stride + 1, (W + 2 * pad - KW) // stride + 1
    macs = float(N) * oh * ow * Cout * Cin * KH * KW
    with _Timed(("conv2d_padded_fwd", N, H, W, Cin, Cout, KH, stride), 2 * macs, _nbytes(x, w2d, y, add)):
        _L.check(_L.load().hulc_conv2d_padded_fwd(ctypes.byref(d), _i(pad), _p(x), _p(w2d), _p(bias), _p(add), _p(y),
                                                  ctypes.c_void_p(_stream())), "hulc_conv2d_padded_fwd")
    return y


def r3m_normalize(x, y, mean3, std3):
    """x fp32 (N,3,H,W) in [0,255] -> y NHWC (N,H,W,8): ((x / 255) - mean) / std in channels 0..2, zeros in 3..7 (r3m's forward)."""
    _require_cuda(x, y)
    _require_contiguous(x=x, y=y)
    if x.dtype != torch.float32 or x.dim() != 4 or x.shape[1] != 3:
        raise TypeError("r3m_normalize: x must be fp32 (N, 3, H, W)")
    n, _, h, w = x.shape
    m = (ctypes.c_float * 3)(*[float(v) for v in mean3])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std3])
    _call("hulc_r3m_normalize", x, _i(n), _i(h), _i(w), m, sd, y, _i(_dt(y)))
    return y


def r3m_packed_width(W: int) -> int:
    return int(_L.load().hulc_r3m_packed_width(int(W)))


def r3m_normalize_packed(x, xp, mean3, std3):
    """x fp32 (N,3,H,W) in [0,255] -> xp bf16 (N, H+6, r3m_packed_width(W), 4): normalised RGB + 0 inside a zero border (the stem's input)."""
    _require_cuda(x, xp)
    _require_contiguous(x=x, xp=xp)
    n, _, h, w = x.shape
    if x.dtype != torch.float32 or xp.dtype != torch.bfloat16 or tuple(xp.shape) != (n, h + 6, r3m_packed_width(w), 4):
        raise TypeError("r3m_normalize_packed: x fp32 (N,3,H,W), xp bf16 (N, H+6, packed width, 4)")
    m = (ctypes.c_float * 3)(*[float(v) for v in mean3])
    sd = (ctypes.c_float * 3)(*[float(v) for v in std3])
    _call("hulc_r3m_normalize_packed", x, _i(n), _i(h), _i(w), m, sd, xp)
    return xp


def r3m_stem_fwd(xp, w, bias, y, N, H, W, Cout, relu=True):
    """the 7x7 stride-2 stem on the packed input: w bf16 (Cout, 7*8*4) = [o][kh][kw][c] with zeros at kw = 7 and c = 3; y NHWC."""
    _require_contiguous(xp=xp, w=w, y=y)
    oh, ow = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    macs = float(N) * oh * ow * Cout * 147
    _call("hulc_r3m_stem_fwd", xp, w, bias, y, _i(_dt(y)), _i(N), _i(H), _i(W), _i(Cout), _i(int(relu)),
          key=("r3m_stem_fwd", N, H, W, Cout), flops=2 * macs, nbytes=_nbytes(xp, w, y))
    return y


def nhwc_bn_train_fwd(z, M, C, gamma, beta, eps, momentum, run_mean, run_var, y, add=None, relu=False, saved=None):
    """nn.BatchNorm2d in training mode over NHWC rows z (M, C) -> y (+ add, ReLU); running statistics updated in place (may be None);
    saved (2, C) fp32 (optional): the batch mean and rstd, what nhwc_bn_train_bwd needs"""
    lib = _L.load()
    lib.hulc_nhwc_bn_train_workspace.restype = ctypes.c_long
    ws = _ws(lib.hulc_nhwc_bn_train_workspace(_l(M), _i(C)), z.device)
    _call("hulc_nhwc_bn_train_fwd_saved", z, _i(_dt(z)), _l(M), _i(C), gamma, beta, _f(eps), _f(momentum), run_mean, run_var, add,
          _i(_dt(add) if add is not None else F32), _i(int(relu)), y, _i(_dt(y)), saved, ws)
    return y


def nhwc_bn_train_bwd(dy, y, z, M, C, gamma, saved, dz, g_out=None, dgamma=None, dbeta=None, accumulate_params=False):
    """backward of nhwc_bn_train_fwd (+ its ReLU when y is given): dz (and g_out = the masked incoming gradient, the shortcut branch's share)
    from dy, the saved fp32 convolution output z and the forward's (mean, rstd); dgamma / dbeta optional (hulc_nhwc_bn_train_bwd)"""
    if z.dtype != torch.float32 or saved.dtype != torch.float32:
        raise TypeError("nhwc_bn_train_bwd: z and saved are fp32")
    if g_out is not None and g_out.dtype != dz.dtype:
        raise TypeError("nhwc_bn_train_bwd: g_out shares dz's dtype")
    _require_contiguous(dy=dy, y=y, z=z, dz=dz, g_out=g_out)
    lib = _L.load()
    lib.hulc_nhwc_bn_train_workspace.restype = ctypes.c_long
    ws = _ws(lib.hulc_nhwc_bn_train_workspace(_l(M), _i(C)), z.device)
    _call("hulc_nhwc_bn_train_bwd", dy, _i(_dt(dy)), y, _i(_dt(y) if y is not None else F32), z, _l(M), _i(C), gamma, saved, dz, g_out, _i(_dt(dz)),
          dgamma, dbeta, _i(int(accumulate_params)), ws)
    return dz


def maxpool_nhwc_bwd(x, dy, dx, N, H, W, C, k, stride, pad):
    """dx of maxpool_nhwc: the gradient goes to the first maximum of every window (nn.MaxPool2d's recorded index); x, dy, dx one dtype"""
    if not (x.dtype == dy.dtype == dx.dtype):
        raise TypeError("maxpool_nhwc_bwd: x, dy, dx share a dtype")
    _require_contiguous(x=x, dy=dy, dx=dx)
    _call("hulc_maxpool_nhwc_bwd", x, dy, _i(_dt(x)), _i(N), _i(H), _i(W), _i(C), _i(k), _i(stride), _i(pad), dx)
    return dx


def nhwc_scatter(x, y, step: int, off: int):
    """y (N, Hy, Wy, C) = x (N, H, W, C) placed at (off + step * row, off + step * col), zeros elsewhere (zero insertion / zero padding)"""
    if x.dtype != y.dtype or x.dim() != 4 or y.dim() != 4 or x.shape[0] != y.shape[0] or x.shape[3] != y.shape[3]:
        raise TypeError("nhwc_scatter: x (N, H, W, C) and y (N, Hy, Wy, C) of one dtype")
    _require_contiguous(x=x, y=y)
    n, h, w, c = x.shape
    _call("hulc_nhwc_scatter", x, _i(_dt(x)), _i(n), _i(h), _i(w), _i(c), _i(y.shape[1]), _i(y.shape[2]), _i(step), _i(off), y)
    return y


def maxpool_nhwc(x, y, N, H, W, C, k, stride, pad):
    _call("hulc_maxpool_nhwc", x, _i(_dt(x)), _i(N), _i(H), _i(W), _i(C), _i(k), _i(stride), _i(pad), y)
    return y


def conv2d_bwd_data(dy, wt, dx, relu_src, N, H, W, Cin, Cout, KH, KW, stride, compute=None, relu_bits=None):
    """dx (NHWC [N][H][W][Cin]) from dy (NHWC); wt = weight as [Cin][KH][KW][Cout].  relu_bits: the sign planes of the layer input
    (int32, N * H * W * Cin / 32 words) — read instead of relu_src where the band kernel takes the launch."""
    _require_cuda(dy, wt, dx, relu_src, relu_bits)
    d = _conv_desc(N, H, W, Cin, Cout, KH, KW, stride, False, _dt(dx), _dt(dy), _dt(wt), False, compute)
    oh, ow = conv_out_hw(H, W, KH, KW, stride)
    if relu_bits is not None:
        if relu_bits.dtype != torch.int32 or not relu_bits.is_contiguous() or relu_bits.numel() != N * H * W * (Cin // 32):
            raise TypeError("relu_bits: contiguous int32 tensor of N * H * W * Cin / 32 words")
        d.relu_bits = relu_bits.data_ptr()
    macs = float(N) * oh * ow * Cout * Cin * KH * KW
    with _Timed(("conv2d_bwd_data", N, H, W, Cin, Cout, KH, stride), 2 * macs,
                _nbytes(dy, wt, dx, relu_bits if relu_bits is not None else relu_src)):
        _L.check(_L.load().hulc_conv2d_bwd_data(ctypes.byref(d), _p(dy), _p(wt), _p(dx), _p(relu_src),
                                                ctypes.c_void_p(_stream())), "hulc_conv2d_bwd_data")
    return dx


def conv2d_bwd_weight(x, dy, dw, db, N, H, W, Cin, Cout, KH, KW, stride, x_nchw, compute=None, dw_oihw=False, accumulate=False,
                      aug_shift=None, aug_pad=0, frame_index=None, x2=None):
    """dw [Cout][K] / db [Cout] (fp32) from x and dy (NHWC).  dw_oihw: dw in the parameter's OIHW order (else the forward k order);
    accumulate: add into dw / db (gradient arena sinks)."""
    _require_cuda(x, dy, dw, db)
    _require_contiguous(x=x, dy=dy, dw=dw)
    lib = _L.load()
    lib.hulc_conv2d_bwd_weight_workspace.restype = ctypes.c_long
    d = _conv_desc(N, H, W, Cin, Cout, KH, KW, stride, x_nchw, F32 if x.dtype == torch.uint8 else _dt(x), _dt(dy), F32, False, compute)
    d.dw_oihw, d.dw_accumulate = int(dw_oihw), int(accumulate)
    _u8_frames(d, x, aug_shift, aug_pad, frame_index)
    _second_frames(d, x, x2, N, Cin, H, W)
    if compute is None and frame_index is None and dy.dtype == torch.bfloat16:
        _slot_fields(d, x, x2, H, W, Cin, Cout, KH, stride, x_nchw)
    nbytes = lib.hulc_conv2d_bwd_weight_workspace(ctypes.byref(d))
    ws = torch.empty(nbytes // 4, dtype=torch.float32, device=x.device)
    oh, ow = conv_out_hw(H, W, KH, KW, stride)
    macs = float(N) * oh * ow * Cout * Cin * KH * KW
    with _Timed(("conv2d_bwd_weight", N, H, W, Cin, Cout, KH, stride), 2 * macs, _nbytes(x, dy, dw, db) + (_nbytes(x2) if x2 is not None else 0)):
        _L.check(lib.hulc_conv2d_bwd_weight(ctypes.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws),
                                            ctypes.c_void_p(_stream())), "hulc_conv2d_bwd_weight")
    return dw, db


# ------------------------------------------------------------------------------------------------
# generic call helper: tensors -> pointers, python numbers -> ctypes by annotation
# ------------------------------------------------------------------------------------------------
_c = ctypes


def _call(name, *args, key=None, flops=0.0, nbytes=0.0):
    lib = _L.load()
    conv = []
    for a in args:
        if isinstance(a, torch.Tensor):
            _require_cuda(a)
            conv.append(_c.c_void_p(a.data_ptr()))
        elif a is None:
            conv.append(_c.c_void_p(0))
        else:
            conv.append(a)
    conv.append(_c.c_void_p(_stream()))
    with _Timed(key or (name,), flops, nbytes):
        _L.check(getattr(lib, name)(*conv), name)


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes) // 4, 1), dtype=torch.float32, device=device)


def _i(v):
    return _c.c_int(int(v))


def _l(v):
    return _c.c_long(int(v))


def _f(v):
    return _c.c_float(float(v))


def _u64(v):
    return _c.c_ulonglong(int(v) & 0xFFFFFFFFFFFFFFFF)


def spatial_softmax_fwd(x, N, HW, C, xmap, ymap, temperature, out, stats):
    _call("hulc_spatial_softmax_fwd", x, _i(_dt(x)), _i(N), _i(HW), _i(C), xmap, ymap, temperature, out, stats)


def spatial_softmax_bwd(x, N, HW, C, xmap, ymap, temperature, out, stats, dout, dx, relu_mask=True):
    _call("hulc_spatial_softmax_bwd", x, _i(_dt(x)), _i(N), _i(HW), _i(C), xmap, ymap, temperature, out, stats, dout, dx,
          _i(_dt(dx)), _i(relu_mask))


def _sd(t, drop_p):
    return step_state(t.device) if drop_p > 0.0 else None


def layernorm_fwd(x, o, drop_p, seed, gamma, beta, eps, R, D, pre_out, y, mean, rstd):
    _call("hulc_layernorm_fwd", x, o, _f(drop_p), _u64(seed), _sd(x, drop_p), gamma, beta, _f(eps), _i(R), _i(D), pre_out, y, mean, rstd)


def layernorm_bwd(dy, pre, mean, rstd, gamma, R, D, dpre, do_out, drop_p, seed, dgamma, dbeta, accumulate_params=False):
    lib = _L.load()
    lib.hulc_layernorm_bwd_workspace.restype = _c.c_long
    ws = _ws(lib.hulc_layernorm_bwd_workspace(_i(R), _i(D)), dy.device)
    _call("hulc_layernorm_bwd", dy, pre, mean, rstd, gamma, _i(R), _i(D), dpre, do_out, _f(drop_p), _u64(seed), _sd(dy, drop_p), dgamma,
          dbeta, _i(accumulate_params), ws)


def colsum(x, M, N, ld, out, accumulate=False):
    lib = _L.load()
    lib.hulc_colsum_workspace.restype = _c.c_long
    ws = _ws(lib.hulc_colsum_workspace(_l(M), _i(N)), x.device)
    _call("hulc_colsum", x, _i(_dt(x)), _l(M), _i(N), _l(ld), out, _i(accumulate), ws, nbytes=float(M) * N * x.element_size())


def seq_mean_fwd(x, y, B, S, D, scale=1.0):
    _call("hulc_seq_mean_fwd", x, y, _i(B), _i(S), _i(D), _f(scale))


def strided_seq_sum(x, y, B, S, D, stride_b, stride_s, ldy, scale=1.0):
    _call("hulc_strided_seq_sum", x, _i(_dt(x)), y, _i(B), _i(S), _i(D), _l(stride_b), _l(stride_s), _l(ldy), _f(scale))


def seq_mean_bwd(dy, dx, B, S, D):
    _call("hulc_seq_mean_bwd", dy, dx, _i(B), _i(S), _i(D))


def add_pos_fwd(x, pos, pos_ids, y, B, S, D, drop_p, seed):
    _call("hulc_add_pos_fwd", x, pos, pos_ids, y, _i(B), _i(S), _i(D), _f(drop_p), _u64(seed), _sd(x, drop_p))


def dropout_bwd(dy, dx, n, drop_p, seed):
    _call("hulc_dropout_bwd", dy, dx, _l(n), _f(drop_p), _u64(seed), _sd(dy, drop_p))


def relu_bwd(dy, y, dx, n, scale=1.0):
    _call("hulc_relu_bwd", dy, y, _i(_dt(y)), dx, _l(n), _f(scale))


def attention_fwd(qkv, out, probs, B, S, H, head_dim, drop_p, seed):
    _call("hulc_attention_fwd", qkv, out, probs, _i(B), _i(S), _i(H), _i(head_dim), _f(drop_p), _u64(seed), _sd(qkv, drop_p))


def attention_bwd(qkv, probs, dout, dqkv, B, S, H, head_dim, drop_p, seed):
    _call("hulc_attention_bwd", qkv, probs, dout, dqkv, _i(B), _i(S), _i(H), _i(head_dim), _f(drop_p), _u64(seed), _sd(qkv, drop_p))


def _mix_desc(T, A, n_mix, num_classes, ld, log_scale_min, gripper_alpha, act_min, act_max, nseg=1, time_major_B=0):
    d = _L.MixDesc()
    d.T, d.A, d.n_mix, d.num_classes, d.ld, d.nseg = T, A, n_mix, num_classes, ld, nseg
    d.time_major_B = int(time_major_B)
    d.log_scale_min, d.gripper_alpha = log_scale_min, gripper_alpha
    _require_cuda(act_min, act_max)
    d.act_min, d.act_max = act_min.data_ptr(), act_max.data_ptr()
    return d


# Kernels with a device-wide barrier (rnn_wavefront) need every workgroup co-resident: they must not share the GPU with
# kernels on other streams (e.g. RCCL all-reduces overlapped with backward).  The trainer declares that situation here and
# the recurrent decoder then takes its per-step GEMM path.
_concurrent_streams = False


def set_concurrent_streams(flag: bool) -> None:
    global _concurrent_streams
    _concurrent_streams = bool(flag)


def concurrent_streams() -> bool:
    return _concurrent_streams


# ---- two cooperative launches side by side (round 6; include/hulc2_amd.h hulc_set_coop_share) -------------------------------------------------
# The prior branch (goal encoders -> plan proposal: persistent MLP chains) and the posterior branch (the transformer trunk) of a step do not
# depend on each other (hulc2/models/hulc2.py:228-233).  Both are cooperative launches that take one workgroup per CU — on two streams (two
# branches of the captured graph) they fit the device only if each keeps to HALF of it.  `coop_share_scope(2)` is where the model forks; the
# autograd Functions of the cooperative launches remember the share they ran under and launch their backward with it.
_coop_share = 1
_branch_streams = {}


def coop_share() -> int:
    return _coop_share


class coop_share_scope:
    def __init__(self, n: int):
        self.n = int(n)

    def __enter__(self):
        global _coop_share
        self.old, _coop_share = _coop_share, self.n

    def __exit__(self, *exc):
        global _coop_share
        _coop_share = self.old


class no_gc:
    """Around a hipGraph capture: cyclic garbage is collected BEFORE it and the collector stays off until it ends.  A collection that starts in
    the middle of a capture (autograd's worker thread allocates Python objects all the time) finalises whatever garbage earlier steps left —
    graphs, pooled tensors, events of other models — and a runtime call from such a finaliser while a capture is open aborts the process
    (seen as `Fatal Python error: Aborted ... Garbage-collecting` inside a captured backward, once in a few runs)."""

    def __enter__(self):
        import gc
        gc.collect()
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()


def fork_branches() -> bool:
    """run the prior and the posterior branch of a step on two streams (HULC_FORK=0: one after the other, as until round 5)"""
    import os
    return os.environ.get("HULC_FORK", "1") != "0" and not concurrent_streams() and _timing is None


_wgrad_branch = False


class wgrad_branch_scope:
    """ArenaTrainer's forward + backward: the recurrent weight gradients may run as a third branch (functional.DecoderRNNFn.backward)"""

    def __enter__(self):
        global _wgrad_branch
        self.old, _wgrad_branch = _wgrad_branch, True

    def __exit__(self, *exc):
        global _wgrad_branch
        _wgrad_branch = self.old


def wgrad_branch_ok() -> bool:
    import os
    return _wgrad_branch and os.environ.get("HULC_WGRAD_FORK", "0") not in ("", "0")


def branch_stream(device, which: int = 0):
    """the stream of a forked branch of the step: 0 the posterior, 1 weight gradients that nothing downstream consumes"""
    st = _branch_streams.get((device, which))
    if st is None:
        st = _branch_streams[(device, which)] = capture_stream(device)      # (distinct from every other cached stream)
    return st


_made_streams = []       # (kept alive: a stream wrapped by ExternalStream is never destroyed by torch)


def capture_stream(device):
    """A stream to capture a graph on that NO earlier capture has used: a brand-new HIP stream (hulc_stream_create: hipStreamCreateWithFlags,
    non-blocking), not one of torch's pooled streams.  torch.cuda.Stream() hands out the 32 streams of its pool round-robin, so in a long
    process the "new" stream of a capture is one an earlier capture ran on — and with forked branches inside the captures (round 6) that ended,
    now and then, in a segfault inside hipGraphLaunch of a step-node backward graph captured later (seen with a cached third branch stream,
    with a pool of four capture streams, and after ~390 tests of the full suite on torch's own pool).  Falls back to a pooled stream that is
    none of the cached branch / side streams if the creation fails."""
    import ctypes
    try:
        h = ctypes.c_void_p()
        with torch.cuda.device(device):
            rc = _L.load().hulc_stream_create(ctypes.byref(h))
        if rc == 0 and h.value:
            st = torch.cuda.ExternalStream(h.value, device=device)
            _made_streams.append(st)
            return st
    except AttributeError:
        pass
    taken = {st.cuda_stream for st in _branch_streams.values()}
    try:
        from .models.perceptual_encoders.concat_encoders import _side_streams
        taken |= {st.cuda_stream for st in _side_streams.values()}
    except Exception:                        # noqa: BLE001
        pass
    for _ in range(64):
        st = torch.cuda.Stream(device=device)
        if st.cuda_stream not in taken:
            return st
    return st


def note_producer_stream(device, stream) -> None:
    """work issued on `stream` writes gradients straight into the arena and feeds no autograd node: the end-of-pass weight-gradient launch
    (wgrad_flush, before the optimizer / the end of a captured backward graph) joins it"""
    _wg_streams.setdefault(device, set()).add(stream.cuda_stream)


def _call_shared(share: int, name, *args, **kw):
    """a cooperative launch issued for 1 / share of the device (host-side setting, read by the launcher; None: the scope's share)"""
    share = _coop_share if share is None else share
    if share <= 1:
        return _call(name, *args, **kw)
    lib = _L.load()
    lib.hulc_set_coop_share(_i(share))
    try:
        _call(name, *args, **kw)
    finally:
        lib.hulc_set_coop_share(_i(1))


def gemm_fuses_rowsum(M: int, a_kmajor: bool) -> bool:
    """hulc_gemm computes rowsum_a (the bias gradient of a weight-gradient GEMM) in the same launch on the tiled path"""
    return M > 64


def mlp_chain_ok(M: int, K0: int, widths, device, share=None) -> bool:
    """shapes hulc_mlp_chain takes (include/hulc2_amd.h) on a whole MI355X with nothing else sharing the GPU — or, under a coop share of n, on
    1 / n of it: a layer of N columns needs N / 16 workgroups"""
    import os
    if os.environ.get("HULC_NO_MLP_CHAIN") or _compute_mode != BF16 or concurrent_streams() or device_cu_count(device) < 256:
        return False
    if not (1 <= M <= 64) or not (1 <= len(widths) <= 8):
        return False
    share = _coop_share if share is None else share
    if share == 0 or (share > 1 and max(widths) > 16 * (256 // share)):     # (0: a branch that runs beside a whole-device cooperative launch)
        return False
    k = K0
    for n in widths:
        if n < 16 or n % 16 or n > 4096 or k < 8 or k % 8 or k > 4096 or (k + 127) // 128 not in (1, 2, 3, 4, 8, 16, 32):
            return False
        k = n
    return True


_chain_ws = {}


def _chain_desc(x0, layers, M):
    d = _L.MlpChainDesc()
    d.nl, d.M, d.K0 = len(layers), int(M), int(x0.shape[1])
    _require_cuda(x0)
    d.x0, d.ld_x0 = x0.data_ptr(), x0.stride(0)
    flops, nbytes, k = 0.0, float(x0.numel() * 4), x0.shape[1]
    for i, lay in enumerate(layers):
        W, bias, relu, mask, mscale, out = lay[:6]
        W_lo = lay[6] if len(lay) > 6 else None          # (split operands: the remainders of W, same layout)
        if W.dtype != torch.bfloat16 or W.stride(1) != 1:
            raise _L.HulcKernelError("mlp_chain: weights are bf16 k-major shadows")
        _require_cuda(W, bias, mask, out, W_lo)
        e = d.layers[i]
        e.W, e.ldw = W.data_ptr(), W.stride(0)
        if W_lo is not None:
            if W_lo.dtype != torch.bfloat16 or W_lo.shape != W.shape or W_lo.stride() != W.stride():
                raise _L.HulcKernelError("mlp_chain: W_lo is laid out like W")
            e.W_lo = W_lo.data_ptr()
        e.bias = bias.data_ptr() if bias is not None else None
        e.mask, e.ld_mask, e.mask_scale = (mask.data_ptr(), mask.stride(0), float(mscale)) if mask is not None else (None, 0, 1.0)
        e.out, e.ld_out, e.N, e.relu = out.data_ptr(), out.stride(0), int(W.shape[0]), int(bool(relu))
        flops += 2.0 * M * W.shape[0] * k
        nbytes += W.shape[0] * k * 2 + M * W.shape[0] * 4
        k = W.shape[0]
    return d, flops, nbytes


def _chain_workspace(device, need: int):
    # persistent workspace per (device, stream): its header (barrier counters) is zero before the first launch and every launch leaves it
    # zero; check_faults clears it after a barrier timeout (the one case that leaves counts behind)
    key = (device, _stream())
    ws = _chain_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.zeros(max(need, 4 << 20) // 4 + 1, dtype=torch.float32, device=device)
        if not torch.cuda.is_current_stream_capturing():
            _chain_ws[key] = ws
    return ws


def mlp_chain(x0, layers, M, share=None):
    """layers: [(W bf16 [N][K] k-major, bias fp32 or None, relu flag, mask fp32 (M, N) or None, mask_scale, out fp32 (M, N))]; one persistent
    launch (csrc/mlp_chain.hip).  x0 fp32 (M, K0), unit inner stride."""
    d, flops, nbytes = _chain_desc(x0, layers, M)
    lib = _L.load()
    lib.hulc_mlp_chain_workspace.restype = _c.c_long
    ws = _chain_workspace(x0.device, int(lib.hulc_mlp_chain_workspace(_c.byref(d))))
    _call_shared(share, "hulc_mlp_chain", _c.byref(d), ws, fault_word(x0.device), key=("mlp_chain", M, int(x0.shape[1])) + tuple(int(l[0].shape[0]) for l in layers),
                 flops=flops, nbytes=nbytes)


def mlp_chain2_ok(Ma: int, K0a: int, widths_a, Mb: int, K0b: int, widths_b, device, share=None) -> bool:
    """shapes hulc_mlp_chain2 takes: two chains of <= 32 rows, the second no deeper than the first and of its widths where both run"""
    widths_a, widths_b = list(widths_a), list(widths_b)
    return (Ma <= 32 and Mb <= 32 and 1 <= len(widths_b) <= len(widths_a) and widths_a[:len(widths_b)] == widths_b
            and mlp_chain_ok(Ma, K0a, widths_a, device, share) and mlp_chain_ok(Mb, K0b, widths_b, device, share))


def mlp_chain2(xa, layers_a, Ma, xb, layers_b, Mb, share=None):
    """two independent chains (mlp_chain's layer tuples) as ONE persistent launch — the visual and the language goal encoder, and their
    data-gradient chains (hulc_mlp_chain2, include/hulc2_amd.h)"""
    da, fa, na = _chain_desc(xa, layers_a, Ma)
    db, fb, nb = _chain_desc(xb, layers_b, Mb)
    lib = _L.load()
    lib.hulc_mlp_chain_workspace.restype = _c.c_long
    ws = _chain_workspace(xa.device, int(lib.hulc_mlp_chain_workspace(_c.byref(da))) + int(lib.hulc_mlp_chain_workspace(_c.byref(db))))
    _call_shared(share, "hulc_mlp_chain2", _c.byref(da), _c.byref(db), ws, fault_word(xa.device),
          key=("mlp_chain2", Ma, Mb, int(xa.shape[1]), int(xb.shape[1])) + tuple(int(l[0].shape[0]) for l in layers_a), flops=fa + fb, nbytes=na + nb)


# ------------------------------------------------------------------------------------------------
# weight gradients: collected during a backward pass, one grouped launch at its end (csrc/wgrad_group.hip)
# ------------------------------------------------------------------------------------------------
_wg_pending = {}        # device -> list of (A, B, C, rowsum, M, N, K, lda, ldb, ldc, accumulate, rowsum_accumulate)
_wg_ws = {}             # (device, stream) -> workspace (counters zeroed once, left zero by every launch)
_wg_armed = set()


def wgrad_group_ok(A, B, C, M, N, K, lda, ldb, ldc, any_size=False) -> bool:
    """shapes hulc_wgrad_group takes; the three 2048^3 products of the recurrent decoder stay on their own kernel (gemm_tn128) unless any_size"""
    import os
    if _compute_mode != BF16 or os.environ.get("HULC_NO_WGRAD_GROUP"):
        return False
    if K % 32 or M % 8 or N % 8 or C.dtype != torch.float32 or A.dtype not in (torch.float32, torch.bfloat16) \
            or B.dtype not in (torch.float32, torch.bfloat16):
        return False
    if not any_size and min(M, N, K) >= 512 and M % 128 == 0 and N % 128 == 0 and K % 128 == 0 and A.dtype == B.dtype == torch.bfloat16:
        return False
    ea, eb = (4 if A.dtype == torch.float32 else 8), (4 if B.dtype == torch.float32 else 8)
    return lda % ea == 0 and ldb % eb == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0 and C.data_ptr() % 4 == 0


def wgrad(A, B, C, M, N, K, lda, ldb, ldc, accumulate=False, rowsum=None, rowsum_accumulate=False, defer=False, col_perm=0, col_mul=1, store_rows=0, conv_taps_wp=0):
    """C[M,N] (+)= A^T B with A (K, M), B (K, N) row-major (+ rowsum[m] (+)= sum_k A[k][m]).  defer=True (C and rowsum are final
    destinations nobody reads before the backward pass ends — the trainer's gradient arena): the product joins the grouped launch issued
    when autograd finishes the pass; otherwise it runs now.  Returns True when rowsum was (or will be) produced by the same launch."""
    _require_cuda(A, B, C, rowsum)
    dev = C.device
    mine = {C.data_ptr()} | ({rowsum.data_ptr()} if rowsum is not None else set())
    q = _wg_pending.get(dev)
    if q and any(e[2].data_ptr() in mine or (e[3] is not None and e[3].data_ptr() in mine) for e in q):
        # a second writer of the same destination (a shared layer applied twice in one pass, e.g. at different row counts): the queued first
        # writer runs NOW, whichever path the second one takes — an immediate GEMM with accumulate=True must not land on a slice the deferred
        # overwrite has not written yet
        wgrad_flush(dev)
    if not wgrad_group_ok(A, B, C, M, N, K, lda, ldb, ldc, any_size=bool(col_perm) or col_mul > 1):
        if col_perm or col_mul > 1 or store_rows:
            raise _L.HulcKernelError("wgrad: col_perm / col_mul need the grouped kernel (check wgrad_group_ok first)")
        fused = rowsum is not None and gemm_fuses_rowsum(M, False) and A.dtype == torch.float32
        gemm(A, B, C, M, N, K, lda, ldb, ldc, a_kmajor=False, b_kmajor=False, accumulate=accumulate,
             rowsum=rowsum if fused else None, rowsum_accumulate=rowsum_accumulate)
        if rowsum is not None and not fused:
            colsum(A, K, M, lda, rowsum, accumulate=rowsum_accumulate)
        return True
    q = _wg_pending.setdefault(dev, [])
    q.append((A, B, C, rowsum, int(M), int(N), int(K), int(lda), int(ldb), int(ldc), bool(accumulate), bool(rowsum_accumulate), int(col_perm), int(col_mul), int(store_rows), int(conv_taps_wp)))
    # the stream the operands were produced on (a backward node runs on the stream of its forward: the gripper camera's encoder lives on a side
    # stream).  The grouped launch is issued on whatever stream is current at the end of the pass and must wait for every producer stream:
    # autograd's own end-of-pass join only covers streams on which a LEAF received a defined gradient — with gradient sinks the Functions return
    # None, and under the step node the parameters' AccumulateGrad nodes belong to the caller's stream (round 5: found as zero / NaN weight
    # gradients of the gripper encoder's head when the grouped launch overtook the side stream)
    _wg_streams.setdefault(dev, set()).add(_stream_of(dev))
    if not defer:
        wgrad_flush(dev)
    elif dev not in _wg_armed:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(lambda d=dev: (_wg_armed.discard(d), wgrad_flush(d)))
            _wg_armed.add(dev)
        except RuntimeError:               # not inside a backward pass
            wgrad_flush(dev)
    return True


def wgrad_reset(device) -> None:
    """drop products left behind by a backward pass that raised (start of a trainer step)"""
    _wg_pending.pop(device, None)
    _wg_streams.pop(device, None)
    _wg_armed.discard(device)
    ent = _wg_side.get(device)
    if ent is not None and len(ent) > 2:
        torch.cuda.current_stream(device).wait_stream(ent[0])
        del ent[2:]
        ent[1].clear()


_wg_side = {}           # device -> (side stream, [item lists kept alive until the join])


def wgrad_flush_early(device) -> None:
    """Issue what is pending NOW on a second stream (called from a gradient hook at the perceptual embedding: everything but the camera
    encoders' layers is queued by then): the grouped launch runs in the gaps of the convolution backward instead of behind it.  The operands
    stay referenced until wgrad_flush joins the stream at the end of the pass."""
    import os
    q = _wg_pending.get(device)
    if not q or os.environ.get("HULC_WGRAD_EARLY", "0") != "1":     # measured: 4.01 vs 3.73 ms/step — the grouped launch and the convolution
        return                                                       # backward slow each other down more than the overlap hides; off by default
    ent = _wg_side.get(device)
    if ent is None:
        ent = _wg_side[device] = [torch.cuda.Stream(device=device), []]
    cur = torch.cuda.current_stream(device)
    ent[0].wait_stream(cur)
    ent[1].append(list(q))
    with torch.cuda.stream(ent[0]):
        _wgrad_issue(device)
    ent.append("joined-pending")


def wgrad_flush(device=None) -> None:
    """issue the pending weight-gradient products (all devices when device is None) and join an early flush's stream"""
    for dev in ([device] if device is not None else list(set(_wg_pending) | set(_wg_side))):
        ent = _wg_side.get(dev)
        if ent is not None and len(ent) > 2:
            torch.cuda.current_stream(dev).wait_stream(ent[0])
            del ent[2:]
            ent[1].clear()
        _wgrad_issue(dev)


_wg_streams = {}        # device -> raw handles of the streams that produced operands of the pending products


def join_stream(dev, other) -> None:
    """the current stream waits for everything `other` holds.  Inside a hipGraph capture only a stream that is part of the capture may be
    joined (an event of an un-captured stream would cross the capture boundary); one that is not has no captured work to wait for."""
    cur = torch.cuda.current_stream(dev)
    if torch.cuda.is_current_stream_capturing():
        with torch.cuda.stream(other):
            inside = torch.cuda.is_current_stream_capturing()
        if not inside:
            return
    cur.wait_stream(other)


def _wgrad_issue(dev) -> None:
    if True:
        q = _wg_pending.pop(dev, None)
        producers = _wg_streams.pop(dev, None)
        if producers:                           # (also with nothing queued: a forked weight-gradient branch writes the arena by itself)
            here = _stream_of(dev)
            for h in producers:
                if h != here and h != 0:
                    join_stream(dev, torch.cuda.ExternalStream(h, device=dev))
        if not q:
            return
        n = len(q)
        items = (_L.WgradItem * n)()
        flops = nbytes = 0.0
        for it, (A, B, C, rs, M, N, K, lda, ldb, ldc, acc, racc, *rest) in zip(items, q):
            it.A, it.B, it.C = A.data_ptr(), B.data_ptr(), C.data_ptr()
            it.rowsum = rs.data_ptr() if rs is not None else None
            it.M, it.N, it.K, it.lda, it.ldb, it.ldc = M, N, K, lda, ldb, ldc
            it.a_dtype, it.b_dtype = _dt(A), _dt(B)
            it.accumulate, it.rowsum_accumulate = int(acc), int(racc)
            it.col_perm = rest[0] if rest else 0
            it.col_mul = rest[1] if len(rest) > 1 else 1
            it.store_rows = rest[2] if len(rest) > 2 else 0
            it.conv_taps_wp = rest[3] if len(rest) > 3 else 0
            flops += 2.0 * M * N * K * (9 if (len(rest) > 3 and rest[3]) else 1)
            nbytes += K * M * A.element_size() + K * N * B.element_size() + M * N * 4 * (2 if acc else 1)
        lib = _L.load()
        lib.hulc_wgrad_group_workspace.restype = _c.c_long
        need = int(lib.hulc_wgrad_group_workspace(items, _i(n)))
        with torch.cuda.device(dev):
            key = (dev, _stream())
            ws = _wg_ws.get(key)
            if ws is None or ws.numel() * 4 < need:
                ws = torch.zeros((need + (8 << 20)) // 4, dtype=torch.float32, device=dev)
                if not torch.cuda.is_current_stream_capturing():      # (memory of a graph's pool must not outlive the graph in this cache)
                    _wg_ws[key] = ws
            with _Timed(("wgrad_group", n), flops, nbytes):
                _L.check(lib.hulc_wgrad_group(items, _i(n), _c.c_void_p(ws.data_ptr()), _l(ws.numel() * 4), _c.c_void_p(_stream())),
                         "hulc_wgrad_group")


def _ffn_ws(T, FF, device):
    lib = _L.load()
    lib.hulc_ffn_workspace.restype = _c.c_long
    return _ws(lib.hulc_ffn_workspace(_i(T), _i(FF)), device)


def ffn_fwd(x, W1, b1, W2, b2, T, D, FF, drop_p, seed, f):
    """fused transformer feed-forward block (csrc/ffn_fused.hip); W1 / W2 bf16.  f = None: the FF / 128 hidden-slice partials are left
    unsummed in the returned workspace ((FF / 128, T, 128) fp32 at its start) for a consumer that sums them (layernorm_slab_fwd)."""
    fl = 2.0 * 2 * T * D * FF
    ws = _ffn_ws(T, FF, x.device)
    _call("hulc_ffn_fwd", x, W1, b1, W2, b2, _i(T), _i(D), _i(FF), _f(drop_p), _u64(seed), _sd(x, drop_p), f, ws,
          key=("ffn_fwd", T, D, FF), flops=fl, nbytes=_nbytes(x, W1, W2, f))
    return ws


def ffn_bwd(x, df, W1, b1, W1T, W2T, T, D, FF, drop_p, seed, dx, dW1, db1, dW2, accumulate_params=False, dx_accumulate=False):
    fl = 2.0 * 5 * T * D * FF                                  # recompute + two data-gradient + two weight-gradient products
    ws = _ffn_ws(T, FF, x.device)                              # dx = None: the slice partials of dx stay at the start of ws
    _call("hulc_ffn_bwd", x, df, W1, b1, W1T, W2T, _i(T), _i(D), _i(FF), _f(drop_p), _u64(seed), _sd(x, drop_p), dx, _i(dx_accumulate),
          dW1, db1, dW2, _i(accumulate_params), ws,
          key=("ffn_bwd", T, D, FF), flops=fl, nbytes=_nbytes(x, df, W1, W1T, W2T, dx, dW1, dW2))
    return ws


def layernorm_slab_fwd(x, o_slabs, n_o, o_stride, drop_p, seed, gamma, beta, eps, R, D, pre_out, y, mean, rstd):
    """y = LayerNorm(x + dropout(sum of the n_o partial slabs)) — slice sum + residual + norm in one launch"""
    _call("hulc_layernorm_slab_fwd", x, o_slabs, _i(n_o), _l(o_stride), _f(drop_p), _u64(seed), _sd(x, drop_p), gamma, beta, _f(eps), _i(R), _i(D),
          pre_out, y, mean, rstd)


def ln_partial_reduce(partial, P, D, dgamma, dbeta, accumulate=False):
    """partial (P, 2, D): rows of [dgamma | dbeta] partial sums -> the two parameter gradients, fixed order, one launch"""
    _call("hulc_ln_partial_reduce", partial, _i(P), _i(D), dgamma, dbeta, _i(accumulate))


def ln_partial_reduce_multi(partial, P, D, dgammas, dbetas, accumulates):
    """partial (n, P, 2, D) -> n pairs of parameter gradients, one launch (hulc_ln_partial_reduce_multi)"""
    n = len(dgammas)
    _require_cuda(partial, *dgammas, *dbetas)
    dg = (_c.c_void_p * n)(*[t.data_ptr() for t in dgammas])
    db = (_c.c_void_p * n)(*[t.data_ptr() for t in dbetas])
    acc = (_c.c_int * n)(*[int(bool(a)) for a in accumulates])
    _call("hulc_ln_partial_reduce_multi", partial, _i(n), _i(P), _i(D), dg, db, acc)


def _txl_desc(x, Wqkv, bqkv, gamma, B, S, H, drop_p, seed_attn, seed_ln, eps):
    d = _L.TxlAttnDesc()
    _require_cuda(x, Wqkv, bqkv, gamma)
    d.x, d.Wqkv, d.bqkv, d.gamma = x.data_ptr(), Wqkv.data_ptr(), bqkv.data_ptr(), gamma.data_ptr()
    d.eps, d.B, d.S, d.H, d.E = float(eps), int(B), int(S), int(H), int(x.shape[-1])
    d.drop_p, d.seed_attn, d.seed_ln = float(drop_p), int(seed_attn) & 0xFFFFFFFFFFFFFFFF, int(seed_ln) & 0xFFFFFFFFFFFFFFFF
    d.seed_dev = step_state(x.device).data_ptr() if drop_p > 0.0 else None
    return d


def txl_attn_fwd(x, Wqkv, bqkv, Wo, bo, gamma, beta, eps, B, S, H, drop_p, seed_attn, seed_ln, y, pre=None, mean=None, rstd=None, ctx=None):
    """attention half of the post-norm transformer layer, one launch (csrc/txl_fused.hip); Wqkv / Wo bf16"""
    for t in (Wqkv, Wo):
        if t.dtype != torch.bfloat16:
            raise _L.HulcKernelError("txl_attn_fwd: weights are the bf16 shadows (bf16 compute mode)")
    _require_contiguous(x=x, Wqkv=Wqkv, Wo=Wo, y=y)
    _require_cuda(Wo, bo, beta, y, pre, mean, rstd, ctx)
    d = _txl_desc(x, Wqkv, bqkv, gamma, B, S, H, drop_p, seed_attn, seed_ln, eps)
    d.Wo, d.bo, d.beta, d.y = Wo.data_ptr(), bo.data_ptr(), beta.data_ptr(), y.data_ptr()
    d.pre, d.mean, d.rstd = [t.data_ptr() if t is not None else None for t in (pre, mean, rstd)]
    d.ctx = ctx.data_ptr() if ctx is not None else None
    T, E = B * S, x.shape[-1]
    fl = 2.0 * T * E * 4 * E + 2.0 * 2 * B * H * S * S * (E // H)
    _call("hulc_txl_attn_fwd", _c.byref(d), key=("txl_attn_fwd", B, S), flops=fl, nbytes=_nbytes(x, Wqkv, Wo, y, pre, ctx))


def txl_attn_bwd(x, Wqkv, WqkvT, WoT, bqkv, gamma, eps, B, S, H, drop_p, seed_attn, seed_ln, pre, mean, rstd, dy, dy_slab, n_slab,
                 slab_stride, dx, d_o, dqkv, ln_partial):
    for t in (Wqkv, WqkvT, WoT):
        if t.dtype != torch.bfloat16:
            raise _L.HulcKernelError("txl_attn_bwd: weights are the bf16 shadows (bf16 compute mode)")
    _require_contiguous(x=x, Wqkv=Wqkv, WqkvT=WqkvT, WoT=WoT, dy=dy, dx=dx, d_o=d_o, dqkv=dqkv, pre=pre)
    _require_cuda(WqkvT, WoT, pre, mean, rstd, dy, dy_slab, dx, d_o, dqkv, ln_partial)
    d = _txl_desc(x, Wqkv, bqkv, gamma, B, S, H, drop_p, seed_attn, seed_ln, eps)
    d.WqkvT, d.WoT = WqkvT.data_ptr(), WoT.data_ptr()
    d.pre, d.mean, d.rstd, d.dy = pre.data_ptr(), mean.data_ptr(), rstd.data_ptr(), dy.data_ptr()
    d.dy_slab, d.n_slab, d.slab_stride = (dy_slab.data_ptr() if dy_slab is not None else None), int(n_slab), int(slab_stride)
    d.dx, d.d_o, d.dqkv, d.ln_partial = dx.data_ptr(), d_o.data_ptr(), dqkv.data_ptr(), ln_partial.data_ptr()
    T, E = B * S, x.shape[-1]
    fl = 2.0 * T * E * (2 * E + 5 * E + 3 * E) + 2.0 * 10 * B * H * S * S * (E // H)
    _call("hulc_txl_attn_bwd", _c.byref(d), key=("txl_attn_bwd", B, S), flops=fl, nbytes=_nbytes(x, Wqkv, WqkvT, WoT, dy, dx, d_o, dqkv, pre))


def exact_site_in_bf16_step() -> bool:
    """inside a forward-only exact scope of a bf16 step (site_scope of a selected site): kernels with a split-operand forward take it"""
    return _base_mode == "bf16" and _compute_mode != BF16 and _bwd_mode == "bf16"


def mlp2_rows_ok(x, W1, W2) -> bool:
    """shapes hulc_mlp2_rows_* take (include/hulc2_amd.h): the camera encoders' fc1 -> ReLU -> fc2 head"""
    import os
    H, K = W1.shape
    OUT = W2.shape[0]
    return ((_compute_mode == BF16 or exact_site_in_bf16_step()) and not os.environ.get("HULC_NO_MLP2_ROWS") and x.is_cuda and x.dim() == 2 and x.dtype == torch.float32
            and x.is_contiguous() and K == 128 and x.shape[1] == 128 and H % 128 == 0 and OUT % 32 == 0 and 32 <= OUT <= 128 and W2.shape[1] == H
            and x.data_ptr() % 16 == 0)


def mlp2_rows_fwd(x, W1, b1, W2, b2, y, W1_lo=None, W2_lo=None):
    T, H, OUT = x.shape[0], W1.shape[0], W2.shape[0]
    _call("hulc_mlp2_rows_fwd", x, W1, b1, W2, b2, W1_lo, W2_lo, _i(T), _i(128), _i(H), _i(OUT), y, key=("mlp2_rows_fwd", T, H, OUT, W1_lo is not None),
          flops=2.0 * T * H * (128 + OUT) * (3 if W1_lo is not None else 1), nbytes=_nbytes(x, W1, W2, y))


def mlp2_rows_bwd(x, dy, W1, b1, W1T, W2T, dx, h, dh):
    T, H, OUT = x.shape[0], W1.shape[0], W2T.shape[1]
    _call("hulc_mlp2_rows_bwd", x, dy, W1, b1, W1T, W2T, _i(T), _i(128), _i(H), _i(OUT), dx, h, dh, key=("mlp2_rows_bwd", T, H, OUT),
          flops=2.0 * T * H * (2 * 128 + OUT), nbytes=_nbytes(x, dy, W1, W1T, W2T, dx, h, dh))


def txl_block_desc(emb, pos, pos_ids, B, S, H, FF, drop_p, seed_pos, eps, layers, pooled=None, dpooled=None, demb=None):
    """hulc_txl_block_desc from tensors.  layers: one dict per layer, keys = the fields of hulc_txl_block_layer (tensors, None, or the four
    integer seeds)."""
    d = _L.TxlBlockDesc()
    _require_cuda(emb, pos, pos_ids, pooled, dpooled, demb)
    _require_contiguous(emb=emb, pos=pos)
    if len(layers) > _L.TXL_MAX_LAYERS:
        raise _L.HulcKernelError("txl_block: at most %d layers" % _L.TXL_MAX_LAYERS)
    d.L, d.B, d.S, d.H, d.E, d.FF = len(layers), int(B), int(S), int(H), int(emb.shape[-1]), int(FF)
    d.eps, d.drop_p, d.seed_pos = float(eps), float(drop_p), int(seed_pos) & 0xFFFFFFFFFFFFFFFF
    d.seed_dev = step_state(emb.device).data_ptr() if drop_p > 0.0 else None
    d.emb, d.pos, d.pos_ids = emb.data_ptr(), pos.data_ptr(), pos_ids.data_ptr()
    d.pooled, d.dpooled, d.demb = [t.data_ptr() if t is not None else None for t in (pooled, dpooled, demb)]
    # sequences shared between workgroups (csrc/txl_block.hip) only on a whole MI355X with the stream to itself, like the chain kernels
    import os
    d.exclusive = int(not concurrent_streams() and not os.environ.get("HULC_TXL_NO_SHARE"))
    if d.exclusive:
        lib = _L.load()
        lib.hulc_txl_block_workspace.restype = _c.c_long
        need = int(lib.hulc_txl_block_workspace(_i(B), _i(len(layers))))
        key = (emb.device, _stream(), "txl")
        ws = _chain_ws.get(key)
        if ws is None or ws.numel() * 4 < need:
            ws = torch.zeros(need // 4 + 1, dtype=torch.float32, device=emb.device)
            if not torch.cuda.is_current_stream_capturing():
                _chain_ws[key] = ws
        d._ws = ws                                   # (kept alive with the description)
        d.ws, d.err_sticky = ws.data_ptr(), fault_word(emb.device).data_ptr()
    for i, rec in enumerate(layers):
        e = d.layers[i]
        for name, v in rec.items():
            if name.startswith("seed_"):
                setattr(e, name, int(v) & 0xFFFFFFFFFFFFFFFF)
            elif v is not None:
                _require_cuda(v)
                if not v.is_contiguous():
                    raise _L.HulcKernelError("txl_block: %s must be contiguous" % name)
                if name[0] == "W" and v.dtype != torch.bfloat16:
                    raise _L.HulcKernelError("txl_block: weights are the bf16 shadows (bf16 compute mode)")
                setattr(e, name, v.data_ptr())
    return d


def _txl_block_flops(B, S, H, E, FF, L, bwd):
    T = B * S
    attn = 2.0 * T * E * 4 * E + 2.0 * 2 * B * H * S * S * (E // H)
    ffn = 2.0 * T * E * FF * 2
    return L * ((2.5 * attn + 2.5 * ffn) if bwd else (attn + ffn))


def txl_block_fwd(d, B, S, H, E, FF, L, share=None):
    """the whole posterior trunk (position embedding -> L transformer layers -> sequence mean) as one launch (csrc/txl_block.hip)"""
    _call_shared(share, "hulc_txl_block_fwd", _c.byref(d), key=("txl_block_fwd", B, S, L), flops=_txl_block_flops(B, S, H, E, FF, L, False))


def txl_block_bwd(d, B, S, H, E, FF, L, share=None):
    _call_shared(share, "hulc_txl_block_bwd", _c.byref(d), key=("txl_block_bwd", B, S, L), flops=_txl_block_flops(B, S, H, E, FF, L, True))


def residual_bf16(p32, hi, lo, segments):
    """lo = bf16(p32 - float(hi)) on the segments {src offset, count, dst offset} (int64 (n, 3) device tensor)"""
    _call("hulc_residual_bf16", p32, hi, lo, segments, _i(segments.shape[0]))


def gather_chunks(src0, src1, dst, idx):
    """dst 8-byte chunk c = chunk idx[c] of src0 (or of src1 when bit 31 is set): the packed weight copies of a step in one launch"""
    _call("hulc_gather_chunks", src0, src1, dst, idx, _l(idx.numel()))


def ffn_frag_perm(layout: int, FF: int):
    """host: numpy int32 (FF * 128,) — hulc_ffn_frag_perm (include/hulc2_amd.h)"""
    import numpy as np
    out = np.empty(FF * 128, dtype=np.int32)
    lib = _L.load()
    _L.check(lib.hulc_ffn_frag_perm(_i(layout), _i(FF), out.ctypes.data_as(_c.c_void_p)), "hulc_ffn_frag_perm")
    return out


def repack_conv_weights(src_f32, dst_bf16, table):
    """table: int64 (n, 7) device tensor {src offset, dst offset, Cout, Cin, KH, KW, mode} (mode 0 oihw_flat, 1 ohwi, 2 ihwo)"""
    _call("hulc_repack_conv_weights", src_f32, dst_bf16, table, _i(table.shape[0]))


def transpose_bf16_tiles(src, dst, tiles):
    """tiles: int64 (ntiles, 5) device tensor {offset, rows, cols, tile row, tile col}"""
    _call("hulc_transpose_bf16_tiles", src, dst, tiles, _i(tiles.shape[0]))


def rnn_wavefront(z0, z_step, S, B, H, wA, wB1, wB2, transposed, add1=None, add1_step=0, ld_add1=0, bias1=(None, None), bias2=(None, None),
                  mask1=None, mask1_step=0, ld_mask1=0, mask2=None, mask2_step=0, ld_mask2=0, relu=False, mirror_t=False, add1c=None, zero_edges=False):
    """Both RNN layers of one direction as one persistent kernel (csrc/rnn_wavefront.hip).  z0: view of the (zero) state row
    wave step 0 reads; rows advance by z_step elements.  Weights are bf16 (H, H) matrices, `transposed` applies to all three."""
    for w in (wA, wB1, wB2):
        if w.dtype != torch.bfloat16:
            raise _L.HulcKernelError("rnn_wavefront: weights must be bf16 shadows (bf16 compute mode)")
    d = _L.RnnWaveDesc()
    d.z, d.z_step = z0.data_ptr(), int(z_step)
    d.wA, d.wB1, d.wB2 = wA.data_ptr(), wB1.data_ptr(), wB2.data_ptr()
    d.ldA, d.ldB1, d.ldB2 = wA.stride(0), wB1.stride(0), wB2.stride(0)
    d.tA = d.tB1 = d.tB2 = int(bool(transposed))
    d.add1, d.add1_step, d.ld_add1 = (add1.data_ptr() if add1 is not None else None), int(add1_step), int(ld_add1)
    d.bias1a, d.bias1b = [b.data_ptr() if b is not None else None for b in bias1]
    d.bias2a, d.bias2b = [b.data_ptr() if b is not None else None for b in bias2]
    d.mask1, d.mask1_step, d.ld_mask1 = (mask1.data_ptr() if mask1 is not None else None), int(mask1_step), int(ld_mask1)
    d.mask2, d.mask2_step, d.ld_mask2 = (mask2.data_ptr() if mask2 is not None else None), int(mask2_step), int(ld_mask2)
    d.relu, d.S, d.B, d.H = int(relu), int(S), int(B), int(H)
    d.mirror_t = int(bool(mirror_t))
    d.err_sticky = fault_word(z0.device).data_ptr()
    d.add1c, d.ld_add1c = (add1c.data_ptr(), add1c.stride(0)) if add1c is not None else (None, 0)
    d.zero_edges = int(bool(zero_edges))
    lib = _L.load()
    lib.hulc_rnn_wavefront_workspace.restype = _c.c_long
    ws = _ws(lib.hulc_rnn_wavefront_workspace(_i(S), _i(B), _i(H)), z0.device)
    # algorithmic work: S wave steps of a (B x 2H) x (2H x 2H) product with one H x H block structurally zero; bytes: the
    # three weight matrices once, per step the fp32 state row written + the bf16 copy written and read + add / masks read
    flops = 2.0 * B * 3 * H * H * S
    nbytes = 3.0 * H * H * 2 + S * B * (2 * H * 4 + 2 * 2 * H * 2 + (H * 4 if add1 is not None else 0)
                                        + (H * 4 if mask1 is not None else 0) + (H * 4 if mask2 is not None else 0))
    _call("hulc_rnn_wavefront", _c.byref(d), ws, key=("rnn_wavefront", S, B, H, int(bool(transposed))), flops=flops, nbytes=nbytes)
    lib.hulc_rnn_wavefront_mirror_offset.restype = ctypes.c_long
    lib.hulc_rnn_wavefront_mirror_t_offset.restype = ctypes.c_long
    off = lib.hulc_rnn_wavefront_mirror_offset() // 2
    w16 = ws.view(torch.bfloat16)
    z16 = w16[off:off + (S + 2) * B * 2 * H].view(S + 2, B, 2 * H)
    offt = lib.hulc_rnn_wavefront_mirror_t_offset(_i(S), _i(B), _i(H)) // 2
    z16t = w16[offt:offt + (S + 2) * B * 2 * H].view(2 * H, (S + 2) * B) if (offt and mirror_t) else None     # (feature, token = row * B + b)
    return z16, z16t      # bf16 mirror of the S+2 state rows


def mix_loss_fwd(y, act, out, T, A, n_mix, num_classes, ld, log_scale_min, gripper_alpha, act_min, act_max, nseg=1, time_major_B=0):
    """out: (nseg, 3) = {total, nll_mean, ce_mean} per segment of T / nseg tokens (time_major_B: see hulc_mix_desc)."""
    d = _mix_desc(T, A, n_mix, num_classes, ld, log_scale_min, gripper_alpha, act_min, act_max, nseg, time_major_B)
    lib = _L.load()
    lib.hulc_mix_loss_workspace.restype = _c.c_long
    ws = _ws(lib.hulc_mix_loss_workspace(_c.byref(d)), y.device)
    _call("hulc_mix_loss_fwd", _c.byref(d), y, act, out, ws)


def mix_loss_bwd(y, act, gout, dy, ld_dy, T, A, n_mix, num_classes, ld, log_scale_min, gripper_alpha, act_min, act_max, nseg=1, time_major_B=0):
    """writes every column of dy (the pad columns beyond 3 * A * n_mix + 2 as zeros)"""
    d = _mix_desc(T, A, n_mix, num_classes, ld, log_scale_min, gripper_alpha, act_min, act_max, nseg, time_major_B)
    _call("hulc_mix_loss_bwd", _c.byref(d), y, act, gout, dy, _l(ld_dy))


def cat_kl_fwd(pp, pr, B, G, CLS, beta, out, kl_group, nseg=1):
    _call("hulc_cat_kl_fwd", pp, pr, _i(B), _i(G), _i(CLS), _f(beta), _i(nseg), out, kl_group)


def cat_kl_bwd(pp, pr, kl_group, B, G, CLS, beta, mix, gout, dpp, dpr, nseg=1):
    _call("hulc_cat_kl_bwd", pp, pr, kl_group, _i(B), _i(G), _i(CLS), _f(beta), _f(mix), gout, _i(nseg), dpp, dpr)


def plan_sample_fwd(logits, idx_in, seed, NG, CLS, idx_out, plan):
    _call("hulc_plan_sample_fwd", logits, idx_in, _u64(seed), step_state(logits.device) if idx_in is None else None, _i(NG), _i(CLS),
          idx_out, plan)


def plan_sample_bwd(logits, dplan, NG, CLS, dlogits, accumulate=False):
    _call("hulc_plan_sample_bwd", logits, dplan, _i(NG), _i(CLS), dlogits, _i(accumulate))


def clip_loss_fwd(im, tx, use, logit_scale, M, D, out, row0=0):
    _call("hulc_clip_loss_fwd", im, tx, use, _i(row0), logit_scale, _i(M), _i(D), out)


def clip_loss_bwd(im, tx, use, logit_scale, M, D, gout, dim, dtx, dscale, row0=0):
    _call("hulc_clip_loss_bwd", im, tx, use, _i(row0), logit_scale, _i(M), _i(D), gout, dim, dtx, dscale)


def loss_combine_fwd(kls, acts, clip, n, beta, out):
    _call("hulc_loss_combine_fwd", kls, acts, clip, _i(n), _f(beta), out)


def loss_combine_bwd(g, n, beta, dkls, dacts, dclip):
    _call("hulc_loss_combine_bwd", g, _i(n), _f(beta), dkls, dacts, dclip)


def emb_fanout_fwd(emb, N, S, D, n_last, lo, hi, e0, elast, edec_t):
    _call("hulc_emb_fanout_fwd", emb, _i(N), _i(S), _i(D), _i(n_last), _i(lo), _i(hi), e0, elast, edec_t)


def emb_fanin_bwd(g_rec, g0, g_last, g_dec_t, N, S, D, n_last, lo, hi, demb):
    _call("hulc_emb_fanin_bwd", g_rec, g0, g_last, g_dec_t, _i(N), _i(S), _i(D), _i(n_last), _i(lo), _i(hi), demb)


def actions_time_major(acts, obss, B, S, obs_dim, to_tcp, out):
    """acts / obss: lists (<= 4) of contiguous fp32 (B, S, 7) / (B, S, obs_dim) device tensors -> out (S * nseg * B, 7), see the header"""
    n = len(acts)
    _require_cuda(*acts, out)
    A = (_c.c_void_p * n)(*[a.data_ptr() for a in acts])
    O = (_c.c_void_p * n)(*[o.data_ptr() for o in obss]) if to_tcp else None
    if to_tcp:
        _require_cuda(*obss)
    _call("hulc_actions_time_major", A, O, _i(n), _i(B), _i(S), _i(obs_dim), _i(int(bool(to_tcp))), out)


def layernorm_fwd_ld(x, gamma, beta, eps, R, D, y, ld_y, mean, rstd):
    _call("hulc_layernorm_fwd_ld", x, gamma, beta, _f(eps), _i(R), _i(D), y, _l(ld_y), mean, rstd)


def layernorm_bwd_ld(dy, ld_dy, pre, mean, rstd, gamma, R, D, dpre, dgamma, dbeta, accumulate_params=False):
    lib = _L.load()
    lib.hulc_layernorm_bwd_workspace.restype = _c.c_long
    ws = _ws(lib.hulc_layernorm_bwd_workspace(_i(R), _i(D)), dy.device)
    _call("hulc_layernorm_bwd_ld", dy, _l(ld_dy), pre, mean, rstd, gamma, _i(R), _i(D), dpre, dgamma, dbeta, _i(accumulate_params), ws)


def world_to_tcp(act, robot_obs, n, obs_dim, out):
    _call("hulc_world_to_tcp", act, robot_obs, _i(n), _i(obs_dim), out)


# ---- sentence encoder pieces (csrc/lang_encoder.hip) --------------------------------------------------------------
def embed_ln_fwd(ids, word, pos, type0, gamma, beta, eps, T, S, D, out):
    if ids.dtype != torch.int64 or not ids.is_contiguous():
        raise TypeError("embed_ln_fwd: token ids are a contiguous int64 tensor")
    _call("hulc_embed_ln_fwd", ids, word, pos, type0, gamma, beta, _f(eps), _i(T), _i(S), _i(D), out)
    return out


def ln_wide_fwd(x, add, gamma, beta, eps, R, D, y):
    _call("hulc_ln_wide_fwd", x, add, gamma, beta, _f(eps), _i(R), _i(D), y)
    return y


def mha_masked_fwd(qkv, mask, B, S, nhead, hd, out):
    if mask.dtype != torch.int32 or not mask.is_contiguous():
        raise TypeError("mha_masked_fwd: the padding mask is a contiguous int32 (B, S) tensor")
    _call("hulc_mha_masked_fwd", qkv, mask, _i(B), _i(S), _i(nhead), _i(hd), out)
    return out


def masked_mean_fwd(x, mask, B, S, D, out):
    _call("hulc_masked_mean_fwd", x, mask, _i(B), _i(S), _i(D), out)
    return out


def tcp_to_world(act, robot_obs, n, obs_dim, out):
    _call("hulc_tcp_to_world", act, robot_obs, _i(n), _i(obs_dim), out)


def mix_sample(y, ld, T, A, n_mix, log_scale_min, gripper_bounds, act_out, seed, u_mix=None, u_inv=None, idx_out=None):
    """LogisticDecoderRNN._sample; u_mix / u_inv inject the uniforms (parity tests), else the counter RNG on `seed`."""
    d = _mix_desc(T, A, n_mix, 0, ld, log_scale_min, 0.0, gripper_bounds, gripper_bounds)
    # validation / rollout run outside the training graphs: the caller advances `seed` itself, no device step word involved
    _call("hulc_mix_sample", _c.byref(d), y, u_mix, u_inv, _u64(seed), None, gripper_bounds, act_out, idx_out)


def window_index(starts, sizes, B, S, out):
    """out (B, S) int32 = starts[b] + min(t, sizes[b] - 1): a padded play window as store frame numbers (base_dataset.py:94-112,149-154)"""
    for t in (starts, sizes, out):
        if t.dtype != torch.int32 or not t.is_contiguous():
            raise TypeError("window_index: starts / sizes / out are contiguous int32 tensors")
    _call("hulc_window_index", starts, sizes, _i(B), _i(S), out)
    return out


def window_rows(store, starts, sizes, B, S, out, zero_cols=(0, 0)):
    """out (B, S, D) fp32 gathered from store (n, D); padded steps repeat the last row, columns [zero_cols) are zero-padded
    (base_dataset.py:121-147 pad_sequence)"""
    if store.dtype != torch.float32 or out.dtype != torch.float32 or starts.dtype != torch.int32 or sizes.dtype != torch.int32:
        raise TypeError("window_rows: fp32 store / out, int32 starts / sizes")
    _require_contiguous(store=store, out=out, starts=starts, sizes=sizes)
    _call("hulc_window_rows", store, _i(store.shape[-1]), starts, sizes, _i(B), _i(S), _i(zero_cols[0]), _i(zero_cols[1]), out)
    return out


def adam_step(p, g, m, v, shadow, n, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, step_state_dev=None, lo=None, lo_ranges=(),
              loss_scale_dev=None, found_inf_dev=None):
    """step_state_dev: device {rng, step} words (see step_state); when given, the step count is read on device.  The update is skipped
    on the device while the fault word is set (a barrier kernel timed out upstream) — check_faults() then raises on the host.
    lo (bf16 arena like shadow) + lo_ranges (<= 8 (begin, end) element ranges): the rounding remainders w - bf16(w) of the updated weights
    inside the ranges are written by the same pass (hulc_adam_step_lo)."""
    # algorithmic bytes per element (bench.py's roofline): p, g, m, v read (16 B), p, m, v written (12 B), the bf16 shadow (2 B) and, inside
    # lo_ranges, the remainder (2 B)
    n_lo = sum(int(e) - int(b) for b, e in lo_ranges) if (lo is not None and lo_ranges) else 0
    nbytes = float(n) * (16 + 12 + (2 if shadow is not None else 0)) + 2.0 * n_lo
    if loss_scale_dev is not None or found_inf_dev is not None:
        # torch.amp.GradScaler's device scalars (hulc_adam_step_amp, ABI 5): fp32 tensors of one element on the arena's device
        for t in (loss_scale_dev, found_inf_dev):
            if t is not None and (t.dtype != torch.float32 or t.numel() != 1 or t.device != p.device):
                raise _L.HulcKernelError("adam_step: loss_scale / found_inf are one-element fp32 tensors on the parameters' device")
        has_lo = lo is not None and bool(lo_ranges)
        flat = [int(x) for r in lo_ranges for x in r] if has_lo else [0, 0]
        arr = (_c.c_long * len(flat))(*flat)
        _call("hulc_adam_step_amp", p, g, m, v, shadow, _l(n), _f(lr), _f(beta1), _f(beta2), _f(eps), _f(weight_decay), _i(step),
              step_state_dev, _f(grad_scale), fault_word(p.device), lo if has_lo else None, arr if has_lo else None,
              _i(len(lo_ranges) if has_lo else 0), loss_scale_dev, found_inf_dev, nbytes=nbytes)
        return
    if lo is None or not lo_ranges:
        _call("hulc_adam_step", p, g, m, v, shadow, _l(n), _f(lr), _f(beta1), _f(beta2), _f(eps), _f(weight_decay), _i(step),
              step_state_dev, _f(grad_scale), fault_word(p.device), nbytes=nbytes)
        return
    flat = [int(x) for r in lo_ranges for x in r]
    arr = (_c.c_long * len(flat))(*flat)
    _call("hulc_adam_step_lo", p, g, m, v, shadow, _l(n), _f(lr), _f(beta1), _f(beta2), _f(eps), _f(weight_decay), _i(step),
          step_state_dev, _f(grad_scale), fault_word(p.device), lo, arr, _i(len(lo_ranges)), nbytes=nbytes)


def step_count_advance_if(state, found_inf_dev=None) -> None:
    """state[1] += 1 unless the GradScaler's found_inf (device float) is set — the device-resident step count of hulc2_amd.optim.Adam"""
    _call("hulc_step_count_advance_if", state, found_inf_dev)


def derive_copies(bf16, bf16_t, tiles, p32, conv_dst, conv_table):
    """the transposed tiles (bf16 -> bf16_t) and the conv repacks (p32 -> conv_dst) of a step as one launch; either table may be None"""
    _call("hulc_derive_copies", bf16, bf16_t, tiles, _i(0 if tiles is None else tiles.shape[0]), p32, conv_dst, conv_table,
          _i(0 if conv_table is None else conv_table.shape[0]))


def gather_chunks2(a0, a1, ad, ai, b0, bd, bi):
    """two 8-byte-chunk gathers as one launch: (a0 | a1 by bit 31 of ai) -> ad, b0 -> bd; either index tensor may be None"""
    _call("hulc_gather_chunks2", a0, a1, ad, ai, _l(0 if ai is None else ai.numel()), b0, bd, bi, _l(0 if bi is None else bi.numel()))


def cast_f32_to_bf16(src, dst, n):
    _call("hulc_cast_f32_to_bf16", src, dst, _l(n))


def cast_bf16_to_f32(src, dst, n):
    _call("hulc_cast_bf16_to_f32", src, dst, _l(n))


def sum_chunks(src, W, chunk, dst):
    """dst (chunk,) = sum over the W rank chunks of src (W * chunk,), rank order, fp32 accumulation (direct gradient all-reduce)"""
    if src.dtype != dst.dtype:
        raise TypeError("sum_chunks: src and dst share a dtype")
    _call("hulc_sum_chunks", src, _i(_dt(src)), _i(W), _l(chunk), dst)


# ------------------------------------------------------------------------------------------------
# affordance model: padded-grid convolutions and their pointwise / reduction kernels (csrc/gridconv.hip, affordance.hip)
# ------------------------------------------------------------------------------------------------
class Grid:
    """A bf16 map (N, H, W, C) on the padded grid (layout: include/hulc2_amd.h, "PADDED GRID"): `rows` = the 2-D tensor of all rows including
    the zero guards, `t` = rows [guard, guard + R) — the tensor the kernels address.  Every kernel writes all R rows of its output (borders as
    zero), so only the guards are cleared here."""

    _pool = {}            # (device, N, H, W, C) -> [buffers]: guards zeroed once, recycled step after step (Grid.begin_step)
    _cursor = {}
    _pooling = False

    @classmethod
    def begin_step(cls) -> None:
        """a training step starts: every buffer handed out during the previous step (forward + backward) is free again"""
        cls._cursor = {}
        cls._pooling = True

    @classmethod
    def end_pooling(cls) -> None:
        cls._pooling = False
        cls._pool.clear()
        cls._cursor = {}

    def __init__(self, N, H, W, C, device):
        self.N, self.H, self.W, self.C = int(N), int(H), int(W), int(C)
        self.R = self.N * (self.H + 2) * (self.W + 2)
        self.guard = (self.W + 3 + 7) // 8 * 8
        tail = self.guard + 32                                  # + the rows that round R up to a multiple of 32 (weight-gradient K)
        key = (device, self.N, self.H, self.W, self.C)
        rows = None
        if Grid._pooling:
            i = Grid._cursor.get(key, 0)
            bufs = Grid._pool.setdefault(key, [])
            if i < len(bufs):
                rows = bufs[i]
            Grid._cursor[key] = i + 1
        if rows is None:
            rows = torch.empty(self.guard + self.R + tail, self.C, dtype=torch.bfloat16, device=device)
            rows[:self.guard].zero_()
            rows[self.guard + self.R:].zero_()
            if Grid._pooling and not torch.cuda.is_current_stream_capturing():      # (a graph's private pool dies with the graph)
                Grid._pool[key].append(rows)
        self.rows = rows
        self.t = self.rows[self.guard:self.guard + self.R]

    @property
    def Rpad(self) -> int:
        return (self.R + 31) // 32 * 32

    def pixel_strides(self):
        """(base tensor view at pixel (0, 0, 0), stride n, stride y, stride x) in elements: the strided-map form grid_upcat takes"""
        Wp = self.W + 2
        return self.t[Wp + 1:], (self.H + 2) * Wp * self.C, Wp * self.C, self.C

    def interior(self) -> torch.Tensor:
        """(N, H, W, C) view of the pixels"""
        return self.t.view(self.N, self.H + 2, self.W + 2, self.C)[:, 1:-1, 1:-1]


def gridconv3x3(x: Grid, wt, Cout, want_stats=False, y: "Grid" = None, out0=None, bias0=None, cin=None, flip=False):
    """y = conv3x3(x) on the grid; wt bf16 [Cout][9 * Cin]; -> (y Grid or None, stats partials or None)"""
    Cin = x.C if cin is None else cin
    _require_cuda(x.rows, wt, out0, bias0)
    if wt.dtype != torch.bfloat16 or tuple(wt.shape) != (Cout, 9 * Cin) or not wt.is_contiguous():
        raise _L.HulcKernelError("gridconv3x3: weights are bf16 [Cout][9 * Cin]")
    if out0 is None and y is None:
        y = Grid(x.N, x.H, x.W, Cout, x.rows.device)
    lib = _L.load()
    lib.hulc_gridconv_stats_bytes.restype = _c.c_long
    stats = _ws(lib.hulc_gridconv_stats_bytes(_i(x.N), _i(x.H), _i(x.W), _i(Cout)), x.rows.device) if want_stats else None
    _call("hulc_gridconv3x3", x.t, _l(x.C), wt, (y.t if y is not None else None), _l(y.C if y is not None else 0), _i(x.N), _i(x.H), _i(x.W), _i(Cin), _i(Cout),
          _i(flip), stats, out0, bias0, key=("gridconv3x3", x.N, x.H, x.W, Cin, Cout), flops=2.0 * x.R * 9 * Cin * Cout,
          nbytes=float(x.R) * (x.C + Cout) * 2 + Cout * 9 * Cin * 2)
    return y, stats


def gridconv3x3_fused(x: Grid, wt, Cout, bias=None, add: "Grid" = None, relu=False) -> Grid:
    """y = [relu](conv3x3(x) + bias [+ add]) on the grid (a frozen ResNet BasicBlock's convolutions, BatchNorm folded); wt bf16 [Cout][9 Cin]"""
    _require_cuda(x.rows, wt, bias)
    if wt.dtype != torch.bfloat16 or tuple(wt.shape) != (Cout, 9 * x.C) or not wt.is_contiguous():
        raise _L.HulcKernelError("gridconv3x3_fused: weights are bf16 [Cout][9 * Cin]")
    if add is not None and (add.C != Cout or (add.N, add.H, add.W) != (x.N, x.H, x.W)):
        raise _L.HulcKernelError("gridconv3x3_fused: the residual branch must be a grid tensor shaped like the output")
    y = Grid(x.N, x.H, x.W, Cout, x.rows.device)
    _call("hulc_gridconv3x3_fused", x.t, _l(x.C), wt, y.t, _l(Cout), _i(x.N), _i(x.H), _i(x.W), _i(x.C), _i(Cout), bias,
          (add.t if add is not None else None), _l(add.C if add is not None else 0), _i(1 if relu else 0),
          key=("gridconv3x3", x.N, x.H, x.W, x.C, Cout), flops=2.0 * x.R * 9 * x.C * Cout, nbytes=float(x.R) * (x.C + Cout * (2 if add is not None else 1)) * 2 + Cout * 9 * x.C * 2)
    return y


def grid_from_nhwc(x) -> Grid:
    """dense (N, H, W, C) bf16 -> grid tensor (border rows zero)"""
    _require_contiguous(x=x)
    N, H, W, C = x.shape
    g = Grid(N, H, W, C, x.device)
    _call("hulc_grid_from_nhwc", x, _i(N), _i(H), _i(W), _i(C), g.t, _l(C))
    return g


_bn_ctr = {}


def _bn_counters(device):
    """self-resetting arrival counters of the BatchNorm reductions, one set per (device, stream)"""
    key = (device, _stream())
    ctr = _bn_ctr.get(key)
    if ctr is None:
        ctr = _bn_ctr[key] = torch.zeros(64, dtype=torch.int32, device=device)
    return ctr


def grid_bn_finalize(stats, N, H, W, C, gamma, beta, run_mean=None, run_var=None, eps=1e-5, momentum=0.1):
    bn = torch.empty(4 + 64, C, dtype=torch.float32, device=stats.device)          # the table, then the row groups' intermediate sums
    nb = (N * (H + 2) * (W + 2) + 127) // 128
    ctr = _bn_counters(stats.device)
    _call("hulc_grid_bn_finalize", stats, _i(nb), _i(C), _l(N * H * W), gamma, beta, _f(eps), _f(momentum), bn, run_mean, run_var, bn[4:], ctr)
    return bn[:4]


def grid_bn_relu_fwd(y: Grid, bn) -> Grid:
    out = Grid(y.N, y.H, y.W, y.C, y.rows.device)
    _call("hulc_grid_bn_relu_fwd", y.t, _l(y.C), bn, _i(y.N), _i(y.H), _i(y.W), _i(y.C), out.t, _l(out.C), key=("grid_bn_relu_fwd", y.N, y.H, y.W, y.C),
          nbytes=float(y.R) * y.C * 4)
    return out


def grid_bn_relu_bwd(dout: Grid, out: Grid, y: Grid, bn, dgamma, dbeta, accumulate=False) -> Grid:
    dz = Grid(y.N, y.H, y.W, y.C, y.rows.device)
    lib = _L.load()
    lib.hulc_grid_bn_bwd_workspace.restype = _c.c_long
    ws = _ws(lib.hulc_grid_bn_bwd_workspace(_i(y.N), _i(y.H), _i(y.W), _i(y.C)), y.rows.device)
    _call("hulc_grid_bn_relu_bwd", dout.t, _l(dout.C), out.t, _l(out.C), y.t, _l(y.C), bn, _i(y.N), _i(y.H), _i(y.W), _i(y.C), dz.t, _l(dz.C), dgamma, dbeta,
          _i(accumulate), ws, _bn_counters(y.rows.device), key=("grid_bn_relu_bwd", y.N, y.H, y.W, y.C),      # (one shape per table row: ten layers of
          nbytes=float(y.R) * y.C * 14)                                                                         # different sizes must not be averaged)
    return dz


def grid_upcat_fwd(x, xs, g, skip, ss, N, Ho, Wo, s, Cx, Cs) -> Grid:
    """x / skip: (tensor at pixel (0,0,0), stride_n, stride_y, stride_x) strided bf16 maps; g (N, Cx) fp32 or None"""
    out = Grid(N, Ho, Wo, Cx + Cs, x.device)
    _call("hulc_grid_upcat_fwd", x, _l(xs[0]), _l(xs[1]), _l(xs[2]), g, skip, _l(ss[0] if skip is not None else 0), _l(ss[1] if skip is not None else 0),
          _l(ss[2] if skip is not None else 0), _i(N), _i(Ho), _i(Wo), _i(s), _i(Cx), _i(Cs), out.t, key=("grid_upcat_fwd", N, Ho, Wo, Cx + Cs),
          nbytes=float(out.R) * out.C * 4)
    return out


def grid_upcat_bwd(dX: Grid, x, xs, g, N, Hi, Wi, s, Cx, want_dsmall=True, want_dg=True):
    dsmall = Grid(N, Hi, Wi, Cx, dX.rows.device) if want_dsmall else None
    dg = torch.empty(N, Cx, dtype=torch.float32, device=dX.rows.device) if want_dg else None
    _call("hulc_grid_upcat_bwd", dX.t, _l(dX.C), x, _l(xs[0]), _l(xs[1]), _l(xs[2]), g, _i(N), _i(Hi), _i(Wi), _i(s), _i(Cx),
          (dsmall.t if dsmall is not None else None), dg, _i(0), key=("grid_upcat_bwd", N, Hi, Wi, Cx), nbytes=float(dX.R) * Cx * 2)
    return dsmall, dg


def pixel_ce_fwd(logit0, p0, N, H, W):
    lse = torch.empty(N, dtype=torch.float32, device=logit0.device)
    picked = torch.empty(N, dtype=torch.float32, device=logit0.device)
    _call("hulc_pixel_ce_fwd", logit0, p0, _i(N), _i(H), _i(W), lse, picked)
    return lse, picked


def head_conv_fwd(x: Grid, w, bias):
    """the one-channel head on the grid: -> logit0 fp32 [R] (bias added on the pixels, zero on the border)"""
    out0 = torch.empty(x.R, dtype=torch.float32, device=x.rows.device)
    _call("hulc_head_conv_fwd", x.t, _l(x.C), w, bias, _i(x.N), _i(x.H), _i(x.W), _i(x.C), out0, flops=2.0 * x.R * 9 * x.C, nbytes=float(x.R) * (x.C * 2 + 4))
    return out0


def pixel_ce_bwd_rows(logit0, p0, lse, upstream, N, H, W):
    g = torch.empty(N * (H + 2) * (W + 2), dtype=torch.float32, device=logit0.device)
    _call("hulc_pixel_ce_bwd_rows", logit0, p0, lse, upstream, _i(N), _i(H), _i(W), g)
    return g


def head_conv_dgrad(g, w, N, H, W, C) -> Grid:
    dx = Grid(N, H, W, C, g.device)
    _call("hulc_head_conv_dgrad", g, w, _i(N), _i(H), _i(W), _i(C), dx.t, _l(C), flops=2.0 * dx.R * 9 * C, nbytes=float(dx.R) * (C * 2 + 4))
    return dx


_head_ws = {}


def head_conv_wgrad(x: Grid, g, dw, accumulate=False):
    lib = _L.load()
    lib.hulc_head_conv_wgrad_workspace.restype = _c.c_long
    need = int(lib.hulc_head_conv_wgrad_workspace(_i(x.N), _i(x.H), _i(x.W), _i(x.C)))
    key = (x.rows.device, _stream())
    ws = _head_ws.get(key)
    if ws is None or ws.numel() * 4 < need:
        ws = torch.empty(need // 4 + 16, dtype=torch.float32, device=x.rows.device)
        if not torch.cuda.is_current_stream_capturing():
            _head_ws[key] = ws
    _call("hulc_head_conv_wgrad", x.t, _l(x.C), g, _i(x.N), _i(x.H), _i(x.W), _i(x.C), dw, _i(1 if accumulate else 0), ws, flops=2.0 * x.R * 9 * x.C,
          nbytes=float(x.R) * (x.C * 2 + 4))


def depth_nll_fwd(x, w_mu, b_mu, w_sigma, b_sigma, target):
    """-> (mu (B, 1), sigma (B, 1), log_sigma (B,), loss ()) of hulc_depth_nll_fwd"""
    B, D = x.shape
    mu = torch.empty(B, 1, dtype=torch.float32, device=x.device)
    sigma = torch.empty(B, 1, dtype=torch.float32, device=x.device)
    ls = torch.empty(B, dtype=torch.float32, device=x.device)
    loss = torch.empty((), dtype=torch.float32, device=x.device)
    _call("hulc_depth_nll_fwd", x, _i(B), _i(D), w_mu, b_mu, w_sigma, b_sigma, target, mu, sigma, ls, loss)
    return mu, sigma, ls, loss


def depth_nll_bwd(x, w_mu, w_sigma, mu, sigma, ls, target, gout, dx, dw_mu, db_mu, dw_sigma, db_sigma, accumulate_mask=0):
    B, D = x.shape
    _call("hulc_depth_nll_bwd", x, _i(B), _i(D), w_mu, w_sigma, mu, sigma, ls, target, gout, dx, dw_mu, db_mu, dw_sigma, db_sigma, _i(accumulate_mask))


def pixel_ce_bwd(logit0, p0, lse, upstream, N, H, W, C) -> Grid:
    dz = Grid(N, H, W, C, logit0.device)
    _call("hulc_pixel_ce_bwd", logit0, p0, lse, upstream, _i(N), _i(H), _i(W), _i(C), dz.t)
    return dz
