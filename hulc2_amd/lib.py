"""ctypes loader for libhulc2_amd.so — the only way compute enters the product path.

There is no CPU or eager-PyTorch fallback: if the HIP library is missing or a kernel rejects a call,
the caller gets an exception (the oracle under oracle/ is test infrastructure and is never imported
from here).
"""
import ctypes
import os
from pathlib import Path

import torch  # noqa: F401  -- must be imported BEFORE the CDLL below: torch ships its own libamdhip64; loading ours first
#                              would bring a second HIP runtime into the process that owns no device context

_LIB_PATH = Path(__file__).resolve().parent / "libhulc2_amd.so"
if os.environ.get("HULC_LIB"):          # A/B measurements: another BUILD of the same library (e.g. the previous commit's), same ABI
    _LIB_PATH = Path(os.environ["HULC_LIB"]).resolve()
_lib = None


class HulcKernelError(RuntimeError):
    pass


class GemmDesc(ctypes.Structure):
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p),
        ("bias", ctypes.c_void_p), ("add", ctypes.c_void_p), ("mask", ctypes.c_void_p),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("lda", ctypes.c_long), ("ldb", ctypes.c_long), ("ldc", ctypes.c_long),
        ("ld_add", ctypes.c_long), ("ld_mask", ctypes.c_long),
        ("a_dtype", ctypes.c_int), ("b_dtype", ctypes.c_int), ("c_dtype", ctypes.c_int),
        ("add_dtype", ctypes.c_int), ("mask_dtype", ctypes.c_int),
        ("a_kmajor", ctypes.c_int), ("b_kmajor", ctypes.c_int),
        ("relu", ctypes.c_int), ("accumulate", ctypes.c_int),
        ("alpha", ctypes.c_float), ("mask_scale", ctypes.c_float), ("drop_p", ctypes.c_float),
        ("drop_seed", ctypes.c_ulonglong),
        ("compute", ctypes.c_int),
        ("ws", ctypes.c_void_p), ("ws_bytes", ctypes.c_long),
        ("seed_dev", ctypes.c_void_p),
        ("rowsum_a", ctypes.c_void_p), ("rowsum_accumulate", ctypes.c_int),
    ]


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load the shared library once; raise loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise HulcKernelError(
            f"{_LIB_PATH} is missing: build it with `python -m hulc2_amd.build` "
            "(hipcc --offload-arch=gfx950). hulc2_amd has no non-HIP fallback."
        )
    lib = ctypes.CDLL(os.fspath(_LIB_PATH))
    lib.hulc_last_error.restype = ctypes.c_char_p
    lib.hulc_abi_version.restype = ctypes.c_int
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().hulc_last_error().decode(errors="replace")
        raise HulcKernelError(f"{what} failed with code {rc}: {msg}")


class ConvDesc(ctypes.Structure):
    _fields_ = [
        ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int), ("Cout", ctypes.c_int),
        ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("stride", ctypes.c_int),
        ("x_nchw", ctypes.c_int),
        ("x_dtype", ctypes.c_int), ("y_dtype", ctypes.c_int), ("w_dtype", ctypes.c_int),
        ("relu", ctypes.c_int),
        ("compute", ctypes.c_int),
        ("dw_oihw", ctypes.c_int), ("dw_accumulate", ctypes.c_int),
        ("x_u8_nhwc", ctypes.c_int), ("aug_pad", ctypes.c_int), ("aug_shift", ctypes.c_void_p), ("frame_index", ctypes.c_void_p),
        ("relu_bits", ctypes.c_void_p),
        ("w_lo", ctypes.c_void_p),
        ("x2", ctypes.c_void_p), ("n_split", ctypes.c_int),
        ("x_slot", ctypes.c_void_p), ("x2_slot", ctypes.c_void_p),
        ("y_bf16", ctypes.c_void_p),
    ]


class RnnWaveDesc(ctypes.Structure):
    """mirror of hulc_rnn_wave_desc (include/hulc2_amd.h)"""
    _fields_ = [
        ("z", ctypes.c_void_p), ("z_step", ctypes.c_long),
        ("wA", ctypes.c_void_p), ("wB1", ctypes.c_void_p), ("wB2", ctypes.c_void_p),
        ("ldA", ctypes.c_long), ("ldB1", ctypes.c_long), ("ldB2", ctypes.c_long),
        ("tA", ctypes.c_int), ("tB1", ctypes.c_int), ("tB2", ctypes.c_int),
        ("add1", ctypes.c_void_p), ("add1_step", ctypes.c_long), ("ld_add1", ctypes.c_long),
        ("bias1a", ctypes.c_void_p), ("bias1b", ctypes.c_void_p), ("bias2a", ctypes.c_void_p), ("bias2b", ctypes.c_void_p),
        ("mask1", ctypes.c_void_p), ("mask1_step", ctypes.c_long), ("ld_mask1", ctypes.c_long),
        ("mask2", ctypes.c_void_p), ("mask2_step", ctypes.c_long), ("ld_mask2", ctypes.c_long),
        ("relu", ctypes.c_int), ("S", ctypes.c_int), ("B", ctypes.c_int), ("H", ctypes.c_int), ("mirror_t", ctypes.c_int),
        ("err_sticky", ctypes.c_void_p),
        ("add1c", ctypes.c_void_p), ("ld_add1c", ctypes.c_long),
        ("zero_edges", ctypes.c_int),
    ]


class MixDesc(ctypes.Structure):
    _fields_ = [
        ("T", ctypes.c_int), ("A", ctypes.c_int), ("n_mix", ctypes.c_int), ("num_classes", ctypes.c_int), ("nseg", ctypes.c_int),
        ("ld", ctypes.c_long),
        ("log_scale_min", ctypes.c_float), ("gripper_alpha", ctypes.c_float),
        ("act_min", ctypes.c_void_p), ("act_max", ctypes.c_void_p),
        ("time_major_B", ctypes.c_int),
    ]


class TxlAttnDesc(ctypes.Structure):
    """mirror of hulc_txl_attn_desc (include/hulc2_amd.h)"""
    _fields_ = [
        ("x", ctypes.c_void_p),
        ("Wqkv", ctypes.c_void_p), ("Wo", ctypes.c_void_p), ("WqkvT", ctypes.c_void_p), ("WoT", ctypes.c_void_p),
        ("bqkv", ctypes.c_void_p), ("bo", ctypes.c_void_p), ("gamma", ctypes.c_void_p), ("beta", ctypes.c_void_p),
        ("eps", ctypes.c_float),
        ("B", ctypes.c_int), ("S", ctypes.c_int), ("H", ctypes.c_int), ("E", ctypes.c_int),
        ("drop_p", ctypes.c_float),
        ("seed_attn", ctypes.c_ulonglong), ("seed_ln", ctypes.c_ulonglong),
        ("seed_dev", ctypes.c_void_p),
        ("y", ctypes.c_void_p), ("pre", ctypes.c_void_p), ("mean", ctypes.c_void_p), ("rstd", ctypes.c_void_p),
        ("ctx", ctypes.c_void_p),
        ("dy", ctypes.c_void_p), ("dy_slab", ctypes.c_void_p),
        ("n_slab", ctypes.c_int),
        ("slab_stride", ctypes.c_long),
        ("dx", ctypes.c_void_p),
        ("d_o", ctypes.c_void_p), ("dqkv", ctypes.c_void_p),
        ("ln_partial", ctypes.c_void_p),
    ]


class TxlBlockLayer(ctypes.Structure):
    """mirror of hulc_txl_block_layer (include/hulc2_amd.h)"""
    _fields_ = ([(n, ctypes.c_void_p) for n in ("Wqkv", "Wo", "W1", "W2", "WqkvT", "WoT", "W1T", "W2T", "W1p", "W2p", "W2Tp", "W1Tp",
                                                "Wqkv_lo", "Wo_lo", "W1p_lo", "W2p_lo", "bqkv", "bo", "b1", "b2", "g1", "be1", "g2", "be2")]
                + [(n, ctypes.c_ulonglong) for n in ("seed_attn", "seed_ln1", "seed_ffn", "seed_ln2")]
                + [(n, ctypes.c_void_p) for n in ("x", "y1", "pre1", "mean1", "rstd1", "ctx", "y2", "pre2", "mean2", "rstd2",
                                                  "d_o", "dqkv", "df", "h", "dh", "lnp1", "lnp2")])


TXL_MAX_LAYERS = 4


class TxlBlockDesc(ctypes.Structure):
    """mirror of hulc_txl_block_desc (include/hulc2_amd.h)"""
    _fields_ = [
        ("L", ctypes.c_int), ("B", ctypes.c_int), ("S", ctypes.c_int), ("H", ctypes.c_int), ("E", ctypes.c_int), ("FF", ctypes.c_int),
        ("eps", ctypes.c_float), ("drop_p", ctypes.c_float),
        ("seed_pos", ctypes.c_ulonglong),
        ("seed_dev", ctypes.c_void_p),
        ("emb", ctypes.c_void_p), ("pos", ctypes.c_void_p), ("pos_ids", ctypes.c_void_p),
        ("pooled", ctypes.c_void_p), ("dpooled", ctypes.c_void_p), ("demb", ctypes.c_void_p),
        ("ws", ctypes.c_void_p), ("exclusive", ctypes.c_int), ("err_sticky", ctypes.c_void_p),
        ("layers", TxlBlockLayer * TXL_MAX_LAYERS),
    ]


class WgradItem(ctypes.Structure):
    """mirror of hulc_wgrad_item (include/hulc2_amd.h)"""
    _fields_ = [
        ("A", ctypes.c_void_p), ("B", ctypes.c_void_p), ("C", ctypes.c_void_p), ("rowsum", ctypes.c_void_p),
        ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int),
        ("lda", ctypes.c_int), ("ldb", ctypes.c_int), ("ldc", ctypes.c_int),
        ("a_dtype", ctypes.c_int), ("b_dtype", ctypes.c_int),
        ("accumulate", ctypes.c_int), ("rowsum_accumulate", ctypes.c_int),
        ("col_perm", ctypes.c_int), ("col_mul", ctypes.c_int), ("store_rows", ctypes.c_int), ("conv_taps_wp", ctypes.c_int),
    ]


class MlpChainLayer(ctypes.Structure):
    """mirror of hulc_mlp_chain_layer (include/hulc2_amd.h)"""
    _fields_ = [
        ("W", ctypes.c_void_p), ("ldw", ctypes.c_long),
        ("W_lo", ctypes.c_void_p),
        ("bias", ctypes.c_void_p),
        ("mask", ctypes.c_void_p), ("ld_mask", ctypes.c_long), ("mask_scale", ctypes.c_float),
        ("out", ctypes.c_void_p), ("ld_out", ctypes.c_long),
        ("N", ctypes.c_int), ("relu", ctypes.c_int),
    ]


class MlpChainDesc(ctypes.Structure):
    """mirror of hulc_mlp_chain_desc (include/hulc2_amd.h)"""
    _fields_ = [
        ("nl", ctypes.c_int), ("M", ctypes.c_int), ("K0", ctypes.c_int),
        ("x0", ctypes.c_void_p), ("ld_x0", ctypes.c_long),
        ("layers", MlpChainLayer * 8),
    ]
