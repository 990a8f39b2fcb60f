"""ctypes binding of libstar_nugget.so (C ABI declared in include/star_nugget.h).

The library is the ONLY implementation of the hot path: if it is missing or a call fails, the
caller gets an exception -- there is no CPU or eager-PyTorch fallback.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libstar_nugget.so")

_lib = None


class StarNuggetError(RuntimeError):
    pass


class XAttnShape(C.Structure):
    _fields_ = [("B", C.c_int), ("h", C.c_int), ("L", C.c_int), ("Kn", C.c_int), ("hd", C.c_int),
                ("q_sb", C.c_int64), ("q_sh", C.c_int64), ("q_sr", C.c_int64),
                ("k_sb", C.c_int64), ("k_sh", C.c_int64), ("k_sr", C.c_int64),
                ("v_sb", C.c_int64), ("v_sh", C.c_int64), ("v_sr", C.c_int64),
                ("o_sb", C.c_int64), ("o_sh", C.c_int64), ("o_sr", C.c_int64),
                ("bias_len", C.c_int), ("ilk_diag", C.c_int), ("scale", C.c_float),
                ("dropout_p", C.c_float), ("seed", C.c_uint64), ("offset", C.c_uint64), ("rng_state", C.c_void_p)]


_P = C.c_void_p
_I = C.c_int
_I64 = C.c_int64
_F = C.c_float
_SZ = C.c_size_t

# name -> (restype, argtypes); mirrors include/star_nugget.h one to one
SIGNATURES = {
    "star_version": (_I, []),
    "star_last_error": (C.c_char_p, []),
    "star_test_override": (_I, [_I, _I]),
    "star_topk_select": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P]),
    "star_nugget_type_table": (_I, [_P, _P, _I, _I, _I, _P, _P, _P]),
    "star_gather_rows_fwd": (_I, [_P, _P, _P, _I, _I, _I, _I64, _P]),
    "star_gather_rows_bwd": (_I, [_P, _P, _P, _I, _I, _I, _I, _I, _P]),
    "star_type_embed_add": (_I, [_P, _P, _P, _I, _I64, _I, _P, _P]),
    "star_type_embed_grad_workspace": (C.c_size_t, [_I64, _I, _I]),
    "star_type_embed_grad": (_I, [_P, _P, _I, _I64, _I, _P, _P, C.c_size_t, _P]),
    "star_diag_fanout_fwd": (_I, [_P, _P, _I, _I64, _I, _P, _P]),
    "star_xattn_dbias_sum": (_I, [C.POINTER(C.c_void_p), _I, _I, _I, _I, _I, _P, _P]),
    "star_diag_fanout_bwd": (_I, [C.POINTER(C.c_void_p), _I, _P, _I64, _I, _P, _P]),
    "star_cif_quantity_fwd": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _P]),
    "star_cif_quantity_bwd": (_I, [_P, _P, _P, _F, _I, _I, _I, _P, _P]),
    "star_cif_resize": (_I, [_P, _P, _F, _I, _I, _I, _P, _P, _P]),
    "star_cif_resize_bwd": (_I, [_P, _P, _P, _P, _I, _I, _P, _P]),
    "star_gather_layernorm_fwd": (_I, [_P, _I, _P, _P, _P, C.c_float, _I, _I, _I, _I, _P, _P, _P, _P]),
    "star_gather_layernorm_bwd_workspace": (_SZ, [_I, _I, _I]),
    "star_gather_layernorm_bwd": (_I, [_P, _I, _P, _P, _P, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P, _SZ, _P]),
    "star_cif_fire_scan": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "star_cif_segment": (_I, [_P, _P, _P, _F, _I, _I, _I, _I] + [_P] * 12 + [_P]),
    "star_cif_fire_step": (_I, [_P, _P, _P, _I, _I, _P, _P, _I, _P, _P, _P, _P]),
    "star_cif_stream_push": (_I, [_P, _P, _I, _P, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "star_cif_block_mask": (_I, [_P, _P, _P, _P, _I, _I, _I, _P, _P]),
    "star_cif_weights": (_I, [_P, _F, _I, _I, _I, _P, _P, _P, _P, _P, _P]),
    "star_cif_aggregate_fwd": (_I, [_P, _I, _P, _P, _P, _I, _I, _I, _I, _P, _P]),
    "star_cif_aggregate_bwd": (_I, [_P, _P, _I, _P, _P, _I, _I, _I, _I, _P, _P, _P, _P]),
    "star_scorer_fwd_workspace": (_SZ, [_I64, _I]),
    "star_scorer_relu_bits_bytes": (_SZ, [_I64, _I]),
    "star_scorer_fwd": (_I, [_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P, _SZ, _P]),
    "star_scorer_fwd_ragged": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P, _I, _P, _P, _P, _SZ, _P]),
    "star_scorer_fwd_compact": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I, _I, _I, _P, _P, _P, _SZ, _P]),
    "star_scorer_fwd_static": (_I, [_P, _P, _P, _P, _P, _I64, _I, _P, _P, _P]),
    "star_scorer_bwd_workspace": (_SZ, [_I64, _I]),
    "star_scorer_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _P, _P, _P, _P, _P, _I, _P, _SZ, _P, _P,
                             C.POINTER(C.c_void_p)]),
    "star_scorer_bwd_mapped": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I64, _I64, _I, _P, _P, _P, _P, _P, _I, _P, _SZ,
                                    _P, _P, C.POINTER(C.c_void_p)]),
    "star_cast_transpose_bf16": (_I, [_P, _I, _I, _P, _P, _P, _P]),
    "star_gemm_nt_bf16": (_I, [_P, _I64, _P, _I64, _P, _I64, _I, _I, _P, _I64, _P]),
    "star_gemm_tn_batched_bf16": (_I, [C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, _I, _I64, _I, _I, _I64, _I64, _P, _P]),
    "star_transpose_list_bf16": (_I, [C.POINTER(C.c_void_p), _I, _I, _I, _I, _P, _P]),
    "star_colsum_workspace": (_SZ, [_I64, _I]),
    "star_colsum": (_I, [_P, _I, _I64, _I, _I64, _P, _P, _SZ, _P]),
    "star_colsum_batched": (_I, [C.POINTER(C.c_void_p), _I, _I, _I64, _I, _I64, _P, _P, _SZ, _P]),
    "star_xattn_fwd": (_I, [_P, _P, _P, _P, _P, C.POINTER(XAttnShape), _P, _P, _P, _P]),
    "star_xattn_bwd": (_I, [_P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(XAttnShape), _P, _P, _P, _P, _P, _P, _I, _P]),
}


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise StarNuggetError(
                f"{LIB_PATH} not found: build it with `python star_amd/csrc/build.py` "
                "(there is no fallback implementation)")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        msg = lib().star_last_error().decode("utf-8", "replace")
        if rc == -1:
            raise ValueError(f"{what}: {msg}")
        raise StarNuggetError(f"{what} failed ({rc}): {msg}")


def ptr(t):
    """Device pointer of a torch tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_ptr():
    import torch
    return torch.cuda.current_stream().cuda_stream
