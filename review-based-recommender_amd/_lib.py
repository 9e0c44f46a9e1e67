"""ctypes binding of librbr_hip.so (the C ABI of include/rbr_hip.h).

No fallback: `lib()` raises if the shared library is missing or lacks a symbol, and the
helpers below raise if a tensor is not a contiguous HIP tensor of the expected dtype.
"""
from __future__ import annotations

import ctypes as C
import os
import threading
from typing import Optional, Sequence

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
# RBR_LIB_PATH: another build of the same library (dev: A/B measurements of two builds in one gpurun call, tools/gpu_ab2.sh)
LIB_PATH = os.environ.get("RBR_LIB_PATH") or os.path.join(HERE, "csrc", "librbr_hip.so")

RBR_MAX_WIDTHS = 8
PAD_SAME, PAD_VALID = 0, 1
ACT_RELU, ACT_TANH = 0, 1
PROD_F32, PROD_BF16X3, PROD_BF16X2, PROD_BF16 = 0, 1, 2, 3     # rbr_set_prod_precision (RBR_PROD_* of rbr_hip.h)
PROD_PRECISIONS = {"f32": PROD_F32, "bf16x3": PROD_BF16X3, "bf16x2": PROD_BF16X2, "bf16": PROD_BF16}

c_f32p = C.c_void_p
c_i64p = C.c_void_p
c_u8p = C.c_void_p
c_i32p = C.c_void_p
c_stream = C.c_void_p


class TextCNNDesc(C.Structure):
    _fields_ = [
        ("n_docs", C.c_int32), ("L", C.c_int32), ("D", C.c_int32), ("V", C.c_int32),
        ("n_widths", C.c_int32),
        ("kz", C.c_int32 * RBR_MAX_WIDTHS), ("ch", C.c_int32 * RBR_MAX_WIDTHS),
        ("pad_mode", C.c_int32), ("act", C.c_int32), ("padding_idx", C.c_int32), ("flags", C.c_int32),
    ]


class HeadParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("Wu", "bu", "Eu", "Wi", "bi", "Ei", "h", "g", "ub", "ib")]


class HeadGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dWu", "dbu", "dEu", "dWi", "dbi", "dEi", "dh", "dg", "dub", "dib")]


class AttnParams(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("W_rv", "W_id", "h", "b1", "b2", "ebd")]


class IdSet(C.Structure):
    _fields_ = [("inp", C.c_void_p), ("out", C.c_void_p), ("n", C.c_int64), ("limit", C.c_int64), ("replace", C.c_int64)]


MAX_ID_SETS = 8


class RowGrad(C.Structure):       # rbr_row_grad
    _fields_ = [("tensor", C.c_int32), ("V", C.c_int32), ("D", C.c_int32), ("row_of_token", C.c_void_p), ("rows", C.c_void_p),
                ("sq_part", C.c_void_p), ("n_sq", C.c_int32)]


G_BUILD, G_PRODUCT, G_ACCUMULATE, G_ROWS, G_ZEROED = 1, 2, 4, 8, 16      # RBR_G_* of rbr_hip.h


def conv_gate_split(k: int) -> int:
    """RBR_CONV_GATE_SPLIT(k) of rbr_hip.h."""
    return (int(k) & 0xF) << 8


class AttnGrads(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("dW_rv", "dW_id", "dh", "db1", "db2", "debd")]


_PP = C.POINTER(C.c_void_p)
_DESC = C.POINTER(TextCNNDesc)
i32 = C.c_int32

# name -> (restype, argtypes); mirrors include/rbr_hip.h one to one
SIGNATURES = {
    "rbr_version": (C.c_int, []),
    "rbr_last_error": (C.c_char_p, []),
    "rbr_pair_begin": (C.c_int, []),
    "rbr_pair_next": (C.c_int, []),
    "rbr_pair_end": (C.c_int, [C.POINTER(i32), C.POINTER(i32)]),
    "rbr_pair_abort": (None, []),
    "rbr_textcnn_packed_floats": (C.c_size_t, [_DESC]),
    "rbr_textcnn_partial_elems": (C.c_size_t, [_DESC]),
    "rbr_textcnn_pack": (C.c_int, [_DESC, _PP, c_f32p, c_stream]),
    "rbr_textcnn_fwd_ws_bytes": (C.c_size_t, [_DESC]),
    "rbr_set_conv_mode": (None, [i32]),
    "rbr_set_prod_precision": (None, [i32]),
    "rbr_get_prod_precision": (i32, []),
    "rbr_textcnn_desc_stamp": (None, [_DESC]),
    "rbr_set_b16_storage": (None, [i32]),
    "rbr_textcnn_conv_fwd": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, _PP, c_f32p, c_f32p, c_i32p, C.c_void_p,
                                       c_stream]),
    "rbr_textcnn_prod_prepare": (C.c_int, [_DESC, c_i64p, c_u8p, _PP, c_i32p, C.c_void_p, c_stream]),
    "rbr_textcnn_prod_table": (C.c_int, [_DESC, c_f32p, C.c_void_p, c_stream]),
    "rbr_textcnn_prod_pool": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_i32p, C.c_void_p, c_stream]),
    "rbr_textcnn_pool_finalize": (C.c_int, [_DESC, c_f32p, c_i32p, _PP, c_f32p, c_i32p, c_stream]),
    "rbr_textcnn_bwd_ws_floats": (C.c_size_t, [_DESC]),
    "rbr_textcnn_bwd_dw": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, _PP, _PP, c_f32p,
                                     c_stream]),
    "rbr_textcnn_bwd_dtable": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, c_f32p,
                                         c_f32p, c_stream]),
    "rbr_textcnn_bwd_prod_ws_bytes": (C.c_size_t, [_DESC]),
    "rbr_textcnn_bwd_dtable_prod": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_void_p,
                                              c_f32p, c_f32p, c_stream]),
    "rbr_textcnn_bwd_dtable_prod_acc": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_void_p,
                                                  c_f32p, c_f32p, c_stream]),
    "rbr_textcnn_bwd_g_build": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_void_p, c_f32p,
                                          c_stream]),
    "rbr_textcnn_bwd_g_product": (C.c_int, [_DESC, C.c_void_p, C.c_void_p, c_f32p, c_stream]),
    "rbr_textcnn_bwd_dtable_prod_ex": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_i32p, c_f32p, C.c_void_p, C.c_void_p,
                                                 c_f32p, c_f32p, c_f32p, i32, c_stream]),
    "rbr_textcnn_row_grad_partials": (C.c_size_t, [_DESC]),
    "rbr_textcnn_token_list": (C.c_int, [_DESC, C.c_void_p, _PP, _PP, _PP, C.POINTER(i32)]),
    "rbr_textcnn_prod_prepare_ids": (C.c_int, [_DESC, i32, C.POINTER(IdSet), C.c_void_p, c_i64p, c_u8p, _PP, c_i32p, C.c_void_p,
                                               c_stream]),
    "rbr_dedup_fold_rows": (C.c_int, [i32, i32, c_i64p, c_f32p, c_stream]),
    "rbr_pair_head_fwd_pool": (C.c_int, [_DESC, c_f32p, c_i32p, _PP, c_f32p, c_i32p, c_i64p, i32, c_i64p, c_i64p, C.POINTER(HeadParams),
                                         c_f32p, C.c_float, C.c_uint64, C.c_void_p, c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p,
                                         c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_stream]),
    "rbr_clip_adam_step_rows": (C.c_int, [i32, _PP, _PP, _PP, _PP, C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float,
                                          C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, C.POINTER(RowGrad), c_stream]),
    "rbr_row_grad_to_dense": (C.c_int, [i32, i32, c_i32p, c_f32p, c_f32p, c_stream]),
    "rbr_textcnn_bwd_dtable_list_ws_bytes": (C.c_size_t, [_DESC]),
    "rbr_textcnn_bwd_dtable_list": (C.c_int, [_DESC, c_i64p, c_u8p, _PP, c_f32p, c_i32p, c_f32p, C.c_void_p, c_f32p, c_stream]),
    "rbr_textcnn_bwd_dw_from_g_ws_floats": (C.c_size_t, [_DESC]),
    "rbr_textcnn_bwd_dw_from_g": (C.c_int, [_DESC, c_f32p, c_f32p, c_f32p, C.c_void_p, C.c_void_p, _PP, _PP, c_f32p, c_stream]),
    "rbr_textcnn_taps_count": (C.c_size_t, [_DESC]),
    "rbr_textcnn_bwd_taps": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_i32p, c_f32p, c_i32p, c_f32p, c_stream]),
    "rbr_textcnn_dtable_from_taps_ws_bytes": (C.c_size_t, [_DESC, i32]),
    "rbr_textcnn_dtable_from_taps": (C.c_int, [_DESC, i32, c_i32p, c_f32p, _PP, C.c_void_p, c_f32p, c_stream]),
    "rbr_textcnn_taps_owner_rows": (i32, [_DESC, i32]),
    "rbr_textcnn_dtable_from_taps_owner": (C.c_int, [_DESC, i32, i32, c_i32p, c_f32p, _PP, C.c_void_p, c_f32p, c_i32p, c_stream]),
    "rbr_textcnn_bwd": (C.c_int, [_DESC, c_i64p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_i32p, c_f32p, _PP, _PP,
                                  c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_datt_local_gate_prod_ws_bytes": (C.c_size_t, [i32, i32, i32, i32, i32]),
    "rbr_datt_local_gate_fwd_prod": (C.c_int, [i32, i32, i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, C.c_void_p,
                                               C.c_void_p, c_stream]),
    "rbr_datt_local_gate_bwd_prod": (C.c_int, [i32, i32, i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p,
                                               c_f32p, c_f32p, C.c_void_p, C.c_void_p, i32, c_stream]),
    "rbr_review_bag_fwd": (C.c_int, [i32, i32, i32, c_i64p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_review_bag_bwd_ws_bytes": (C.c_size_t, [i32, i32]),
    "rbr_review_bag_bwd": (C.c_int, [i32, i32, i32, i32, c_i64p, c_u8p, c_f32p, c_f32p, c_f32p, i32, c_f32p, C.c_void_p,
                                     c_stream]),
    "rbr_additive_attn_fwd": (C.c_int, [i32, i32, i32, i32, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                        c_f32p, c_stream]),
    "rbr_additive_attn_bwd_ws_floats": (C.c_size_t, [i32, i32, i32, i32]),
    "rbr_additive_attn_bwd": (C.c_int, [i32, i32, i32, i32, c_f32p, c_u8p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p,
                                        c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_clip_adam_ws_floats": (C.c_size_t, []),
    "rbr_clip_adam_step": (C.c_int, [i32, _PP, _PP, _PP, _PP, C.POINTER(C.c_int64), C.c_float, C.c_float, C.c_float,
                                     C.c_float, C.c_float, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_pair_head_fwd": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_i64p, c_i64p, C.POINTER(HeadParams), c_f32p,
                                    c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_pair_head_fwd_train": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_i64p, c_i64p, C.POINTER(HeadParams), C.c_float,
                                          C.c_uint64, C.c_void_p, c_f32p, c_f32p, C.c_int64, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_pair_head_bwd_ws_floats": (C.c_size_t, [i32, i32]),
    "rbr_pair_head_bwd": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_i64p, c_i64p, C.POINTER(HeadParams), c_f32p,
                                    c_f32p, c_f32p, c_f32p, i32, i32, C.POINTER(HeadGrads), c_f32p, c_f32p, c_f32p,
                                    c_stream]),
    "rbr_dropout_multiplier": (C.c_int, [C.c_int64, C.c_float, C.c_uint64, C.c_void_p, c_f32p, c_stream]),
    "rbr_mse_loss_fwd": (C.c_int, [C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_pair_dot_fwd": (C.c_int, [i32, i32, c_f32p, c_f32p, c_stream]),
    "rbr_pair_dot_bwd": (C.c_int, [i32, i32, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_block_cat": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_block_split": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_mse_loss_bwd": (C.c_int, [C.c_int64, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_review_attn_fwd": (C.c_int, [i32, i32, i32, i32, c_f32p, c_i64p, C.POINTER(AttnParams), c_f32p, c_f32p, c_f32p,
                                      c_f32p, c_stream]),
    "rbr_review_attn_bwd_ws_floats": (C.c_size_t, [i32, i32, i32, i32]),
    "rbr_review_attn_bwd": (C.c_int, [i32, i32, i32, i32, c_f32p, c_i64p, C.POINTER(AttnParams), c_f32p, c_f32p, c_f32p,
                                      c_f32p, c_f32p, i32, C.POINTER(AttnGrads), c_f32p, c_f32p, c_stream]),
    "rbr_review_attn2_fwd": (C.c_int, [i32, i32, i32, i32, c_f32p, c_i64p, C.POINTER(AttnParams), C.POINTER(AttnParams), c_f32p,
                                       c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_review_attn2_bwd": (C.c_int, [i32, i32, i32, i32, c_f32p, c_i64p, C.POINTER(AttnParams), C.POINTER(AttnParams), c_f32p,
                                       c_f32p, c_f32p, c_f32p, c_f32p, i32, i32, C.POINTER(AttnGrads), C.POINTER(AttnGrads),
                                       C.c_int64, C.c_int64, c_f32p, c_f32p, c_stream]),
    "rbr_datt_local_gate_fwd": (C.c_int, [i32, i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_datt_global_gate_fwd": (C.c_int, [i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_datt_gate_bwd_ws_floats": (C.c_size_t, [i32, i32, i32, i32, i32]),
    "rbr_datt_local_gate_bwd": (C.c_int, [i32, i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p,
                                          c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_datt_global_gate_bwd": (C.c_int, [i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p,
                                           c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_datt_token_rows_ws_bytes": (C.c_size_t, [i32, i32, i32]),
    "rbr_datt_token_rows": (C.c_int, [i32, i32, i32, c_i64p, C.c_void_p, c_stream]),
    "rbr_datt_global_gate_bwd_rows_ws_floats": (C.c_size_t, [i32, i32, i32, i32]),
    "rbr_datt_global_gate_bwd_rows": (C.c_int, [i32, i32, i32, i32, c_i64p, c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p,
                                                c_f32p, c_f32p, c_f32p, C.c_void_p, i32, c_stream]),
    "rbr_linear_fwd": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_f32p, i32, c_f32p, c_f32p, c_stream]),
    "rbr_linear_bwd_ws_floats": (C.c_size_t, [i32, i32]),
    "rbr_linear_bwd": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p, c_f32p, c_f32p, c_f32p,
                                 c_f32p, c_stream]),
    "rbr_linear_fwd_ws_floats": (C.c_size_t, [i32, i32, i32]),
    "rbr_linear_bwd_ex_ws_floats": (C.c_size_t, [i32, i32, i32]),
    "rbr_linear_fwd_ex": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_f32p, i32, c_f32p, c_f32p, c_f32p, c_stream]),
    "rbr_linear_bwd_ex": (C.c_int, [i32, i32, i32, c_f32p, c_f32p, c_f32p, c_f32p, i32, c_f32p, c_f32p, c_f32p, c_f32p,
                                    c_f32p, c_stream]),
    "rbr_conv_shift_add_fwd": (C.c_int, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), c_f32p, _PP, c_f32p, c_stream]),
    "rbr_conv_shift_add_bwd": (C.c_int, [i32, i32, i32, C.POINTER(i32), C.POINTER(i32), c_f32p, c_f32p, _PP, c_stream]),
    "rbr_sanitize_ids": (C.c_int, [i32, C.POINTER(IdSet), C.c_void_p, c_stream]),
    "rbr_dedup_ws_bytes": (C.c_size_t, [i32, i32]),
    "rbr_dedup_rows": (C.c_int, [i32, i32, c_i64p, c_i64p, i32, i32, c_u8p, C.c_void_p, c_i64p, c_u8p, c_stream]),
    "rbr_embedding_fwd": (C.c_int, [C.c_int64, i32, c_i64p, c_f32p, c_f32p, c_stream]),
    "rbr_embedding_bwd": (C.c_int, [C.c_int64, i32, c_i64p, c_f32p, i32, c_f32p, c_stream]),
    "rbr_hier_pool_fwd": (C.c_int, [i32, i32, i32, i32, c_i64p, c_u8p, c_f32p, i32, c_f32p, c_i32p, c_stream]),
    "rbr_hier_pool_bwd": (C.c_int, [i32, i32, i32, i32, c_i64p, c_u8p, c_i32p, c_f32p, c_f32p, i32, i32, c_f32p,
                                    c_stream]),
}

_lib = None
_lock = threading.Lock()


def lib() -> C.CDLL:
    """Loads librbr_hip.so once; raises RuntimeError (never falls back) when it is unusable."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950). This package has no CPU or eager fallback.")
        try:
            handle = C.CDLL(LIB_PATH)
        except OSError as e:
            raise RuntimeError(f"cannot load {LIB_PATH}: {e}") from e
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as e:
                raise RuntimeError(f"{LIB_PATH} does not export {name}; rebuild it") from e
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().rbr_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (code {rc}): {msg}")


def dev_ptr(t: Optional[torch.Tensor], dtype: torch.dtype, name: str) -> Optional[int]:
    """Raw device pointer of a contiguous HIP tensor (None passes through as NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on a HIP device (got {t.device}); there is no CPU path")
    if t.dtype != dtype:
        raise RuntimeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    return t.data_ptr()


def ptr_array(tensors: Sequence[torch.Tensor], dtype: torch.dtype, name: str):
    arr = (C.c_void_p * max(RBR_MAX_WIDTHS, len(tensors)))()
    for i, t in enumerate(tensors):
        arr[i] = dev_ptr(t, dtype, f"{name}[{i}]")
    return arr


# ---- capture guard.  While train_step.GraphedTrainStep / GraphedForward record a hipGraph, every op of this package must be
# enqueued on the capturing stream or on a stream this package forked from it with events (functional._side_stream: the weight-
# gradient chain).  An op issued on any OTHER stream during a capture is refused with a Python error -- round 3's "item tower on a
# stream of its own" experiment ended in a segmentation fault inside the capture of the backward instead (the
# word table's AccumulateGrad node lived on another stream than the gradient's producer, and autograd's cross-stream hand-over to
# it is what hipGraph capture died in); a two-stream forward must fork and join INSIDE one autograd function, as _textcnn_backward
# and _DattTowers.backward do, so that autograd sees one node on one stream.
_CAPTURE_GUARD = None            # None, or the set of stream handles an op may use while a capture is being recorded
FORKED_STREAMS: set = set()      # streams functional forks from the current one with events (allowed under the guard)


class capture_guard:
    """with capture_guard(stream): ... -- ops of this package raise if they find another current stream than `stream` (or one of
    the package's own forked streams)."""

    def __init__(self, stream):
        self.allowed = {stream.cuda_stream}

    def __enter__(self):
        global _CAPTURE_GUARD
        self.prev, _CAPTURE_GUARD = _CAPTURE_GUARD, self.allowed
        return self

    def __exit__(self, *a):
        global _CAPTURE_GUARD
        _CAPTURE_GUARD = self.prev
        return False


def current_stream() -> int:
    s = torch.cuda.current_stream().cuda_stream
    if _CAPTURE_GUARD is not None and s not in _CAPTURE_GUARD and s not in FORKED_STREAMS:
        raise RuntimeError("an op of this package was issued on a stream that is neither the capturing stream nor one the package "
                           "forked from it: a hipGraph capture cannot follow it (fork and join inside ONE autograd function "
                           "instead; see _lib.capture_guard)")
    return s


CONV_PAD_RUNS = 1      # RBR_CONV_PAD_RUNS


def make_desc(n_docs, L, D, V, kernel_sizes, channels, pad_mode, act, padding_idx, flags: int = 0) -> TextCNNDesc:
    if len(kernel_sizes) > RBR_MAX_WIDTHS:
        raise RuntimeError(f"at most {RBR_MAX_WIDTHS} kernel widths are supported")
    d = TextCNNDesc()
    d.n_docs, d.L, d.D, d.V = int(n_docs), int(L), int(D), int(V)
    d.n_widths = len(kernel_sizes)
    for i, (k, c) in enumerate(zip(kernel_sizes, channels)):
        d.kz[i] = int(k)
        d.ch[i] = int(c)
    d.pad_mode, d.act = int(pad_mode), int(act)
    d.padding_idx = -1 if padding_idx is None else int(padding_idx)
    d.flags = int(flags)
    # the arithmetic class in force NOW travels with the descriptor (forward -> backward -> any later call on the same workspace)
    lib().rbr_textcnn_desc_stamp(C.byref(d))
    return d


# --------------------------------------------------------------------------- kernel timing hooks
class KernelTimer:
    """Optional HIP-event timing of individual C-ABI calls on the current stream (bench.py)."""

    def __init__(self):
        self.enabled = False
        self.events = {}

    def start(self):
        self.enabled = True
        self.events = {}

    def stop(self):
        self.enabled = False

    def record(self, name):
        if not self.enabled:
            return None
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        self.events.setdefault(name, []).append((a, b))
        a.record()
        return b

    def summary(self):
        """name -> (calls, mean ms); call after torch.cuda.synchronize()."""
        return {k: (len(v), sum(a.elapsed_time(b) for a, b in v) / len(v)) for k, v in self.events.items()}


TIMER = KernelTimer()
