"""ctypes binding of ``libuformer_hip.so`` (C ABI declared in ``include/uformer_hip.h``).

There is NO fallback: if the library is missing or a call fails this module raises.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libuformer_hip.so")
# A/B builds and out-of-tree installs: UFORMER_HIP_LIB=/path/to/libuformer_hip.so overrides the in-tree library
LIB_PATH = os.environ.get("UFORMER_HIP_LIB", LIB_PATH)

UF_F32, UF_BF16, UF_F16 = 0, 1, 2

c_int, c_void_p, c_size_t, c_float_p = C.c_int, C.c_void_p, C.c_size_t, C.c_void_p


class BlockParams(C.Structure):
    """``uf_block_params`` (include/uformer_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "modulator", "rpb_dense", "rpb_fm", "rpb_tab", "wqkv_fm", "bqkv", "wproj", "wproj_fm", "bproj",
        "norm2_w", "norm2_b", "w1_fm", "b1", "wdw9", "bdw", "w2_fm", "b2")] + [
        ("shift", C.c_int32), ("heads", C.c_int32)]


class BlockTrainParams(C.Structure):
    """``uf_block_train_params`` (include/uformer_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "norm2_w", "norm2_b", "modulator", "rpb_dense", "wqkv", "wqkv_t", "bqkv", "wproj", "wproj_t", "bproj",
        "w1", "w1_t", "b1", "wdw9", "wdw9_flip", "bdw", "w2_t")] + [("shift", C.c_int32), ("heads", C.c_int32), ("w2", C.c_void_p)]


class BlockRawParams(C.Structure):
    """``uf_block_raw_params`` (include/uformer_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "norm2_w", "norm2_b", "modulator", "rpb_table", "rpb_index", "to_q_w", "to_q_b", "to_kv_w", "to_kv_b", "proj_w", "proj_b",
        "lin1_w", "lin1_b", "dw_w", "dw_b", "lin2_w", "lin2_b")] + [("index_is_standard", C.c_int32)]


class BlockGrads(C.Structure):
    """``uf_block_grads`` (include/uformer_hip.h)."""
    _fields_ = [(n, C.c_void_p) for n in (
        "norm1_w", "norm1_b", "norm2_w", "norm2_b", "modulator", "rpb_table", "wqkv", "bqkv", "wproj", "bproj", "w1", "b1", "wdw", "bdw", "w2", "b2")]


class ModelDesc(C.Structure):
    """``uf_model_desc`` (include/uformer_hip.h)."""
    _fields_ = [
        ("embed_dim", C.c_int32), ("dd_in", C.c_int32), ("in_chans", C.c_int32),
        ("depths", C.c_int32 * 9),
        ("blocks", C.POINTER(BlockParams)),
        ("in_w27", C.c_void_p), ("in_b", C.c_void_p), ("out_w", C.c_void_p), ("out_b", C.c_void_p),
        ("down_w", C.c_void_p * 4), ("down_b", C.c_void_p * 4),
        ("up_w", C.c_void_p * 4), ("up_b", C.c_void_p * 4),
        ("down_w_fm", C.c_void_p * 4),
    ]


P = c_void_p
I = c_int
# name -> (restype, argtypes); must list every function declared in include/uformer_hip.h
SIGNATURES = {
    "uf_version": (I, []),
    "uf_last_error": (I, [C.c_char_p, c_size_t]),
    "uf_timing_enable": (I, [I]),
    "uf_timing_report": (I, [C.c_char_p, c_size_t]),
    "uf_debug_set_tbuf": (I, [P]),
    "uf_weight_fm_elems": (c_size_t, [I, I]),
    "uf_pack_weight_fm": (I, [P, P, I, I, I, P]),
    "uf_window_partition": (I, [P, P, I, I, I, I, I, I, P]),
    "uf_window_reverse": (I, [P, P, I, I, I, I, I, I, P]),
    "uf_shift_mask": (I, [P, I, I, I, P]),
    "uf_layernorm_fwd": (I, [P, I, P, P, P, P, I, I, I, I, I, I, I, P]),
    "uf_linear_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "uf_qkv_fwd": (I, [P, P, P, P, P, P, I, I, I, I, P]),
    "uf_ln_qkv_fwd": (I, [P, I, P, P, P, P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "uf_ln_linear_gelu_fwd": (I, [P, I, P, P, P, P, P, I, I, I, I, P]),
    "uf_window_attention_fwd": (I, [P, P, P, P, P, I, P, I, I, I, I, I, I, I, P]),
    "uf_dwconv3x3_gelu_fwd": (I, [P, P, P, P, I, I, I, I, I, P]),
    "uf_dwconv3x3_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "uf_gelu_bwd": (I, [P, P, P, C.c_longlong, I, P]),
    "uf_gelu_fwd": (I, [P, P, C.c_longlong, I, P]),
    "uf_linear_pre_gelu_fwd": (I, [P, P, P, P, P, I, I, I, I, P]),
    "uf_linear_mul_dgelu": (I, [P, P, P, P, P, I, I, I, I, P]),
    "uf_dwconv3x3_pre_gelu_fwd": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "uf_dwconv3x3_gelu_in_pre_gelu_fwd": (I, [P, P, P, P, P, I, I, I, I, I, P]),
    "uf_dwconv3x3_mul_dgelu": (I, [P, P, P, P, I, I, I, I, I, P]),
    "uf_linear_residual_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, I, I, P]),
    "uf_layernorm_bwd_workspace_bytes": (c_size_t, [I, I]),
    "uf_layernorm_bwd": (I, [P, I, P, P, I, P, I, P, P, I, I, P, c_size_t, P]),
    "uf_layernorm_bwd_fused": (I, [P, I, P, P, I, I, P, P, I, P, P, I, I, I, I, I, I, I, P, c_size_t, P]),
    "uf_layernorm_bwd_cast": (I, [P, I, P, P, I, I, P, P, I, P, P, I, I, I, I, I, I, I, P, P, I, I, P, c_size_t, P]),
    "uf_linear_wgrad_workspace_bytes": (c_size_t, [I, I, I]),
    "uf_linear_wgrad": (I, [P, I, P, I, P, P, I, I, I, I, P, c_size_t, P]),
    "uf_window_attention_bwd_workspace_bytes": (c_size_t, [I, I]),
    "uf_window_attention_bwd": (I, [P, P, P, P, P, I, P, I, P, P, P, P, I, I, I, I, I, I, I, P, c_size_t, P]),
    "uf_window_attention_bwd_qkv": (I, [P, P, P, P, P, I, P, I, P, P, I, I, I, I, I, I, I, P, c_size_t, P]),
    "uf_dwconv3x3_wgrad_workspace_bytes": (c_size_t, [I, I]),
    "uf_dwconv3x3_wgrad": (I, [P, P, P, P, I, I, I, I, I, P, c_size_t, P]),
    "uf_dwconv3x3_bwd_workspace_bytes": (c_size_t, [I, I]),
    "uf_dwconv3x3_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P, c_size_t, P]),
    "uf_dwconv_linear2_fwd": (I, [P, P, P, P, P, P, I, I, I, I, I, I, P]),
    "uf_block_workspace_bytes": (c_size_t, [I, I, I]),
    "uf_lewin_attn_fwd": (I, [C.POINTER(BlockParams), P, I, I, I, I, I, P, I, I, P, c_size_t, P]),
    "uf_leff_fwd": (I, [C.POINTER(BlockParams), P, I, I, I, I, I, I, P, c_size_t, P]),
    "uf_lewin_attn_train_fwd": (I, [C.POINTER(BlockParams), P, I, P, I, I, I, I, I, P, I, P, P, P, P, P, P, P, P]),
    "uf_lewin_block_fwd": (I, [C.POINTER(BlockParams), P, I, I, I, I, I, P, I, I, P, c_size_t, P]),
    "uf_lewin_block_train_fwd": (I, [C.POINTER(BlockParams), P, I, I, I, I, I, P, P, I, P, c_size_t, P]),
    "uf_downsample_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, P]),
    "uf_downsample_fm_fwd": (I, [P, I, P, P, P, P, I, I, I, I, I, I, P]),
    "uf_upsample_fwd": (I, [P, I, P, P, P, I, I, I, I, I, I, I, P]),
    "uf_input_proj_fwd": (I, [P, P, P, P, I, I, I, I, I, I, P]),
    "uf_output_proj_fwd": (I, [P, I, P, P, P, P, I, I, I, I, I, P]),
    "uf_rows_sum_workspace_bytes": (c_size_t, [I, I]),
    "uf_rows_sum": (I, [P, I, P, I, I, I, P, c_size_t, P]),
    "uf_rpb_table_grad": (I, [P, P, I, P]),
    "uf_im2col": (I, [P, I, P, I, I, I, I, I, I, I, I, I, I, P]),
    "uf_col2im": (I, [P, I, P, I, I, I, I, I, I, I, I, I, I, I, P]),
    "uf_pack_block_train_bytes": (C.c_size_t, [I, I, I]),
    "uf_pack_block_train": (I, [P, I, I, I, I, P, C.c_size_t, P, P, P]),
    "uf_lewin_block_bwd_workspace_bytes": (C.c_size_t, [I, I, I, I, I, I]),
    "uf_lewin_block_bwd": (I, [P, P, P, P, P, P, P, I, I, I, I, I, P, C.c_size_t, P]),
    "uf_leff_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P, C.c_size_t, P]),
    "uf_lewin_attn_bwd": (I, [P, P, P, P, P, P, I, I, I, I, I, P, C.c_size_t, P]),
    "uf_downsample_bwd_workspace_bytes": (C.c_size_t, [I, I, I, I, I, I]),
    "uf_downsample_bwd": (I, [P, I, P, P, P, I, I, P, P, I, I, I, I, I, I, P, C.c_size_t, P]),
    "uf_upsample_cat_bwd_workspace_bytes": (C.c_size_t, [I, I, I, I, I, I]),
    "uf_upsample_cat_bwd": (I, [P, I, P, I, P, P, P, P, I, I, I, I, I, I, P, C.c_size_t, P]),
    "uf_conv3x3_bwd_workspace_bytes": (C.c_size_t, [I, I, I, I, I]),
    "uf_conv3x3_bwd": (I, [P, I, P, P, C.c_float, P, P, P, P, I, I, I, I, I, P, C.c_size_t, P]),
    "uf_residual_combine": (I, [P, P, I, P, P, I, I, I, I, I, I, I, P]),
    "uf_grad_fork": (I, [P, P, P, P, P, I, I, I, I, I, I, I, P]),
    "uf_qkv_grad_merge": (I, [P, P, P, P, I, I, I, I, P]),
    "uf_charbonnier_workspace_bytes": (c_size_t, [C.c_longlong]),
    "uf_charbonnier_fwd_bwd": (I, [P, P, P, P, C.c_longlong, C.c_float, C.c_float, P, c_size_t, P]),
    "uf_adamw_step": (I, [P, P, P, P, P, I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, I, C.c_double, P]),
    "uf_grad_scaler_check": (I, [P, P, I, P, P]),
    "uf_adamw_step_scaled": (I, [P, P, P, P, P, I, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, P, P]),
    "uf_grad_scaler_update": (I, [P, C.c_double, C.c_double, I, P]),
    "uf_image_metric_workspace_bytes": (c_size_t, [I, I, I, I]),
    "uf_batch_mse": (I, [P, P, P, I, I, I, I, I, P, c_size_t, P]),
    "uf_batch_ssim": (I, [P, P, P, I, I, I, I, P, c_size_t, P]),
    "uf_expand2square": (I, [P, P, P, I, I, I, I, I, P]),
    "uf_crop_clamp": (I, [P, P, I, I, I, I, I, I, P]),
    "uf_crop_augment": (I, [P, I, I, P, P, I, I, I, I, I, P]),
    "uf_mixup": (I, [P, P, P, P, I, C.c_longlong, P]),
    "uf_uformer_workspace_bytes": (c_size_t, [C.POINTER(ModelDesc), I, I, I, I]),
    "uf_uformer_fwd": (I, [C.POINTER(ModelDesc), P, P, I, I, I, I, P, c_size_t, P]),
}

_lock = threading.Lock()
_lib = None


class UformerHipError(RuntimeError):
    pass


def load():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise UformerHipError(
                f"{LIB_PATH} not found: build it with `python __graft_entry__.py` "
                "(hipcc --offload-arch=gfx950).  There is no CPU / PyTorch fallback for the hot path.")
        lib = C.CDLL(LIB_PATH)
        older = os.environ.get("UF_ALLOW_OLDER_LIB") == "1"   # A/B against a library of an earlier round (scripts/official_run.sh): skip what it lacks
        for name, (res, args) in SIGNATURES.items():
            if older and not hasattr(lib, name):
                continue
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
            fn.restype = res
            fn.argtypes = args
        if lib.uf_version() != 1:
            raise UformerHipError(f"ABI version mismatch: library reports {lib.uf_version()}, binding expects 1")
        _lib = lib
    return _lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().uf_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc: int, what: str) -> None:
    if rc != 0:
        raise UformerHipError(f"{what} failed (code {rc}): {last_error()}")
