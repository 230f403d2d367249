"""ctypes binding of libhudiff_hip.so (the C ABI in include/hudiff_hip.h).

The library is the product path; there is deliberately no CPU fallback here: if the shared object is
missing or no gfx950 device is visible, construction raises.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HUDIFF_LIB") or os.path.join(HERE, "libhudiff_hip.so")    # HUDIFF_LIB: A/B builds

HD_ABI_VERSION = 1
HD_KIND_ANTIBODY, HD_KIND_NANOBODY = 0, 1
HD_ACT_RELU, HD_ACT_GELU = 1, 2
HD_DROPOUT_FAITHFUL, HD_DROPOUT_OFF, HD_DROPOUT_INJECT, HD_NO_GRAPH, HD_NO_PRUNE, HD_ONE_LANE, HD_LOOP_GRAPH = 0, 1, 2, 4, 8, 16, 32
HD_PRECISION_DEFAULT, HD_PRECISION_F32_GEMM, HD_PRECISION_F32_ALL, HD_PRECISION_SPLIT = 0, 1, 2, 3
PRECISIONS = {"default": HD_PRECISION_DEFAULT, "split": HD_PRECISION_SPLIT, "f32_gemm": HD_PRECISION_F32_GEMM, "f32_all": HD_PRECISION_F32_ALL}
PRECISION_NAMES = {HD_PRECISION_SPLIT: "split", HD_PRECISION_F32_GEMM: "f32_gemm", HD_PRECISION_F32_ALL: "f32_all", HD_PRECISION_DEFAULT: "default"}
HD_OK, HD_ERR_INVALID, HD_ERR_UNSUPPORTED, HD_ERR_STATE, HD_ERR_HIP, HD_ERR_NO_DEVICE, HD_ERR_NUMERIC = range(7)

EXPORTS = [
    "hd_device_count", "hd_create", "hd_load_tensor", "hd_finalize", "hd_destroy", "hd_last_error",
    "hd_forward", "hd_sample", "hd_sample_begin", "hd_sample_run", "hd_sample_restart", "hd_sample_end", "hd_sync",
    "hd_last_run_ms", "hd_flops_per_row_forward", "hd_flops_per_row_sample_step", "hd_device_info", "hd_debug_stop_after", "hd_debug_read", "hd_precision_info",
    "hd_set_precision", "hd_precision_report", "hd_precision_reset", "hd_sample_tokens", "hd_debug_fail_next_lnsync",
    "hd_set_option", "hd_get_option", "hd_debug_scatter_lnsync",
]

# tuning options (include/hudiff_hip.h, HdOption): name -> id; the names are the enum's, lower case without the HD_OPT_ prefix
OPTIONS = {n: i for i, n in enumerate((
    "lanes", "lane_min_rows", "split_min_rows", "big_min_rows", "lnsync_level", "tail_form", "tail_max_rows", "small_grid", "tiny_grid",
    "loader_waves", "tiny_stages", "small_stages", "small_stages3_max_grid", "attn_qsplit_max_grid", "attn_waves", "loop_graph",
    "prune_value_via_rows", "split_tile", "gemm_small_tiles", "store_nt", "split_layer_mask", "split_attn", "fused_attn", "fused_attn_min_grid",
    "bn_chain", "bn_chain_min_tiles"))}


class HdConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "kind", "n_tokens", "max_len", "h_len", "d_model", "sum_d_model",
        "n_encoder_layers", "dual_layers", "kernel_size", "r", "att_model", "nhead", "dim_feedforward",
        "cs_layers", "n_region", "r_embedding", "n_side", "s_embedding", "enc_act", "conv_act")] + \
        [("dropout", C.c_float)]


class HdPrecisionInfo(C.Structure):
    _fields_ = [("precision", C.c_int32), ("split_built", C.c_int32), ("split_in_use", C.c_int32), ("lnsync_in_use", C.c_int32),
                ("range_fallbacks", C.c_int64), ("lnsync_fallbacks", C.c_int64), ("last_call_repeated", C.c_int32), ("lnsync_cross_xcd", C.c_int32)]


class HudiffError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"libhudiff_hip status {status}: {message}")
        self.status = status


_lib = None


def load():
    """dlopen the in-tree shared object and declare prototypes.  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError(
            f"{LIB_PATH} is missing: build it with `python -m hudiff_amd.build` (hipcc, gfx950). "
            "hudiff_amd has no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    i32p, f32p, u8p, vp = P(C.c_int32), P(C.c_float), P(C.c_uint8), C.c_void_p
    lib.hd_device_count.restype = C.c_int
    lib.hd_last_error.restype = C.c_char_p
    lib.hd_create.argtypes = [P(HdConfig), C.c_int, P(vp)]
    lib.hd_load_tensor.argtypes = [vp, C.c_char_p, f32p, P(C.c_int64), C.c_int32]
    lib.hd_finalize.argtypes = [vp]
    lib.hd_destroy.argtypes = [vp]
    lib.hd_destroy.restype = None
    lib.hd_forward.argtypes = [vp, i32p, i32p, i32p, C.c_int32, C.c_uint32, C.c_uint64, C.c_uint64,
                               C.c_uint32, u8p, u8p, f32p]
    sample_args = [i32p, i32p, i32p, i32p, C.c_int32, C.c_int32, C.c_uint32, C.c_uint64, C.c_uint64,
                   f32p, u8p, u8p]
    lib.hd_sample.argtypes = [vp, i32p] + sample_args
    lib.hd_sample_begin.argtypes = [vp, i32p] + sample_args
    lib.hd_sample_run.argtypes = [vp, C.c_int32, C.c_int32]
    lib.hd_sample_restart.argtypes = [vp, C.c_uint64]
    lib.hd_sample_end.argtypes = [vp, i32p]
    lib.hd_sample_tokens.argtypes = [vp, i32p]
    lib.hd_sync.argtypes = [vp]
    lib.hd_last_run_ms.argtypes = [vp, f32p, i32p]
    lib.hd_flops_per_row_forward.argtypes = [P(HdConfig)]
    lib.hd_flops_per_row_forward.restype = C.c_double
    lib.hd_flops_per_row_sample_step.argtypes = [P(HdConfig)]
    lib.hd_flops_per_row_sample_step.restype = C.c_double
    lib.hd_device_info.argtypes = [C.c_int, C.c_char_p, C.c_size_t, i32p, P(C.c_int64)]
    lib.hd_precision_info.argtypes = [vp, i32p, i32p, P(C.c_int64)]
    lib.hd_set_precision.argtypes = [vp, C.c_int32]
    lib.hd_precision_report.argtypes = [vp, P(HdPrecisionInfo), C.c_size_t]
    lib.hd_precision_reset.argtypes = [vp]
    lib.hd_debug_fail_next_lnsync.argtypes = [vp]
    lib.hd_set_option.argtypes = [vp, C.c_int32, C.c_int64]
    lib.hd_get_option.argtypes = [vp, C.c_int32, P(C.c_int64)]
    lib.hd_debug_scatter_lnsync.argtypes = [vp, C.c_int32]
    lib.hd_debug_stop_after.argtypes = [vp, C.c_int32]
    lib.hd_debug_read.argtypes = [vp, C.c_char_p, C.c_int32, f32p, C.c_int64]
    _lib = lib
    return lib


def check(status):
    if status != HD_OK:
        raise HudiffError(status, load().hd_last_error().decode("utf-8", "replace"))


def as_i32(a, shape=None):
    a = np.ascontiguousarray(np.asarray(a), dtype=np.int32)
    if shape is not None and tuple(a.shape) != tuple(shape):
        raise ValueError(f"expected shape {tuple(shape)}, got {tuple(a.shape)}")
    return a


def ptr(a, ctype):
    return None if a is None else a.ctypes.data_as(C.POINTER(ctype))
