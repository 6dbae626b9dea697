"""ctypes binding of libd3d_hip.so (include/d3d.h).  There is no fallback: a missing library is an ImportError-like
RuntimeError at first use, and every compute entry point needs a HIP device."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libd3d_hip.so")

PREC_FP32, PREC_F16X3, PREC_BF16 = 0, 1, 2
KC_COUNT = 13
RANGE_ACT, RANGE_WEIGHT, RANGE_STATS, RANGE_INDEX, RANGE_RECOMPUTE = 1, 2, 4, 8, 16
RANGE_PRECISION = RANGE_ACT | RANGE_WEIGHT | RANGE_STATS        # the bits that mean "not fp32-accurate in F16X3"
PRECISIONS = {"fp32": PREC_FP32, "f16x3": PREC_F16X3, "bf16": PREC_BF16}

# every symbol include/d3d.h declares (tests/test_abi.py checks the library exports all of them)
ABI_SYMBOLS = [
    "d3d_last_error", "d3d_version", "d3d_engine_create", "d3d_engine_destroy", "d3d_engine_num_weights",
    "d3d_engine_weight_info", "d3d_engine_set_weight", "d3d_engine_set_time_freqs", "d3d_engine_commit_weights",
    "d3d_engine_set_schedule", "d3d_engine_set_sqrt_alphas_cumprod", "d3d_ddim_times", "d3d_workspace_bytes",
    "d3d_denoise", "d3d_ddim_sample", "d3d_q_sample", "d3d_tta_mpjpe", "d3d_allgather_pred", "d3d_op_linear", "d3d_op_layernorm",
    "d3d_op_attention", "d3d_op_time_embedding", "d3d_engine_set_profiling", "d3d_engine_profile_reset",
    "d3d_engine_profile_read", "d3d_kernel_class_name", "d3d_op_linear_bench", "d3d_op_linear_postnorm", "d3d_engine_set_graph_mode", "d3d_num_windows", "d3d_window_gather",
    "d3d_engine_set_trace", "d3d_engine_trace_read", "d3d_engine_range_flags", "d3d_op_head", "d3d_engine_set_option",
    "d3d_weighted_loss", "d3d_repeat_batch", "d3d_hypothesis_mean", "d3d_engine_get_info", "d3d_probe_machine",
    "d3d_engine_range_post", "d3d_engine_range_take", "d3d_window_gather_s2f", "d3d_pose_metrics",
]


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_frame", "num_joints", "in_chans", "embed_dim", "depth", "num_heads",
                                          "mlp_hidden", "with_time_emb", "seq2frame", "precision")]


class D3DError(RuntimeError):
    pass


_lib: Optional[C.CDLL] = None


def _bind(lib: C.CDLL) -> None:
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t
    sig = {
        "d3d_last_error": (C.c_char_p, []),
        "d3d_version": (C.c_int, []),
        "d3d_engine_create": (C.c_int, [C.POINTER(Config), C.POINTER(vp)]),
        "d3d_engine_destroy": (None, [vp]),
        "d3d_engine_num_weights": (C.c_int, [vp]),
        "d3d_engine_weight_info": (C.c_int, [vp, C.c_int, C.POINTER(C.c_char_p), C.POINTER(i64)]),
        "d3d_engine_set_weight": (C.c_int, [vp, C.c_char_p, vp, i64]),
        "d3d_engine_set_time_freqs": (C.c_int, [vp, vp, i32]),
        "d3d_engine_commit_weights": (C.c_int, [vp]),
        "d3d_engine_set_schedule": (C.c_int, [vp, i32, vp, vp, i32, f32, i32, vp]),
        "d3d_engine_set_sqrt_alphas_cumprod": (C.c_int, [vp, vp, i32]),
        "d3d_ddim_times": (C.c_int, [i32, i32, C.POINTER(i32)]),
        "d3d_workspace_bytes": (sz, [vp, i32]),
        "d3d_denoise": (C.c_int, [vp, vp, vp, i32, vp, i32, vp, i32, vp, sz, vp]),
        "d3d_ddim_sample": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, i32, vp, sz, vp]),
        "d3d_q_sample": (C.c_int, [vp, vp, vp, vp, vp, i32, i64, vp]),
        "d3d_allgather_pred": (C.c_int, [vp, vp, vp, i64, vp]),
        "d3d_weighted_loss": (C.c_int, [vp, vp, vp, vp, vp, i32, i64, i32, i32, vp]),
        "d3d_repeat_batch": (C.c_int, [vp, vp, i32, i64, i32, vp]),
        "d3d_hypothesis_mean": (C.c_int, [vp, vp, i32, i64, i32, vp]),
        "d3d_engine_get_info": (C.c_int, [vp, C.c_char_p, C.POINTER(i64)]),
        "d3d_probe_machine": (C.c_int, [i32, f32, C.POINTER(f32), vp]),
        "d3d_tta_mpjpe": (C.c_int, [vp, vp, vp, vp, f32, C.POINTER(i32), C.POINTER(i32), i32, vp, vp, i32, i32, i32, vp]),
        "d3d_pose_metrics": (C.c_int, [vp, vp, vp, vp, i32, i32, vp]),
        "d3d_engine_set_profiling": (C.c_int, [vp, i32]),
        "d3d_engine_set_graph_mode": (C.c_int, [vp, i32]),
        "d3d_engine_set_option": (C.c_int, [vp, C.c_char_p, i64]),
        "d3d_num_windows": (C.c_int, [i32, i32]),
        "d3d_window_gather": (C.c_int, [vp, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), i32, vp, vp, vp]),
        "d3d_window_gather_s2f": (C.c_int, [vp, i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), i32, i32, i32, vp, vp]),
        "d3d_engine_range_flags": (C.c_int, [vp, C.POINTER(C.c_uint32), i32, vp]),
        "d3d_engine_range_post": (C.c_int, [vp, vp, C.POINTER(i64)]),
        "d3d_engine_range_take": (C.c_int, [vp, i64, i32, C.POINTER(C.c_uint32), C.POINTER(i32)]),
        "d3d_engine_set_trace": (C.c_int, [vp, i32, i32]),
        "d3d_engine_trace_read": (C.c_int, [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint32), i32, C.POINTER(i32), vp]),
        "d3d_engine_profile_reset": (C.c_int, [vp]),
        "d3d_engine_profile_read": (C.c_int, [vp, i32, C.POINTER(C.c_double), C.POINTER(i64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
        "d3d_kernel_class_name": (C.c_char_p, [i32]),
        "d3d_op_time_embedding": (C.c_int, [vp, vp, i32, vp, vp, vp]),
        "d3d_op_linear": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp]),
        "d3d_op_linear_bench": (C.c_int, [vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, C.POINTER(C.c_float), vp]),
        "d3d_op_linear_postnorm": (C.c_int, [vp, vp, vp, vp, vp, vp, f32, vp, i32, i32, vp, i64, i32, vp, vp, i32, i32, i32, i32,
                                             C.POINTER(C.c_float), vp]),
        "d3d_op_head": (C.c_int, [vp, vp, vp, i32, vp]),
        "d3d_op_layernorm": (C.c_int, [vp, vp, vp, vp, i32, i32, f32, vp]),
        "d3d_op_attention": (C.c_int, [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args


def lib() -> C.CDLL:
    """Load the HIP library (once).  Raises if it has not been built -- there is no Python/CPU substitute."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise D3DError(f"{LIB_PATH} not found: build it with `python -m diff3dhpe_amd.build` "
                           "(the engine has no CPU or eager fallback)")
        l = C.CDLL(LIB_PATH)
        _bind(l)
        _lib = l
    return _lib


def check(rc: int) -> None:
    if rc != 0:
        msg = lib().d3d_last_error()
        raise D3DError(f"libd3d_hip error {rc}: {msg.decode() if msg else '?'}")


def ddim_times(num_timesteps: int, sampling_timesteps: int):
    """Host-only: the reversed integer DDIM schedule (reference DIFF:270-272), S+1 entries."""
    out = (C.c_int32 * (sampling_timesteps + 1))()
    check(lib().d3d_ddim_times(num_timesteps, sampling_timesteps, out))
    return list(out)
