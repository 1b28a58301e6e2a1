"""ctypes binding of include/mppi_hip.h (the drop-in boundary).

There is no CPU fallback: if libmppi_hip.so is missing or no MI355X is visible the calls
raise.  PyTorch-ROCm tensors are passed zero-copy by ``tensor.data_ptr()``.
"""
from __future__ import annotations

import ctypes as C
import os

PKG = os.path.dirname(os.path.abspath(__file__))
ABI_VERSION = 4  # MPPI_ABI_VERSION of include/mppi_hip.h this mirror was written against
MIN_AB_ABI_VERSION = 3  # oldest library an MPPI_LIB override may point at: same mppi_config layout as now (an ABI-3 library
                        # fills the first fields of mppi_stats only: the struct grew at its end in version 4)
LIB_PATH = os.environ.get("MPPI_LIB") or os.path.join(PKG, "lib", "libmppi_hip.so")  # MPPI_LIB: A/B a diagnostic build

# enums of mppi_hip.h
MODEL_DIFFDRIVE, MODEL_RACECAR, MODEL_DIFFDRIVE_MLP = 0, 1, 2
PREC_F32, PREC_F64 = 0, 1
WAYPOINT_SEQUENTIAL, WAYPOINT_FROZEN, WAYPOINT_PER_ROLLOUT = 0, 1, 2
BETA_INV_EXPLORATION, BETA_INV_LAMBDA, BETA_LAMBDA = 0, 1, 2
FILTER_DIFFDRIVE, FILTER_RACECAR, FILTER_NONE, FILTER_TORCH = 0, 1, 2, 3
OBSTACLE_NONE, OBSTACLE_CIRCLE, OBSTACLE_OUTLINE = 0, 1, 2
OK, ERR_BAD_ARG, ERR_SHAPE, ERR_NO_DEVICE, ERR_HIP, ERR_PATH_END, ERR_UNSUPPORTED, ERR_STATE = 0, -1, -2, -3, -4, -5, -6, -7
ERR_COMM = -8
LAYOUT_FUSED, LAYOUT_DUAL, LAYOUT_PAIR, LAYOUT_TRI, LAYOUT_KIND, LAYOUT_TWICE = 0, 1, 2, 3, 3, 4  # mppi_get_rollout_layout (KIND: mask)


class MppiConfig(C.Structure):
    _fields_ = [
        ("struct_size", C.c_int32), ("device", C.c_int32), ("model", C.c_int32), ("precision", C.c_int32),
        ("K", C.c_int32), ("T", C.c_int32), ("K_global", C.c_int32), ("k_offset", C.c_int32),
        ("delta_t", C.c_double), ("u_max", C.c_double * 2), ("wheel_base", C.c_double),
        ("param_exploration", C.c_double), ("param_lambda", C.c_double), ("param_alpha", C.c_double),
        ("sigma", C.c_double * 4), ("stage_cost_weight", C.c_double * 4), ("terminal_cost_weight", C.c_double * 4),
        ("beta_mode", C.c_int32), ("accumulate_stage_cost", C.c_int32), ("waypoint_mode", C.c_int32),
        ("search_window", C.c_int32), ("wrap_yaw_stage", C.c_int32), ("wrap_yaw_terminal", C.c_int32),
        ("clamp_rollout", C.c_int32), ("clamp_u_after_update", C.c_int32), ("filter_mode", C.c_int32),
        ("filter_window", C.c_int32), ("obstacle_model", C.c_int32), ("raise_at_path_end", C.c_int32),
        ("safety_margin", C.c_double), ("vehicle_w", C.c_double), ("vehicle_l", C.c_double),
        ("collision_penalty", C.c_double), ("seed", C.c_uint64), ("n_agents", C.c_int32), ("noise_stream", C.c_int32),
    ]


class MppiStats(C.Structure):
    _fields_ = [("rho", C.c_double), ("eta", C.c_double), ("ess", C.c_double), ("idx_start", C.c_int32),
                ("idx_after", C.c_int32), ("path_end", C.c_int32), ("rounds", C.c_int32), ("iteration", C.c_int64),
                ("n_collided", C.c_int32), ("reserved", C.c_int32), ("iter_us", C.c_double), ("kernel_us", C.c_double)]


class MppiError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libmppi_hip: {msg} (status {code})")
        self.code = code


_H = C.c_void_p
_D = C.POINTER(C.c_double)

# every symbol include/mppi_hip.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "mppi_abi_version": (C.c_int, []),
    "mppi_last_error": (C.c_char_p, [_H]),
    "mppi_device_count": (C.c_int, []),
    "mppi_create": (C.c_int, [C.POINTER(MppiConfig), C.POINTER(_H)]),
    "mppi_destroy": (C.c_int, [_H]),
    "mppi_set_ref_path": (C.c_int, [_H, _D, C.c_int32, C.c_int32]),
    "mppi_set_obstacles": (C.c_int, [_H, _D, C.c_int32]),
    "mppi_set_mlp": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                              C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                              C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mppi_set_mlp_scaled": (C.c_int, [_H, C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                     C.POINTER(C.POINTER(C.c_float)), C.POINTER(C.POINTER(C.c_float)),
                                     C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_double),
                                     C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "mppi_set_u_prev": (C.c_int, [_H, _D]),
    "mppi_get_u_prev": (C.c_int, [_H, _D]),
    "mppi_set_waypoint_idx": (C.c_int, [_H, C.c_int32]),
    "mppi_get_waypoint_idx": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "mppi_step": (C.c_int, [_H, _D, C.c_void_p, _D, _D, C.POINTER(MppiStats), C.c_void_p]),
    "mppi_partial_len": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "mppi_step_begin": (C.c_int, [_H, _D, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_step_end": (C.c_int, [_H, C.c_void_p, C.c_int32, _D, _D, C.POINTER(MppiStats), C.c_void_p]),
    "mppi_step_end_async": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_void_p]),
    "mppi_sync_result": (C.c_int, [_H, _D, _D, C.POINTER(MppiStats), C.c_void_p]),
    "mppi_comm_handle_bytes": (C.c_int, []),
    "mppi_comm_export": (C.c_int, [_H, C.c_int32, C.c_void_p]),
    "mppi_comm_buffer": (C.c_int, [_H, C.POINTER(C.c_void_p)]),
    "mppi_comm_connect": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mppi_comm_probe": (C.c_int, [_H, C.c_void_p]),
    "mppi_comm_close": (C.c_int, [_H]),
    "mppi_get_costs": (C.c_int, [_H, _D]),
    "mppi_get_weights": (C.c_int, [_H, _D]),
    "mppi_sample_epsilon": (C.c_int, [_H, C.c_int64, C.c_void_p, C.c_void_p]),
    "mppi_set_iteration": (C.c_int, [_H, C.c_int64]),
    "mppi_set_noise_ring": (C.c_int, [_H, C.c_void_p, C.c_int32]),
    "mppi_rollout_viz": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_set_state": (C.c_int, [_H, _D]),
    "mppi_get_state": (C.c_int, [_H, _D]),
    "mppi_run_closed_loop": (C.c_int, [_H, C.c_int32, _D, C.POINTER(MppiStats), C.c_void_p]),
    "mppi_last_kernel_ms": (C.c_int, [_H, C.POINTER(C.c_float)]),
    "mppi_enable_timing": (C.c_int, [_H, C.c_int32]),
    "mppi_set_rollout_repeats": (C.c_int, [_H, C.c_int32]),
    "mppi_get_counters": (C.c_int, [_H, C.POINTER(C.c_int64)]),
    "mppi_comm_unique_id_bytes": (C.c_int, []),
    "mppi_comm_unique_id": (C.c_int, [C.c_void_p]),
    "mppi_comm_init": (C.c_int, [_H, C.c_void_p, C.c_int32, C.c_int32]),
    "mppi_get_rollout_layout": (C.c_int, [_H, C.POINTER(C.c_int32)]),
    "mppi_get_host_timing": (C.c_int, [_H, C.POINTER(C.c_double)]),
    "mppi_get_rollout_kernel": (C.c_int, [_H, C.c_char_p, C.c_int32]),
    "mppi_time_rollout_launch": (C.c_int, [_H, C.c_int32, C.c_int32, C.c_void_p, C.POINTER(C.c_double)]),
    "mppi_step_device_x0": (C.c_int, [_H, C.c_void_p, C.c_void_p, _D, _D, C.POINTER(MppiStats), C.c_void_p]),
    "mppi_eval_state_transition": (C.c_int, [_H, _D, _D, C.c_int32, _D]),
    "mppi_eval_clamp": (C.c_int, [_H, _D, C.c_int32, _D]),
    "mppi_eval_is_collided": (C.c_int, [_H, _D, C.c_int32, _D]),
    "mppi_eval_nearest_waypoint": (C.c_int, [_H, _D, C.c_int32, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32)]),
    "mppi_eval_cost": (C.c_int, [_H, C.c_int32, _D, C.c_int32, C.POINTER(C.c_int32), C.c_int32, _D, C.POINTER(C.c_int32)]),
    "mppi_eval_moving_average": (C.c_int, [_H, _D, _D]),
    "mppi_eval_weights": (C.c_int, [_H, _D, C.c_int32, _D]),
    # pytorch_mppi-style MPPI with built-in models (callback_mppi.py binds the config struct)
    "mppi_cb_last_error": (C.c_char_p, [_H]),
    "mppi_cb_create": (C.c_int, [C.c_void_p, C.POINTER(_H)]),
    "mppi_cb_destroy": (C.c_int, [_H]),
    "mppi_cb_set_nominal": (C.c_int, [_H, _D]),
    "mppi_cb_get_nominal": (C.c_int, [_H, _D]),
    "mppi_cb_command": (C.c_int, [_H, _D, C.c_void_p, C.c_int32, _D, C.c_void_p]),
    "mppi_cb_get_costs": (C.c_int, [_H, _D, _D]),
    "mppi_cb_eval": (C.c_int, [_H, C.c_int32, _D, _D, C.c_int32, C.c_int32, _D]),
    "mppi_cb_nominal_trajectory": (C.c_int, [_H, _D, _D]),
}

_lib = None


def load_library(path: str | None = None):
    """dlopen libmppi_hip.so and bind every prototype; raises if the library is absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise MppiError(ERR_NO_DEVICE, f"{p} not found: build it with __graft_entry__.build() "
                                       "(hipcc --offload-arch=gfx950); there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64.so.7; importing it first makes
    # the dynamic loader resolve this library's DT_NEEDED libamdhip64.so.7 to that same copy, so tensors
    # and streams handed over by data_ptr()/cuda_stream belong to the runtime the kernels launch on.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(p)
    ab_build = bool(os.environ.get("MPPI_LIB")) and path is None  # an older diagnostic build may lack the newest entry points
    for name, (res, args) in PROTOTYPES.items():
        if ab_build and not hasattr(lib, name):
            continue
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.mppi_abi_version() != ABI_VERSION and not ab_build:
        raise MppiError(ERR_BAD_ARG, "ABI version mismatch between _capi.py and libmppi_hip.so")
    if ab_build and lib.mppi_abi_version() < MIN_AB_ABI_VERSION:
        # an older diagnostic build may lack the newest entry points, but not with other struct layouts: mppi_config and
        # mppi_stats are what they are since this version (mppi_create also compares struct_size)
        raise MppiError(ERR_BAD_ARG, f"MPPI_LIB={p} has ABI version {lib.mppi_abi_version()}: its mppi_config / mppi_stats "
                                     f"layouts differ from this binding's (needs >= {MIN_AB_ABI_VERSION})")
    if path is None:
        _lib = lib
    return lib


def check(lib, handle, rc):
    if rc != OK:
        msg = lib.mppi_last_error(handle)
        raise MppiError(rc, msg.decode() if msg else "unknown error")
