"""ctypes binding of the C ABI in include/okenv.h (libokenv.so, built by openkitchen_amd/buildlib.py).

Fails loudly if the shared object is missing or cannot be loaded: there is no Python or CPU fallback for
the hot path.
"""
import ctypes as C
import os

import numpy as np

from . import buildlib as _build

OKENV_OK = 0
ERR_NAMES = {0: "OK", -1: "INVALID", -2: "HIP", -3: "NO_DEVICE", -4: "IO", -5: "STATE"}

MODE_VELOCITY, MODE_ACCELERATION, MODE_MANUAL = 0, 1, 2
RESET_RANDOM_POINT, RESET_RANDOM_LANE, RESET_RANDOM_HEADING, RESET_ONLY_DONE = 1, 2, 4, 8
FLAG_NONE, FLAG_FORCE_GLOBAL_GRID, FLAG_BRUTE_FORCE = 0, 1, 2

(F_POS_X, F_POS_Y, F_ROT, F_SPEED, F_ACC, F_THROTTLE, F_STEER, F_MODE, F_CRASHED, F_TIMED_OUT, F_DISP_CTR, F_DISP_X,
 F_DISP_Y, F_DISP_TO, F_HIT_X, F_HIT_Y, F_REL_X, F_REL_Y, F_DIST) = range(19)
FIELD_NAMES = ["pos_x", "pos_y", "rot", "speed", "acc", "thr", "steer", "mode", "crashed", "timed_out", "disp_ctr",
               "disp_x", "disp_y", "disp_to", "hit_x", "hit_y", "rel_x", "rel_y", "dist"]
FIELD_DTYPE = [np.float32] * 7 + [np.uint8] * 3 + [np.uint32, np.float32, np.float32, np.uint8] + [np.float32] * 5
PER_RAY = {F_HIT_X, F_HIT_Y, F_REL_X, F_REL_Y, F_DIST}
# rollout bookkeeping fields (exist after okenv_tracker_create)
F_REWARD, F_FITNESS, F_TRACK_IDX, F_EPISODE_STEPS, F_EPISODE_RETURN, F_PREV_CRASHED = range(19, 25)
FIELD_NAMES += ["reward", "fitness", "track_idx", "episode_steps", "episode_return", "prev_crashed"]
FIELD_DTYPE += [np.float32, np.float32, np.int32, np.uint32, np.float32, np.uint8]
REWARD_STEP, REWARD_PROGRESS = 0, 1

# every symbol include/okenv.h declares (tests/test_capi_symbols.py checks the library exports them all)
SYMBOLS = [
    "okenv_create", "okenv_destroy", "okenv_get_info", "okenv_last_error", "okenv_set_sensor_offset",
    "okenv_set_centerline", "okenv_set_stream", "okenv_sync", "okenv_set_field", "okenv_get_field",
    "okenv_upload_state", "okenv_download_state", "okenv_set_actions", "okenv_reset_agents", "okenv_get_hits",
    "okenv_get_distances", "okenv_get_flags", "okenv_step", "okenv_collide", "okenv_rollout_random",
    "okenv_init_bench_state", "okenv_nearest_track_idx", "okenv_set_timing", "okenv_get_timing", "okenv_track_load",
    "okenv_track_free", "okenv_track_num_points", "okenv_track_num_segments", "okenv_track_get",
    "okenv_track_segments", "okenv_track_queries", "okenv_debug_sincos", "okenv_debug_cast_rays", "okenv_policy_mlp_create",
    "okenv_policy_mlp_weights_per_agent", "okenv_policy_mlp_get_weights", "okenv_policy_mlp_set_weights",
    "okenv_rollout_policy", "okenv_alive_count", "okenv_reset_all", "okenv_ga_scores", "okenv_ga_select_mate",
    "okenv_q_create", "okenv_q_begin_episode", "okenv_rollout_q", "okenv_q_get_table", "okenv_q_set_table", "okenv_q_get_state",
    "okenv_q_table_sums", "okenv_q_assign_mean", "okenv_q_share_knowledge",
    "okenv_set_lane_bounds", "okenv_reset_random", "okenv_set_auto_reset", "okenv_get_step_count",
    "okenv_set_step_count", "okenv_field_device_ptr", "okenv_tracker_create", "okenv_tracker_begin",
    "okenv_tracker_update", "okenv_step_packed",
    "okenv_controller_create", "okenv_controller_num_params", "okenv_controller_set_params", "okenv_controller_act", "okenv_rollout_controller",
    "okenv_episode_begin", "okenv_episode_compact", "okenv_episode_end", "okenv_episode_tail_limit", "okenv_work_stats",
    "okenv_ga_scores_device", "okenv_get_stream", "okenv_off_grid_count", "okenv_work_stats_split",
]


class OkenvInfo(C.Structure):
    _fields_ = [("num_agents", C.c_int32), ("num_rays", C.c_int32), ("num_segments", C.c_int32),
                ("grid_nx", C.c_int32), ("grid_ny", C.c_int32), ("grid_cell", C.c_float), ("grid_refs", C.c_int32),
                ("grid_in_lds", C.c_int32), ("lds_bytes", C.c_int32), ("block_threads", C.c_int32),
                ("grid_blocks", C.c_int32), ("lanes_per_agent", C.c_int32), ("device", C.c_int32),
                ("agents_per_block", C.c_int32), ("packed_resident", C.c_int32), ("packed_resident_steps", C.c_int32),
                ("packed_fallbacks", C.c_int32), ("compute_units", C.c_int32), ("front_back_bytes", C.c_int32),
                ("back_segments", C.c_int32)]


class OkenvError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("okenv error %s (%d): %s" % (ERR_NAMES.get(code, "?"), code, msg))
        self.code = code


_lib = None


def lib_path():
    return _build.LIB_PATH


def load(build_if_missing=True):
    """Loads libokenv.so; raises if it is absent and cannot be built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        if not build_if_missing:
            raise OkenvError(-3, "libokenv.so is missing at %s (run python -m openkitchen_amd.buildlib)" % path)
        _build.build()
    # One HIP runtime per process: PyTorch-ROCm bundles its own libamdhip64.so.7 and opens it by path, so
    # if libokenv.so pulled in /opt/rocm's copy first the process would hold two runtimes and whichever
    # initialises second sees "no HIP device".  Importing torch first makes libokenv.so's DT_NEEDED
    # libamdhip64.so.7 resolve (by soname) to the copy torch already loaded.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(path)
    vp, i32, u32, f32 = C.c_void_p, C.c_int32, C.c_uint32, C.c_float
    L.okenv_create.argtypes = [C.POINTER(vp), vp, i32, i32, i32, vp, i32, u32, f32]
    L.okenv_destroy.argtypes = [vp]
    L.okenv_get_info.argtypes = [vp, C.POINTER(OkenvInfo)]
    L.okenv_last_error.argtypes = [vp]
    L.okenv_last_error.restype = C.c_char_p
    L.okenv_set_sensor_offset.argtypes = [vp, f32]
    L.okenv_set_centerline.argtypes = [vp, vp, vp, vp, i32]
    L.okenv_set_stream.argtypes = [vp, vp]
    L.okenv_sync.argtypes = [vp]
    L.okenv_set_field.argtypes = [vp, i32, vp]
    L.okenv_get_field.argtypes = [vp, i32, vp]
    L.okenv_upload_state.argtypes = [vp, vp]
    L.okenv_download_state.argtypes = [vp, vp]
    L.okenv_set_actions.argtypes = [vp, vp, vp]
    L.okenv_reset_agents.argtypes = [vp, vp, vp, vp, vp, i32]
    L.okenv_get_hits.argtypes = [vp, vp]
    L.okenv_get_distances.argtypes = [vp, vp]
    L.okenv_get_flags.argtypes = [vp, vp]
    L.okenv_step.argtypes = [vp, i32]
    L.okenv_collide.argtypes = [vp]
    L.okenv_rollout_random.argtypes = [vp, i32, u32, u32, u32]
    L.okenv_init_bench_state.argtypes = [vp, u32, i32]
    L.okenv_nearest_track_idx.argtypes = [vp, vp, vp, i32, vp]
    L.okenv_set_timing.argtypes = [vp, i32]
    L.okenv_get_timing.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    L.okenv_track_load.argtypes = [C.POINTER(vp), C.c_char_p]
    L.okenv_track_free.argtypes = [vp]
    L.okenv_track_num_points.argtypes = [vp]
    L.okenv_track_num_segments.argtypes = [vp]
    L.okenv_track_get.argtypes = [vp, i32, vp]
    L.okenv_track_segments.argtypes = [vp, vp]
    L.okenv_track_queries.argtypes = [vp, vp, vp, i32, vp, vp]
    L.okenv_debug_sincos.argtypes = [i32, vp, vp, vp, i32]
    L.okenv_debug_cast_rays.argtypes = [vp, vp, vp, vp, i32, vp]
    L.okenv_policy_mlp_create.argtypes = [vp, i32, u32, u32]
    L.okenv_policy_mlp_weights_per_agent.argtypes = [vp]
    L.okenv_policy_mlp_get_weights.argtypes = [vp, vp]
    L.okenv_policy_mlp_set_weights.argtypes = [vp, vp]
    L.okenv_rollout_policy.argtypes = [vp, i32]
    L.okenv_alive_count.argtypes = [vp, C.POINTER(i32)]
    L.okenv_reset_all.argtypes = [vp, f32, f32, f32]
    L.okenv_ga_scores.argtypes = [vp, vp]
    L.okenv_ga_select_mate.argtypes = [vp, u32, u32, u32, vp]
    L.okenv_q_create.argtypes = [vp]
    L.okenv_q_begin_episode.argtypes = [vp, i32]
    L.okenv_rollout_q.argtypes = [vp, i32, f32, u32, u32, u32]
    L.okenv_q_get_table.argtypes = [vp, vp]
    L.okenv_q_set_table.argtypes = [vp, vp]
    L.okenv_q_get_state.argtypes = [vp, vp, vp, vp]
    L.okenv_q_table_sums.argtypes = [vp, vp, vp]
    L.okenv_q_assign_mean.argtypes = [vp, vp, vp]
    L.okenv_q_share_knowledge.argtypes = [vp]
    L.okenv_set_lane_bounds.argtypes = [vp, vp, vp, i32]
    L.okenv_reset_random.argtypes = [vp, vp, i32, u32, u32, u32, u32]
    L.okenv_set_auto_reset.argtypes = [vp, i32, u32, u32, u32]
    L.okenv_get_step_count.argtypes = [vp, C.POINTER(u32)]
    L.okenv_set_step_count.argtypes = [vp, u32]
    L.okenv_step_packed.argtypes = [vp, vp, vp, vp, u32]
    L.okenv_controller_create.argtypes = [vp, C.c_int32]
    L.okenv_controller_num_params.argtypes = [vp, C.POINTER(C.c_int32)]
    L.okenv_controller_set_params.argtypes = [vp, vp]
    L.okenv_controller_act.argtypes = [vp, C.c_float, C.c_float]
    L.okenv_rollout_controller.argtypes = [vp, C.c_int32, C.c_float, C.c_float]
    L.okenv_tracker_create.argtypes = [vp, i32]
    L.okenv_tracker_begin.argtypes = [vp]
    L.okenv_tracker_update.argtypes = [vp]
    L.okenv_field_device_ptr.argtypes = [vp, i32, C.POINTER(vp), C.POINTER(C.c_uint64)]
    L.okenv_work_stats.argtypes = [vp, vp]
    L.okenv_work_stats_split.argtypes = [vp, vp]
    L.okenv_episode_begin.argtypes = [vp]
    L.okenv_episode_compact.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    L.okenv_episode_end.argtypes = [vp, C.POINTER(i32), C.POINTER(C.c_uint64)]
    L.okenv_episode_tail_limit.argtypes = [vp, C.POINTER(i32)]
    L.okenv_ga_scores_device.argtypes = [vp, C.POINTER(vp)]
    L.okenv_get_stream.argtypes = [vp, C.POINTER(vp)]
    L.okenv_off_grid_count.argtypes = [vp, C.POINTER(i32), C.POINTER(i32)]
    _lib = L
    return L


def check(rc, handle=None):
    if rc != OKENV_OK:
        msg = load().okenv_last_error(handle)
        raise OkenvError(rc, msg.decode() if msg else "")
    return rc


def ptr(a):
    """Raw pointer of a numpy array, a torch tensor (host or device) or an int address."""
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return C.c_void_p(a.ctypes.data)
    if hasattr(a, "data_ptr"):  # torch tensor
        assert a.is_contiguous()
        return C.c_void_p(a.data_ptr())
    raise TypeError("unsupported buffer type %r" % type(a))
