"""ctypes bindings for the CPU oracle (oracle/libokenv_oracle.so) and, when built, the reference shim
(oracle/_ref/libokref.so).  Test infrastructure only -- nothing under openkitchen_amd/ imports this."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
# OKENV_SANITIZE=1 (tests/test_sanitizers.py, in a child process with libasan preloaded): the ASan + UBSan build of the same source
SANITIZE = os.environ.get("OKENV_SANITIZE") == "1"
SAN_FLAGS = ["-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-fno-omit-frame-pointer", "-g", "-O1"]
ORACLE_SO = os.path.join(ORACLE_DIR, "libokenv_oracle_san.so" if SANITIZE else "libokenv_oracle.so")
REF_SO = os.path.join(ORACLE_DIR, "_ref", "libokref.so")
TRACK_DIR = os.path.join(ROOT, "openkitchen_amd", "tracks")

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")

# field ids (same numbering as include/okenv.h OKENV_F_*)
(F_POS_X, F_POS_Y, F_ROT, F_SPEED, F_ACC, F_THR, F_STEER, F_MODE, F_CRASHED, F_TIMED_OUT, F_DISP_CTR, F_DISP_X,
 F_DISP_Y, F_DISP_TO, F_HIT_X, F_HIT_Y, F_REL_X, F_REL_Y, F_DIST) = range(19)
FIELD_DTYPE = {
    F_POS_X: np.float32, F_POS_Y: np.float32, F_ROT: np.float32, F_SPEED: np.float32, F_ACC: np.float32,
    F_THR: np.float32, F_STEER: np.float32, F_MODE: np.uint8, F_CRASHED: np.uint8, F_TIMED_OUT: np.uint8,
    F_DISP_CTR: np.uint32, F_DISP_X: np.float32, F_DISP_Y: np.float32, F_DISP_TO: np.uint8, F_HIT_X: np.float32,
    F_HIT_Y: np.float32, F_REL_X: np.float32, F_REL_Y: np.float32, F_DIST: np.float32,
}
PER_RAY = {F_HIT_X, F_HIT_Y, F_REL_X, F_REL_Y, F_DIST}
# rollout bookkeeping fields (exist after tracker_create)
F_REWARD, F_FITNESS, F_TRACK_IDX, F_EPISODE_STEPS, F_EPISODE_RETURN = range(19, 24)
FIELD_DTYPE.update({F_REWARD: np.float32, F_FITNESS: np.float32, F_TRACK_IDX: np.int32, F_EPISODE_STEPS: np.uint32,
                    F_EPISODE_RETURN: np.float32})
TRACKER_FIELDS = {"reward": F_REWARD, "fitness": F_FITNESS, "track_idx": F_TRACK_IDX, "episode_steps": F_EPISODE_STEPS,
                  "episode_return": F_EPISODE_RETURN}
FIELD_NAMES = ["pos_x", "pos_y", "rot", "speed", "acc", "thr", "steer", "mode", "crashed", "timed_out", "disp_ctr",
               "disp_x", "disp_y", "disp_to", "hit_x", "hit_y", "rel_x", "rel_y", "dist"]


def build_oracle(with_ref=True):
    subprocess.run(["make", "-s", "-C", ORACLE_DIR] + (["san"] if SANITIZE else []), check=True)
    if with_ref and not SANITIZE and os.path.isdir("/root/reference/Environment"):
        subprocess.run(["make", "-s", "-C", ORACLE_DIR, "ref"], check=True)
        if os.path.exists(os.path.join(ROOT, "openkitchen_amd", "libokenv.so")):
            subprocess.run(["make", "-s", "-C", ORACLE_DIR, "refbind"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(ORACLE_SO):
            build_oracle(with_ref=False)
        L = C.CDLL(ORACLE_SO)
        L.oracle_track_load.restype = C.c_void_p
        L.oracle_track_load.argtypes = [C.c_char_p]
        L.oracle_track_free.argtypes = [C.c_void_p]
        L.oracle_track_num_points.argtypes = [C.c_void_p]
        L.oracle_track_get.argtypes = [C.c_void_p, C.c_int, f32p]
        L.oracle_track_segments.argtypes = [C.c_void_p, f32p]
        L.oracle_nearest_track_idx.argtypes = [f32p, f32p, C.c_int, f32p, f32p, C.c_int, i32p]
        L.oracle_boundary_distance.argtypes = [f32p, f32p, C.c_int, f32p, f32p, C.c_int, f32p]
        L.oracle_lane_center_distance.argtypes = [f32p, f32p, f32p, f32p, C.c_int, f32p, f32p, C.c_int, f32p]
        L.oracle_env_create.restype = C.c_void_p
        L.oracle_env_create.argtypes = [f32p, C.c_int, C.c_int, C.c_int, f32p]
        L.oracle_env_destroy.argtypes = [C.c_void_p]
        L.oracle_env_set_centerline.argtypes = [C.c_void_p, f32p, f32p, f32p, C.c_int]
        L.oracle_env_set_sensor_offset.argtypes = [C.c_void_p, C.c_float]
        L.oracle_env_set_lane_bounds.argtypes = [C.c_void_p, f32p, f32p, C.c_int]
        L.oracle_env_reset_random.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32,
                                              C.c_uint32]
        L.oracle_env_set_auto_reset.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_env_get_step_count.restype = C.c_uint32
        L.oracle_env_get_step_count.argtypes = [C.c_void_p]
        L.oracle_env_set_step_count.argtypes = [C.c_void_p, C.c_uint32]
        L.oracle_tracker_create.argtypes = [C.c_void_p, C.c_int]
        L.oracle_tracker_begin.argtypes = [C.c_void_p]
        L.oracle_tracker_update.argtypes = [C.c_void_p]
        L.oracle_env_set_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_env_get_field.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.oracle_env_reset_agents.argtypes = [C.c_void_p, i32p, f32p, f32p, f32p, C.c_int]
        L.oracle_env_collide.argtypes = [C.c_void_p]
        L.oracle_env_step.argtypes = [C.c_void_p, C.c_int]
        L.oracle_set_threads.argtypes = [C.c_int]
        L.oracle_env_move_only.argtypes = [C.c_void_p]
        L.oracle_env_rollout_random.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_env_rollout_random_mt.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        L.oracle_env_init_bench_state.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
        L.oracle_sincosf.argtypes = [f32p, f32p, f32p, C.c_int]
        L.oracle_ga_decode.argtypes = [f32p, C.c_int, f32p, f32p, f32p, f32p]
        L.oracle_tanhf.argtypes = [f32p, f32p, C.c_int]
        L.oracle_env_controller_act.argtypes = [C.c_void_p, f32p, C.c_int, C.c_float, C.c_float]
        L.oracle_env_controller_act.restype = C.c_int
        L.oracle_set_trig_mode.argtypes = [C.c_int]
        L.oracle_cast_ray.restype = C.c_float
        L.oracle_cast_ray.argtypes = [C.c_float, C.c_float, C.c_float, f32p, C.c_int]
        L.oracle_philox.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                    C.POINTER(C.c_uint32)]
        L.oracle_ga_create.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint32]
        L.oracle_ga_weights_per_agent.argtypes = [C.c_void_p]
        L.oracle_ga_get_weights.argtypes = [C.c_void_p, f32p]
        L.oracle_ga_set_weights.argtypes = [C.c_void_p, f32p]
        L.oracle_env_rollout_policy.argtypes = [C.c_void_p, C.c_int]
        L.oracle_env_alive_count.argtypes = [C.c_void_p]
        L.oracle_env_reset_all.argtypes = [C.c_void_p, C.c_float, C.c_float, C.c_float]
        L.oracle_ga_scores.argtypes = [C.c_void_p, f32p]
        L.oracle_ga_select_mate.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, i32p]
        L.oracle_q_create.argtypes = [C.c_void_p]
        L.oracle_q_begin_episode.argtypes = [C.c_void_p, C.c_int]
        L.oracle_rollout_q.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_uint32, C.c_uint32, C.c_uint32]
        L.oracle_q_get_table.argtypes = [C.c_void_p, f32p]
        L.oracle_q_set_table.argtypes = [C.c_void_p, f32p]
        L.oracle_q_get_state.argtypes = [C.c_void_p, i32p, i32p, i32p]
        _lib = L
    return _lib


def have_ref():
    return os.path.exists(REF_SO)


_ref = None


def ref():
    global _ref
    if _ref is None:
        L = C.CDLL(REF_SO)
        L.ref_track_load.restype = C.c_void_p
        L.ref_track_load.argtypes = [C.c_char_p]
        L.ref_track_free.argtypes = [C.c_void_p]
        L.ref_track_num_points.argtypes = [C.c_void_p]
        L.ref_track_get.argtypes = [C.c_void_p, C.c_int, f32p]
        L.ref_nearest_track_idx.argtypes = [C.c_void_p, f32p, f32p, C.c_int, i32p]
        L.ref_track_queries.argtypes = [C.c_void_p, f32p, f32p, C.c_int, f32p, f32p]
        L.ref_agent_rollout.argtypes = [C.c_int, C.c_float, C.c_float, C.c_float, f32p, f32p, C.c_int, f32p, f32p,
                                        f32p, f32p, f32p]
        L.ref_agent_reset_probe.argtypes = [C.c_float, C.c_float, C.c_float, f32p]
        L.ref_agent_default_rays.argtypes = [f32p, C.c_int]
        L.ref_layout_facts.argtypes = [u32p]
        _ref = L
    return _ref


def track_path(name):
    return os.path.join(TRACK_DIR, name + ".csv")


class Track:
    """Track geometry from either the oracle or the reference shim (same accessor layout)."""

    KEYS = ["x", "y", "wr", "wl", "heading", "li", "lo", "ri", "ro"]

    def __init__(self, name_or_path, source="oracle"):
        path = name_or_path if name_or_path.endswith(".csv") else track_path(name_or_path)
        self.path = path
        if source == "oracle":
            L = lib()
            h = L.oracle_track_load(path.encode())
            assert h, "oracle_track_load failed for " + path
            self.P = L.oracle_track_num_points(h)
            get, free = L.oracle_track_get, L.oracle_track_free
        else:
            L = ref()
            h = L.ref_track_load(path.encode())
            self.P = L.ref_track_num_points(h)
            get, free = L.ref_track_get, L.ref_track_free
        for w, k in enumerate(self.KEYS):
            a = np.zeros(self.P if w < 5 else 2 * self.P, dtype=np.float32)
            assert get(h, w, a) == 0
            setattr(self, k, a)
        if source == "oracle":
            seg = np.zeros(4 * 4 * self.P, dtype=np.float32)
            self.S = lib().oracle_track_segments(h, seg)
            self.segments = seg[: 4 * self.S].reshape(self.S, 4).copy()
        free(h)


def default_ray_fan(R):
    """angle_i = -70 + 140*i/(R-1) degrees in fp32."""
    if R == 1:
        return np.zeros(1, dtype=np.float32)
    i = np.arange(R, dtype=np.float32)
    return (np.float32(-70.0) + np.float32(140.0) * i / np.float32(R - 1)).astype(np.float32)


class OracleEnv:
    def __init__(self, segments, N, R, ray_deg, centerline=None):
        L = lib()
        seg = np.ascontiguousarray(segments, dtype=np.float32).reshape(-1)
        self.N, self.R, self.S = N, R, seg.size // 4
        self.h = L.oracle_env_create(seg, self.S, N, R, np.ascontiguousarray(ray_deg, dtype=np.float32))
        if centerline is not None:
            cx, cy, ch = [np.ascontiguousarray(a, dtype=np.float32) for a in centerline]
            L.oracle_env_set_centerline(self.h, cx, cy, ch, cx.size)

    def __del__(self):
        try:
            lib().oracle_env_destroy(self.h)
        except Exception:
            pass

    def set(self, f, arr):
        a = np.ascontiguousarray(arr, dtype=FIELD_DTYPE[f])
        n = self.N * self.R if f in PER_RAY else self.N
        assert a.size == n, (f, a.size, n)
        assert lib().oracle_env_set_field(self.h, f, a.ctypes.data_as(C.c_void_p)) == 0

    def get(self, f):
        n = self.N * self.R if f in PER_RAY else self.N
        a = np.zeros(n, dtype=FIELD_DTYPE[f])
        assert lib().oracle_env_get_field(self.h, f, a.ctypes.data_as(C.c_void_p)) == 0
        return a.reshape(self.N, self.R) if f in PER_RAY else a

    def snapshot(self):
        return {FIELD_NAMES[f]: self.get(f) for f in range(19)}

    def reset_agents(self, idx, x, y, rot):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        lib().oracle_env_reset_agents(self.h, idx, np.ascontiguousarray(x, dtype=np.float32),
                                      np.ascontiguousarray(y, dtype=np.float32),
                                      np.ascontiguousarray(rot, dtype=np.float32), idx.size)

    def set_lane_bounds(self, left_inner_xy, right_inner_xy):
        l = np.ascontiguousarray(left_inner_xy, dtype=np.float32).reshape(-1)
        r = np.ascontiguousarray(right_inner_xy, dtype=np.float32).reshape(-1)
        lib().oracle_env_set_lane_bounds(self.h, l, r, l.size // 2)

    def reset_random(self, idx=None, flags=1, seed=0, epoch=0, agent_base=0):
        if idx is None:
            lib().oracle_env_reset_random(self.h, None, 0, flags, seed, epoch, agent_base)
        else:
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            lib().oracle_env_reset_random(self.h, idx.ctypes.data_as(C.c_void_p), idx.size, flags, seed, epoch, agent_base)

    def set_auto_reset(self, enabled, flags=1, seed=0, agent_base=0):
        lib().oracle_env_set_auto_reset(self.h, int(bool(enabled)), flags, seed, agent_base)

    @property
    def step_count(self):
        return lib().oracle_env_get_step_count(self.h)

    @step_count.setter
    def step_count(self, v):
        lib().oracle_env_set_step_count(self.h, int(v))

    def tracker_create(self, kind):
        lib().oracle_tracker_create(self.h, int(kind))

    def tracker_begin(self):
        lib().oracle_tracker_begin(self.h)

    def tracker_update(self):
        lib().oracle_tracker_update(self.h)

    def tracker_snapshot(self):
        return {k: self.get(f) for k, f in TRACKER_FIELDS.items()}

    def step(self, n=1):
        lib().oracle_env_step(self.h, n)

    def collide(self):
        lib().oracle_env_collide(self.h)

    def move_only(self):
        lib().oracle_env_move_only(self.h)

    def init_bench_state(self, agent_base=0, mode=0):
        lib().oracle_env_init_bench_state(self.h, agent_base, mode)

    def rollout_random(self, n_steps, seed, agent_base=0, step_base=0, threads=1):
        if threads > 1:
            lib().oracle_env_rollout_random_mt(self.h, n_steps, seed, agent_base, step_base, threads)
        else:
            lib().oracle_env_rollout_random(self.h, n_steps, seed, agent_base, step_base)


class OracleGA:
    """EvolutionaryRacer helpers on top of an OracleEnv (mirror of BatchedEnvironment's policy/GA methods)."""

    def __init__(self, env, hidden=30, seed=1234, agent_base=0):
        self.env = env
        lib().oracle_ga_create(env.h, hidden, seed, agent_base)
        self.per = lib().oracle_ga_weights_per_agent(env.h)

    def weights(self):
        w = np.zeros(self.env.N * self.per, dtype=np.float32)
        lib().oracle_ga_get_weights(self.env.h, w)
        return w.reshape(self.env.N, self.per)

    def set_weights(self, w):
        lib().oracle_ga_set_weights(self.env.h, np.ascontiguousarray(w, dtype=np.float32).reshape(-1))

    def rollout_policy(self, n):
        lib().oracle_env_rollout_policy(self.env.h, n)

    def alive_count(self):
        return lib().oracle_env_alive_count(self.env.h)

    def reset_all(self, x, y, rot):
        lib().oracle_env_reset_all(self.env.h, x, y, rot)

    def scores(self):
        out = np.zeros(self.env.N, dtype=np.float32)
        lib().oracle_ga_scores(self.env.h, out)
        return out

    def select_mate(self, seed, generation, agent_base=0):
        parents = np.zeros(5, dtype=np.int32)
        lib().oracle_ga_select_mate(self.env.h, seed, generation, agent_base, parents)
        return parents


class OracleQ:
    """RLRacers/Q_Learning helpers on top of an OracleEnv."""

    def __init__(self, env):
        self.env = env
        lib().oracle_q_create(env.h)

    def begin_episode(self, reset_idx):
        lib().oracle_q_begin_episode(self.env.h, int(reset_idx))

    def rollout(self, n, epsilon, seed, agent_base=0, step_base=0):
        lib().oracle_rollout_q(self.env.h, n, float(epsilon), seed, agent_base, step_base)

    def table(self):
        t = np.zeros(self.env.N * 243 * 3, dtype=np.float32)
        lib().oracle_q_get_table(self.env.h, t)
        return t.reshape(self.env.N, 243, 3)

    def state(self):
        s, a, p = (np.zeros(self.env.N, dtype=np.int32) for _ in range(3))
        lib().oracle_q_get_state(self.env.h, s, a, p)
        return s, a, p
