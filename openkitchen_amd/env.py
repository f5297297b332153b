"""Python host side of the batched Environment step: thin, typed wrappers over the C ABI.

`Track` mirrors the reference's RaceTrack + TrackSegments (Environment/RaceTrack.h, TrackSegments.h) and
`BatchedEnvironment` mirrors Environment's step surface (Environment/Environment.h:31-75) for N agents at
once with device-resident state.  bench.py, the parity tests and the Python callers use these; the C++
callers use the classes in include/Environment/.  All compute happens in libokenv.so on the GPU.
"""
import ctypes as C
import os

import numpy as np

from . import _capi as capi

TRACK_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tracks")


def track_path(name):
    """Path of a bundled TUMFTM racetrack-database CSV (Austin, Silverstone, Monza, Spa)."""
    p = name if name.endswith(".csv") else os.path.join(TRACK_DIR, name + ".csv")
    if not os.path.exists(p):
        raise FileNotFoundError(p)
    return p


def default_ray_fan(num_rays):
    """angle_i = -70 + 140*i/(R-1) degrees, fp32 (generalises Agent.cpp:11-18)."""
    if num_rays == 1:
        return np.zeros(1, dtype=np.float32)
    i = np.arange(num_rays, dtype=np.float32)
    return (np.float32(-70.0) + np.float32(140.0) * i / np.float32(num_rays - 1)).astype(np.float32)


class Track:
    """RaceTrack(csv) + TrackSegments(track): centre line, headings, four boundary polylines, 4*P segments."""

    KEYS = ["x", "y", "wr", "wl", "heading", "li", "lo", "ri", "ro"]

    def queries(self, qx, qy):
        """RaceTrack::getNearestDistanceToTrackBoundary and RaceTrack::getDistanceToLaneCenter (reference
        Environment/RaceTrack.cpp:33-72) for arrays of query points: (distance to the nearest inner-boundary point [px],
        distance to the nearest centre-line point / lane width there)."""
        L = capi.load()
        qx = np.ascontiguousarray(qx, dtype=np.float32).reshape(-1)
        qy = np.ascontiguousarray(qy, dtype=np.float32).reshape(-1)
        assert qx.size == qy.size
        boundary, lane = np.zeros(qx.size, dtype=np.float32), np.zeros(qx.size, dtype=np.float32)
        h = C.c_void_p()
        capi.check(L.okenv_track_load(C.byref(h), self.path.encode()))
        try:
            capi.check(L.okenv_track_queries(h, capi.ptr(qx), capi.ptr(qy), qx.size, capi.ptr(boundary), capi.ptr(lane)))
        finally:
            L.okenv_track_free(h)
        return boundary, lane

    def __init__(self, name_or_path):
        L = capi.load()
        self.path = track_path(name_or_path)
        h = C.c_void_p()
        capi.check(L.okenv_track_load(C.byref(h), self.path.encode()))
        try:
            self.P = L.okenv_track_num_points(h)
            self.S = L.okenv_track_num_segments(h)
            for w, k in enumerate(self.KEYS):
                a = np.zeros(self.P if w < 5 else 2 * self.P, dtype=np.float32)
                capi.check(L.okenv_track_get(h, w, capi.ptr(a)))
                setattr(self, k, a)
            seg = np.zeros((self.S, 4), dtype=np.float32)
            capi.check(L.okenv_track_segments(h, capi.ptr(seg)))
            self.segments = seg
        finally:
            L.okenv_track_free(h)


class BatchedEnvironment:
    """N agents x R rays on one GPU.  State lives on the device; see include/okenv.h for field semantics."""

    def __init__(self, segments, num_agents, ray_angles_deg, device=0, flags=0, grid_cell=0.0, centerline=None):
        L = capi.load()
        seg = np.ascontiguousarray(segments, dtype=np.float32).reshape(-1, 4)
        rays = np.ascontiguousarray(ray_angles_deg, dtype=np.float32)
        self.N, self.R, self.S = int(num_agents), int(rays.size), int(seg.shape[0])
        self._h = C.c_void_p()
        capi.check(L.okenv_create(C.byref(self._h), capi.ptr(seg), self.S, self.N, self.R, capi.ptr(rays), int(device),
                                  int(flags), float(grid_cell)))
        self._L = L
        self.ray_angles_deg = rays
        if centerline is not None:
            self.set_centerline(*centerline)

    @classmethod
    def from_track(cls, track, num_agents, num_rays=None, ray_angles_deg=None, **kw):
        rays = default_ray_fan(num_rays) if ray_angles_deg is None else ray_angles_deg
        env = cls(track.segments, num_agents, rays, centerline=(track.x, track.y, track.heading), **kw)
        env.set_lane_bounds(track.li, track.ri)
        return env

    def close(self):
        if getattr(self, "_h", None):
            self._L.okenv_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- configuration ---------------------------------------------------------------------------
    def info(self):
        i = capi.OkenvInfo()
        capi.check(self._L.okenv_get_info(self._h, C.byref(i)), self._h)
        return {k: getattr(i, k) for k, _ in capi.OkenvInfo._fields_}

    def set_centerline(self, x, y, heading_deg):
        x, y, hd = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, heading_deg)]
        self.P = int(x.size)
        capi.check(self._L.okenv_set_centerline(self._h, capi.ptr(x), capi.ptr(y), capi.ptr(hd), self.P), self._h)

    def set_lane_bounds(self, left_inner_xy, right_inner_xy):
        """RaceTrack::left_bound_inner_ / right_bound_inner_ as xy pairs (resetAgent's lane randomisation)."""
        l = np.ascontiguousarray(left_inner_xy, dtype=np.float32).reshape(-1)
        r = np.ascontiguousarray(right_inner_xy, dtype=np.float32).reshape(-1)
        assert l.size == r.size and l.size % 2 == 0
        capi.check(self._L.okenv_set_lane_bounds(self._h, capi.ptr(l), capi.ptr(r), l.size // 2), self._h)

    def set_sensor_offset(self, off):
        capi.check(self._L.okenv_set_sensor_offset(self._h, float(off)), self._h)

    def set_stream(self, hip_stream_ptr):
        capi.check(self._L.okenv_set_stream(self._h, C.c_void_p(hip_stream_ptr)), self._h)

    def sync(self):
        capi.check(self._L.okenv_sync(self._h), self._h)

    # ---- state -----------------------------------------------------------------------------------
    def set(self, field, values):
        """values: numpy array (copied from host) or a torch device tensor of the field's dtype."""
        if isinstance(values, np.ndarray) or not hasattr(values, "data_ptr"):
            values = np.ascontiguousarray(values, dtype=capi.FIELD_DTYPE[field])
            n = self.N * self.R if field in capi.PER_RAY else self.N
            assert values.size == n, (capi.FIELD_NAMES[field], values.size, n)
        capi.check(self._L.okenv_set_field(self._h, field, capi.ptr(values)), self._h)

    def get(self, field, out=None):
        n = self.N * self.R if field in capi.PER_RAY else self.N
        if out is None:
            out = np.zeros(n, dtype=capi.FIELD_DTYPE[field])
        capi.check(self._L.okenv_get_field(self._h, field, capi.ptr(out)), self._h)
        if isinstance(out, np.ndarray) and field in capi.PER_RAY:
            return out.reshape(self.N, self.R)
        return out

    def snapshot(self):
        return {capi.FIELD_NAMES[f]: self.get(f) for f in range(19)}

    def set_actions(self, throttle, steer):
        if isinstance(throttle, np.ndarray) or not hasattr(throttle, "data_ptr"):
            throttle = np.ascontiguousarray(throttle, dtype=np.float32)
            steer = np.ascontiguousarray(steer, dtype=np.float32)
        capi.check(self._L.okenv_set_actions(self._h, capi.ptr(throttle), capi.ptr(steer)), self._h)

    def reset_agents(self, idx, x, y, rot_deg):
        idx = np.ascontiguousarray(idx, dtype=np.int32)
        x, y, rot = [np.ascontiguousarray(a, dtype=np.float32) for a in (x, y, rot_deg)]
        capi.check(self._L.okenv_reset_agents(self._h, capi.ptr(idx), capi.ptr(x), capi.ptr(y), capi.ptr(rot), idx.size),
                   self._h)

    def reset_random(self, idx=None, flags=capi.RESET_RANDOM_POINT, seed=0, epoch=0, agent_base=0):
        """Environment::resetAgent on the device for agents `idx` (None: all); flags = capi.RESET_*."""
        if idx is None:
            capi.check(self._L.okenv_reset_random(self._h, None, 0, int(flags), int(seed), int(epoch), int(agent_base)),
                       self._h)
            return
        if isinstance(idx, np.ndarray) or not hasattr(idx, "data_ptr"):
            idx = np.ascontiguousarray(idx, dtype=np.int32)
            n = idx.size
        else:
            n = idx.numel()
        capi.check(self._L.okenv_reset_random(self._h, capi.ptr(idx), int(n), int(flags), int(seed), int(epoch),
                                              int(agent_base)), self._h)

    def set_auto_reset(self, enabled, flags=capi.RESET_RANDOM_POINT, seed=0, agent_base=0):
        """While on, every step begins by re-placing the agents whose crashed_ flag is set (include/okenv.h)."""
        capi.check(self._L.okenv_set_auto_reset(self._h, int(bool(enabled)), int(flags), int(seed), int(agent_base)),
                   self._h)

    @property
    def step_count(self):
        v = C.c_uint32()
        capi.check(self._L.okenv_get_step_count(self._h, C.byref(v)), self._h)
        return v.value

    @step_count.setter
    def step_count(self, value):
        capi.check(self._L.okenv_set_step_count(self._h, int(value)), self._h)

    # ---- rollout bookkeeping of the CMA-ES / PPO style callers (include/okenv.h) -----------------------------
    def tracker_create(self, reward_kind):
        capi.check(self._L.okenv_tracker_create(self._h, int(reward_kind)), self._h)

    def tracker_begin(self):
        capi.check(self._L.okenv_tracker_begin(self._h), self._h)

    def tracker_update(self):
        capi.check(self._L.okenv_tracker_update(self._h), self._h)

    # ---- CMA-ES controllers ---------------------------------------------
    def controller_create(self, hidden=16):
        """Controller.cpp:3-23 for every agent: rays -> hidden -> hidden / 2 -> 2, tanh; returns the parameter count."""
        capi.check(self._L.okenv_controller_create(self._h, int(hidden)), self._h)
        n = C.c_int32()
        capi.check(self._L.okenv_controller_num_params(self._h, C.byref(n)), self._h)
        return int(n.value)

    def controller_set_params(self, params):
        """params [N, num_params]: numpy float32 array or a torch device tensor (torch parameters() order)."""
        if isinstance(params, np.ndarray) or not hasattr(params, "data_ptr"):
            params = np.ascontiguousarray(params, dtype=np.float32)
        capi.check(self._L.okenv_controller_set_params(self._h, capi.ptr(params)), self._h)

    def controller_act(self, throttle=100.0, steering_scale=5.0):
        """CmaEsAgent::updateAction for every agent (main_eigen.cpp:58-68)."""
        capi.check(self._L.okenv_controller_act(self._h, float(throttle), float(steering_scale)), self._h)

    def rollout_controller(self, n_steps, throttle=100.0, steering_scale=5.0):
        """n_steps iterations of the CMA-ES racers' inner loop (controller, Environment::step, fitness bookkeeping) in one launch."""
        capi.check(self._L.okenv_rollout_controller(self._h, int(n_steps), float(throttle), float(steering_scale)), self._h)

    def tracker_snapshot(self):
        return {capi.FIELD_NAMES[f]: self.get(f) for f in range(capi.F_REWARD, capi.F_EPISODE_RETURN + 1)}

    def field_device_ptr(self, field):
        """(address, bytes) of a library-owned device array; see okenv_field_device_ptr."""
        p, b = C.c_void_p(), C.c_uint64()
        capi.check(self._L.okenv_field_device_ptr(self._h, int(field), C.byref(p), C.byref(b)), self._h)
        return p.value, b.value

    def hits(self):
        out = np.zeros((self.N, self.R, 2), dtype=np.float32)
        capi.check(self._L.okenv_get_hits(self._h, capi.ptr(out)), self._h)
        return out

    def distances(self):
        out = np.zeros((self.N, self.R), dtype=np.float32)
        capi.check(self._L.okenv_get_distances(self._h, capi.ptr(out)), self._h)
        return out

    def flags(self):
        out = np.zeros(self.N, dtype=np.uint8)
        capi.check(self._L.okenv_get_flags(self._h, capi.ptr(out)), self._h)
        return out

    # ---- hot path --------------------------------------------------------------------------------
    def step(self, n_steps=1):
        capi.check(self._L.okenv_step(self._h, int(n_steps)), self._h)

    def collide(self):
        capi.check(self._L.okenv_collide(self._h), self._h)

    def rollout_random(self, n_steps, seed, agent_base=0, step_base=0):
        capi.check(self._L.okenv_rollout_random(self._h, int(n_steps), int(seed), int(agent_base), int(step_base)), self._h)

    def init_bench_state(self, agent_base=0, mode=capi.MODE_VELOCITY):
        capi.check(self._L.okenv_init_bench_state(self._h, int(agent_base), int(mode)), self._h)

    def nearest_track_idx(self, qx=None, qy=None):
        if qx is None:
            out = np.zeros(self.N, dtype=np.int32)
            capi.check(self._L.okenv_nearest_track_idx(self._h, None, None, 0, capi.ptr(out)), self._h)
            return out
        qx = np.ascontiguousarray(qx, dtype=np.float32)
        qy = np.ascontiguousarray(qy, dtype=np.float32)
        out = np.zeros(qx.size, dtype=np.int32)
        capi.check(self._L.okenv_nearest_track_idx(self._h, capi.ptr(qx), capi.ptr(qy), qx.size, capi.ptr(out)), self._h)
        return out

    # ---- EvolutionaryRacer on the device (include/okenv.h) -------------------------------------------------
    def policy_mlp_create(self, hidden=30, seed=1234, agent_base=0):
        capi.check(self._L.okenv_policy_mlp_create(self._h, int(hidden), int(seed), int(agent_base)), self._h)
        self.weights_per_agent = self._L.okenv_policy_mlp_weights_per_agent(self._h)

    def policy_weights(self):
        out = np.zeros((self.N, self.weights_per_agent), dtype=np.float32)
        capi.check(self._L.okenv_policy_mlp_get_weights(self._h, capi.ptr(out)), self._h)
        return out

    def set_policy_weights(self, w):
        if isinstance(w, np.ndarray):
            w = np.ascontiguousarray(w, dtype=np.float32)
        capi.check(self._L.okenv_policy_mlp_set_weights(self._h, capi.ptr(w)), self._h)

    def rollout_policy(self, n_steps):
        capi.check(self._L.okenv_rollout_policy(self._h, int(n_steps)), self._h)

    def alive_count(self):
        n = C.c_int32()
        capi.check(self._L.okenv_alive_count(self._h, C.byref(n)), self._h)
        return n.value

    # ---- episodes: step everybody until every agent has crashed, at the cost of the agents still alive (include/okenv.h) ----
    def episode_begin(self):
        capi.check(self._L.okenv_episode_begin(self._h), self._h)

    def episode_compact(self):
        """(agents alive, agents still stepped) -- also shrinks the grid of the next rollouts to the latter."""
        alive, listed = C.c_int32(), C.c_int32()
        capi.check(self._L.okenv_episode_compact(self._h, C.byref(alive), C.byref(listed)), self._h)
        return alive.value, listed.value

    def episode_tail_limit(self):
        """Longest list that is stepped one agent per workgroup (okenv_episode_tail_limit): from there on one rollout call may ask
        for all remaining steps."""
        n = C.c_int32()
        capi.check(self._L.okenv_episode_tail_limit(self._h, C.byref(n)), self._h)
        return n.value

    def episode_end(self):
        """(steps of the reference's loop, live agent-steps); leaves the state as that loop leaves it."""
        steps, live = C.c_int32(), C.c_uint64()
        capi.check(self._L.okenv_episode_end(self._h, C.byref(steps), C.byref(live)), self._h)
        return steps.value, live.value

    def off_grid_count(self):
        """(alive, all) agents outside the raycast grid's box: escaped through the boundaries, nothing left in sensor range."""
        a, b = C.c_int32(0), C.c_int32(0)
        capi.check(self._L.okenv_off_grid_count(self._h, C.byref(a), C.byref(b)), self._h)
        return int(a.value), int(b.value)

    def reset_all(self, x, y, rot_deg):
        capi.check(self._L.okenv_reset_all(self._h, float(x), float(y), float(rot_deg)), self._h)

    def ga_scores(self, out=None):
        if out is None:
            out = np.zeros(self.N, dtype=np.float32)
        capi.check(self._L.okenv_ga_scores(self._h, capi.ptr(out)), self._h)
        return out

    def ga_scores_into(self, device_tensor):
        """assignScores straight into a float32 CUDA tensor of N elements (device-to-device, on this handle's stream): the
        fitness vector never visits the host on its way into the per-generation all-gather."""
        assert device_tensor.is_cuda and device_tensor.is_contiguous() and device_tensor.numel() == self.N
        assert str(device_tensor.dtype) == "torch.float32"
        capi.check(self._L.okenv_ga_scores(self._h, C.c_void_p(device_tensor.data_ptr())), self._h)
        return device_tensor

    def ga_select_mate(self, seed, generation, agent_base=0):
        parents = np.zeros(5, dtype=np.int32)
        capi.check(self._L.okenv_ga_select_mate(self._h, int(seed), int(generation), int(agent_base), capi.ptr(parents)), self._h)
        return parents

    # ---- RLRacers/Q_Learning on the device (include/okenv.h) -------------------------------------------------
    def q_create(self):
        capi.check(self._L.okenv_q_create(self._h), self._h)

    def q_begin_episode(self, reset_idx):
        capi.check(self._L.okenv_q_begin_episode(self._h, int(reset_idx)), self._h)

    def rollout_q(self, n_steps, epsilon, seed, agent_base=0, step_base=0):
        capi.check(self._L.okenv_rollout_q(self._h, int(n_steps), float(epsilon), int(seed), int(agent_base), int(step_base)), self._h)

    def q_table(self):
        out = np.zeros((self.N, 243, 3), dtype=np.float32)
        capi.check(self._L.okenv_q_get_table(self._h, capi.ptr(out)), self._h)
        return out

    def set_q_table(self, table):
        t = np.ascontiguousarray(table, dtype=np.float32)
        capi.check(self._L.okenv_q_set_table(self._h, capi.ptr(t)), self._h)

    def q_state(self):
        s, a, p = (np.zeros(self.N, dtype=np.int32) for _ in range(3))
        capi.check(self._L.okenv_q_get_state(self._h, capi.ptr(s), capi.ptr(a), capi.ptr(p)), self._h)
        return s, a, p

    def q_table_sums(self):
        s, c = np.zeros(729, dtype=np.float32), np.zeros(729, dtype=np.float32)
        capi.check(self._L.okenv_q_table_sums(self._h, capi.ptr(s), capi.ptr(c)), self._h)
        return s, c

    def q_assign_mean(self, sums, counts):
        s = np.ascontiguousarray(sums, dtype=np.float32)
        c = np.ascontiguousarray(counts, dtype=np.float32)
        capi.check(self._L.okenv_q_assign_mean(self._h, capi.ptr(s), capi.ptr(c)), self._h)

    def q_share_knowledge(self):
        capi.check(self._L.okenv_q_share_knowledge(self._h), self._h)

    # ---- measurement / self-checks ------------------------------------------------------------------
    def work_stats(self):
        """{rays, tests, cells, points} the broad phase leaves for the current poses (okenv_work_stats)."""
        out = np.zeros(4, dtype=np.uint64)
        capi.check(self._L.okenv_work_stats(self._h, capi.ptr(out)), self._h)
        return dict(zip(("rays", "tests", "cells", "points"), (int(v) for v in out)))

    def work_stats_split(self):
        """The same for the walk with the front / back split, as the step kernels make it (okenv_work_stats_split): front and back
        walks together, plus the rays of a certified origin, the ambiguous front walks and the rays that walked the back image."""
        out = np.zeros(8, dtype=np.uint64)
        capi.check(self._L.okenv_work_stats_split(self._h, capi.ptr(out)), self._h)
        return dict(zip(("rays", "tests", "cells", "points", "certified", "ambiguous", "back_walked"), (int(v) for v in out[:7])))

    def set_timing(self, enabled):
        capi.check(self._L.okenv_set_timing(self._h, 1 if enabled else 0), self._h)

    def get_timing(self):
        ms, n = C.c_double(), C.c_uint64()
        capi.check(self._L.okenv_get_timing(self._h, C.byref(ms), C.byref(n)), self._h)
        return ms.value, n.value

    def debug_cast_rays(self, ox, oy, angle_rad):
        ox, oy, ang = [np.ascontiguousarray(a, dtype=np.float32) for a in (ox, oy, angle_rad)]
        out = np.zeros(ox.size, dtype=np.float32)
        capi.check(self._L.okenv_debug_cast_rays(self._h, capi.ptr(ox), capi.ptr(oy), capi.ptr(ang), ox.size, capi.ptr(out)),
                   self._h)
        return out


def debug_sincos(x, device=0):
    x = np.ascontiguousarray(x, dtype=np.float32)
    s, c = np.zeros_like(x), np.zeros_like(x)
    capi.check(capi.load().okenv_debug_sincos(int(device), capi.ptr(x), capi.ptr(s), capi.ptr(c), x.size))
    return s, c
