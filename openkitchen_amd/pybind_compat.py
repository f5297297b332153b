"""Drop-in for the reference's pybind11 module `open_kitchen_pybind` (reference Pybind/bindings.cpp:19-79): a single-agent
`Environment(race_track_path, draw_rays, hidden_window)` with `set_action`, `step`, `get_render_target[_info]`, hosted on
the batched device environment.  The window and the image render target are out of scope (the render target is an
all-zero RGBA buffer of the window size, as ScreenGrabber's headless stub returns); the lidar observation and the flags,
which the reference binding does not expose, are available as properties.

    import openkitchen_amd.pybind_compat as ok        # was: import open_kitchen_pybind as ok
    env = ok.Environment("tracks/Montreal.csv", draw_rays=False, hidden_window=True)
    env.step(); env.set_action(30.0, 0.0)
"""
import numpy as np

from . import _capi as capi
from .env import BatchedEnvironment, Track

_WINDOW = (1400, 1600, 4)  # kScreenHeight, kScreenWidth, RGBA (reference Environment/Typedefs.h:7-8)


class RenderTargetInfo:
    def __init__(self):
        self.height, self.width, self.channels = _WINDOW

    def row_bytes(self):
        return self.width * self.channels


class Environment:
    def __init__(self, race_track_path, draw_rays=True, hidden_window=True, seed=None):
        self._track = Track(race_track_path)
        fan = np.arange(-70, 71, 10, dtype=np.float32)  # Agent's default fan (reference Environment/Agent.cpp:11-18)
        self._env = BatchedEnvironment(self._track.segments, 1, fan, centerline=(self._track.x, self._track.y, self._track.heading))
        rng = np.random.default_rng(seed)
        idx = int(rng.integers(0, self._track.P))  # pickRandomResetTrackIdx (bindings.cpp:29)
        self._env.reset_agents([0], [self._track.x[idx]], [self._track.y[idx]], [self._track.heading[idx]])
        self._action = (0.0, 0.0)

    def set_action(self, throttle_delta, steering_delta):
        self._action = (float(throttle_delta), float(steering_delta))

    def step(self):
        self._env.set_actions(np.array([self._action[0]], dtype=np.float32), np.array([self._action[1]], dtype=np.float32))
        self._env.step(1)

    def get_render_target(self):
        return bytes(_WINDOW[0] * _WINDOW[1] * _WINDOW[2])

    def get_render_target_info(self):
        return RenderTargetInfo()

    # ---- not in the reference binding: the lidar observation and flags ------------------------------------------------
    @property
    def sensor_hits(self):
        return self._env.hits()[0]

    @property
    def distances(self):
        return self._env.distances()[0]

    @property
    def crashed(self):
        return bool(self._env.flags()[0] & 1)

    @property
    def pose(self):
        s = self._env
        return float(s.get(capi.F_POS_X)[0]), float(s.get(capi.F_POS_Y)[0]), float(s.get(capi.F_ROT)[0])
