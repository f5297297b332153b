"""Batched Python binding on device tensors: `step(actions) -> obs, done`.

The reference's Python surface is `open_kitchen_pybind.Environment(race_track_path, draw_rays, hidden_window)` with
`set_action(throttle, steering)` and `step()` for ONE agent behind a window (Pybind/bindings.cpp:19-79,
Pybind/example.py:10-20).  `VectorEnvironment` keeps those names and meanings for N agents and hands out the
library-owned struct-of-arrays state as zero-copy torch tensors (okenv_field_device_ptr), so a PyTorch-ROCm policy
reads observations and writes actions without a host round trip; everything is enqueued on torch's current stream.
The tensors also speak DLPack (`torch.Tensor.__dlpack__`) for other consumers.

PyTorch is plumbing here (device memory, streams); the step itself is the HIP kernel behind okenv_step.
"""
import torch

from . import _capi as capi
from .env import BatchedEnvironment, Track, default_ray_fan

_TYPESTR = {torch.float32: "<f4", torch.uint8: "|u1", torch.uint32: "<u4", torch.int32: "<i4"}


class _DeviceArray:
    """`__cuda_array_interface__` view of memory the okenv handle owns (kept alive through `owner`)."""

    def __init__(self, address, shape, dtype, owner):
        self.owner = owner
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": _TYPESTR[dtype], "data": (address, False),
                                         "version": 2, "strides": None}


class VectorEnvironment:
    """N single-agent environments of the reference, stepped by one kernel launch.

    Field tensors (views, no copies): `distances` [N, R] = sensor_hits_[r].norm(), `rel_x/rel_y` [N, R] =
    sensor_hits_, `hit_x/hit_y`, `pos_x, pos_y, rot, speed, acceleration, throttle, steering` [N] f32,
    `crashed, timed_out, mode` [N] u8.  `done` is `crashed` viewed as bool (Agent::crashed_ covers wall hits and
    standstill timeouts).  With `reward="step"` (+1 per step, RLRacers/PPO/ppo_sim.cpp:77-80)
    or `reward="progress"` (centre-line index progress, CovarianceMatrixAdaptationEvolution/main_eigen.cpp:147-158)
    also `reward, fitness, episode_return` [N] f32, `episode_steps` [N] u32, `track_idx` [N] i32, updated by `step`.
    """

    FIELDS = {"pos_x": capi.F_POS_X, "pos_y": capi.F_POS_Y, "rot": capi.F_ROT, "speed": capi.F_SPEED,
              "acceleration": capi.F_ACC, "throttle": capi.F_THROTTLE, "steering": capi.F_STEER, "mode": capi.F_MODE,
              "crashed": capi.F_CRASHED, "timed_out": capi.F_TIMED_OUT, "hit_x": capi.F_HIT_X, "hit_y": capi.F_HIT_Y,
              "rel_x": capi.F_REL_X, "rel_y": capi.F_REL_Y, "distances": capi.F_DIST,
              # DisplacementStats (the standstill bookkeeping, Environment.h:17-27)
              "disp_ctr": capi.F_DISP_CTR, "disp_x": capi.F_DISP_X, "disp_y": capi.F_DISP_Y, "disp_timed_out": capi.F_DISP_TO}
    TRACKER_FIELDS = {"reward": capi.F_REWARD, "fitness": capi.F_FITNESS, "track_idx": capi.F_TRACK_IDX,
                      "episode_steps": capi.F_EPISODE_STEPS, "episode_return": capi.F_EPISODE_RETURN,
                      # the tracker's memory of crashed_ at its last update: part of the state capture() saves and restores
                      "prev_crashed": capi.F_PREV_CRASHED}
    SENSOR_RANGE = 200.0  # Agent::kSensorRange (Environment/Agent.h:10)

    def __init__(self, race_track_path, num_envs, num_rays=15, ray_angles_deg=None, device=0,
                 movement_mode=capi.MODE_VELOCITY, auto_reset=True, pick_random_point=True, randomize_lane=False,
                 randomize_heading=False, seed=0, agent_base=0, reward=None, draw_rays=False, hidden_window=True):
        # draw_rays / hidden_window: accepted for signature compatibility; there is no window (rendering is out of scope)
        del draw_rays, hidden_window
        if not torch.cuda.is_available():
            raise capi.OkenvError(-3, "VectorEnvironment needs a GPU; there is no CPU path")
        self.device = torch.device("cuda", int(device))
        self.track = race_track_path if isinstance(race_track_path, Track) else Track(race_track_path)
        rays = default_ray_fan(num_rays) if ray_angles_deg is None else ray_angles_deg
        self.env = BatchedEnvironment.from_track(self.track, num_envs, ray_angles_deg=rays, device=int(device))
        self.num_envs, self.num_rays = self.env.N, self.env.R
        self.seed, self.agent_base = int(seed), int(agent_base)
        self.reset_flags = ((capi.RESET_RANDOM_POINT if pick_random_point else 0) |
                            (capi.RESET_RANDOM_LANE if randomize_lane else 0) |
                            (capi.RESET_RANDOM_HEADING if randomize_heading else 0))
        with torch.cuda.device(self.device):
            self.env.set_stream(torch.cuda.current_stream().cuda_stream)
            for name, f in self.FIELDS.items():
                setattr(self, name, self._view(f))
            self.done = self.crashed.view(torch.bool)
            self.mode.fill_(int(movement_mode))
            # rollout bookkeeping on the device: reward / fitness / episode length (okenv_tracker_*)
            self.reward_kind = {None: None, "step": capi.REWARD_STEP, "progress": capi.REWARD_PROGRESS}[reward]
            if self.reward_kind is not None:
                self.env.tracker_create(self.reward_kind)
                for name, f in self.TRACKER_FIELDS.items():
                    setattr(self, name, self._view(f))
        self.auto_reset = bool(auto_reset)
        self.env.set_auto_reset(self.auto_reset, self.reset_flags, self.seed, self.agent_base)
        # Pybind/bindings.cpp:27-33: the agent starts on a (random) centre-line point with the track heading
        self.env.reset_random(None, capi.RESET_RANDOM_POINT if pick_random_point else 0, self.seed, 0xFFFFFFFF,
                              self.agent_base)

    def _view(self, field):
        address, nbytes = self.env.field_device_ptr(field)
        dtype = torch.from_numpy(capi.FIELD_DTYPE[field](0).reshape(1)).dtype
        shape = (self.num_envs, self.num_rays) if field in capi.PER_RAY else (self.num_envs,)
        t = torch.as_tensor(_DeviceArray(address, shape, dtype, self.env), device=self.device)
        assert t.data_ptr() == address and t.numel() * t.element_size() == nbytes
        return t

    def use_stream(self, stream):
        """Enqueue the environment's kernels on `stream` (a torch.cuda.Stream) from now on."""
        self.env.set_stream(stream.cuda_stream)

    def _state_tensors(self):
        names = list(self.FIELDS) + (list(self.TRACKER_FIELDS) if self.reward_kind is not None else [])
        return {n: getattr(self, n) for n in names}

    def capture(self, body, warmup=3):
        """Capture `body()` -- typically policy forward + `self.step(actions)` -- into a HIP graph and return it
        (`graph.replay()` runs one iteration).  A loop of one Environment step per policy evaluation is launch-bound
        (a dozen small kernels per iteration); replaying it as a graph removes the per-launch host cost.  The step
        counter that seeds the auto-reset draws lives on the device, so replays keep advancing it.  `body` must not
        synchronise or touch the host.  The `warmup` eager iterations that precede the capture (library and allocator
        initialisation) run on a copy: the environment's state and step count are restored afterwards, so capturing
        has no side effect on the simulation."""
        torch.cuda.synchronize(self.device)
        saved = {n: t.clone() for n, t in self._state_tensors().items()}
        count = self.env.step_count
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream(self.device))
        with torch.cuda.stream(side):
            self.use_stream(side)
            for _ in range(warmup):
                body()
        torch.cuda.current_stream(self.device).wait_stream(side)
        torch.cuda.synchronize(self.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            body()
        self.use_stream(torch.cuda.current_stream(self.device))
        for n, t in self._state_tensors().items():
            t.copy_(saved[n])
        self.env.step_count = count
        torch.cuda.synchronize(self.device)
        return graph

    # ---- the reference binding's two methods, batched -------------------------------------------------------------
    def set_action(self, throttle_delta, steering_delta):
        """Agent::current_action_ of every agent; scalars or [N] tensors (Pybind/bindings.cpp:36-40)."""
        if torch.is_tensor(throttle_delta):
            self.throttle.copy_(throttle_delta, non_blocking=True)
        else:
            self.throttle.fill_(float(throttle_delta))
        if torch.is_tensor(steering_delta):
            self.steering.copy_(steering_delta, non_blocking=True)
        else:
            self.steering.fill_(float(steering_delta))

    def step(self, actions=None, n_steps=1):
        """Environment::step() for all agents.  `actions`: optional [N, 2] (throttle, steering) tensor.
        Returns (distances [N, R], done [N]) -- views that the next step overwrites."""
        if actions is not None:
            self.set_action(actions[:, 0], actions[:, 1])
        self.env.step(n_steps)
        if self.reward_kind is not None:
            self.env.tracker_update()
        return self.distances, self.done

    # ---- conveniences for learners ----------------------------------------------------------------------------------
    def observation(self):
        """sensor_hits_[i].norm() / kSensorRange, the network input of the reference's learners
        (RLRacers/PPO/PPOAgent.hpp:66-74, CovarianceMatrixAdaptationEvolution/main_eigen.cpp:45-56)."""
        return self.distances / self.SENSOR_RANGE

    def reset(self, mask=None, epoch=None):
        """Environment::resetAgent for all agents (or those in the bool/index tensor `mask`) with this environment's
        flags, then one step with the zeroed action for the initial observation (RLRacers/PPO/ppo_sim.cpp:53-60)."""
        epoch = self.env.step_count if epoch is None else int(epoch)
        if mask is None:
            self.env.reset_random(None, self.reset_flags, self.seed, epoch, self.agent_base)
        else:
            idx = mask.nonzero().flatten() if mask.dtype == torch.bool else mask
            idx = idx.to(device=self.device, dtype=torch.int32).contiguous()
            if idx.numel():
                self.env.reset_random(idx, self.reset_flags, self.seed, epoch, self.agent_base)
        self.env.step(1)
        if self.reward_kind is not None:
            if mask is None:
                self.env.tracker_begin()
            else:
                self.env.tracker_update()  # re-placed agents restart their episode, the others take a normal step
        return self.distances, self.done

    def nearest_track_idx(self):
        """RaceTrack::findNearestTrackIndexBruteForce for every agent, as a device tensor."""
        out = torch.empty(self.num_envs, dtype=torch.int32, device=self.device)
        capi.check(self.env._L.okenv_nearest_track_idx(self.env._h, None, None, 0, capi.ptr(out)), self.env._h)
        return out

    def synchronize(self):
        self.env.sync()

    def close(self):
        self.env.close()
