"""Host-side driver of the tabular Q-learning loop (reference RLRacers/Q_Learning/q_racer_sim.cpp:123-216) over the
device-resident population: action choice, Environment::step, reward and table update run fused on the GPU; the host only
ends episodes, decays epsilon and picks the next reset point."""
import time

import numpy as np

from . import _capi as capi


class QLearningRacers:
    def __init__(self, env, track, seed=1234, agent_base=0, steps_per_launch=100, max_episode_steps=4000):
        self.env, self.track = env, track
        self.seed, self.agent_base = int(seed), int(agent_base)
        self.spl, self.max_steps = int(steps_per_launch), int(max_episode_steps)
        env.set(capi.F_MODE, np.full(env.N, capi.MODE_VELOCITY, dtype=np.uint8))
        env.q_create()
        self.epsilon = np.float32(0.9)        # QAgent.hpp:26
        self.reset_idx = 3                    # RaceTrack::kStartingIdx, q_racer_sim.cpp:114
        self.episode = 0
        self.steps_total = 0
        self._rng = np.random.default_rng(self.seed)  # stands in for raylib's GetRandomValue in pickResetPosition

    def run_episode(self):
        e = self.env
        t0 = time.perf_counter()
        e.q_begin_episode(self.reset_idx)
        e.episode_begin()  # launches step the agents that can still change; crashed agents' -200 updates are settled at the end
        tail = e.episode_tail_limit()
        steps, listed = 0, e.N
        while steps < self.max_steps:
            # (a short list is stepped one agent per workgroup, each leaving with its agent: one launch for all that is left)
            n = self.max_steps - steps if listed <= tail else min(self.spl, self.max_steps - steps)
            e.rollout_q(n, float(self.epsilon), self.seed, self.agent_base, self.steps_total + steps)
            steps += n
            alive, listed = e.episode_compact()
            if alive == 0:
                break
        steps, live = e.episode_end()  # the loop's own length: it ends with the step in which the last agent crashes
        self.steps_total += steps
        # q_racer_sim.cpp:192-210
        self.epsilon = self.epsilon - np.float32(0.05) if self.epsilon > np.float32(0.05) else np.float32(0.0)
        self.reset_idx = int(self._rng.integers(0, self.track.P))
        self.episode += 1
        return {"episode": self.episode, "steps": steps, "live_agent_steps": int(live), "wall_s": time.perf_counter() - t0,
                "epsilon_next": float(self.epsilon)}
