"""Host-side driver of the EvolutionaryRacer generation loop (reference EvolutionaryRacer/genetic_learner_sim.cpp:47-96)
over the device-resident population: every step of the loop body runs on the GPU (policy + Environment::step fused in
one kernel, scores, selection, mating); the host only decides when a rollout is over.

With torch.distributed initialised, every rank runs an independent island population on its own GPU (BASELINE config 4)
and the per-generation fitness vectors are all-gathered over RCCL for the colony statistics; selection and mating stay
local to each island, as SURVEY.md section 8e prescribes.
"""
import time

import numpy as np
import torch

from . import sharding
from . import _capi as capi


class EvolutionaryRacer:
    def __init__(self, env, track, hidden=30, seed=1234, agent_base=0, max_steps=4000, steps_per_launch=100, device=None):
        self.env, self.track = env, track
        self.seed, self.agent_base = int(seed), int(agent_base)
        self.max_steps, self.spl = int(max_steps), int(steps_per_launch)
        self.device = device
        self.generation = 0
        env.set(capi.F_MODE, np.full(env.N, capi.MODE_ACCELERATION, dtype=np.uint8))  # GeneticAgent.hpp:28,34
        env.policy_mlp_create(hidden, seed, agent_base)
        self.start = (float(track.x[3]), float(track.y[3]), float(track.heading[0]))  # genetic_learner_sim.cpp:34-36
        self.history = []
        self._fitness = None  # device tensor the scores are written into (GPU runs)
        self.live_agent_steps = 0

    def rollout(self):
        """Reset everybody to the start line and drive until every agent has crashed or timed out
        (genetic_learner_sim.cpp:75-95).  Runs as an episode (include/okenv.h): launches step only the agents that can still
        change, and the loop's own step count T -- it ends with the step in which the last agent crashes -- comes back from
        okenv_episode_end whatever the launches' lengths.  Returns the Environment steps taken (initial observation + T)."""
        e = self.env
        e.reset_all(*self.start)
        e.step(1)  # initial observation (genetic_learner_sim.cpp:75)
        e.episode_begin()
        tail = e.episode_tail_limit()
        steps, listed, alive = 0, e.N, e.N
        budget = self.max_steps - 1
        while steps < budget:
            # once the list is short enough for one agent per workgroup, a workgroup leaves when its agent is done: ask for all the
            # steps that are left -- the launch ends with the last crash, no more launch boundaries
            n = budget - steps if listed <= tail else min(self.spl, budget - steps)
            e.rollout_policy(n)
            steps += n
            alive, listed = e.episode_compact()
            if alive == 0:
                break
        loop_steps, self.live_agent_steps = e.episode_end()
        self.live_agent_steps += e.N  # everybody takes the initial step
        self.alive_at_end = int(alive)  # > 0 only when the step cap ended the loop
        return 1 + loop_steps

    def run_generation(self):
        t0 = time.perf_counter()
        steps = self.rollout()
        t1 = time.perf_counter()
        on_gpu = self.device is not None and str(self.device).startswith("cuda")
        if on_gpu:
            # assignScores into a device tensor, all-gather from there (RCCL reads device memory): no host hop on the data path
            if self._fitness is None:
                self._fitness = torch.empty(self.env.N, dtype=torch.float32, device=self.device)
            local = self.env.ga_scores_into(self._fitness)
            self.env.sync()  # the copy ran on the environment's stream; the collective runs on torch's
        else:
            local = torch.as_tensor(self.env.ga_scores(), dtype=torch.float32)
        tg = time.perf_counter()
        colony = sharding.all_gather_fitness(local)  # [world, N]: global colony statistics (showColonyScore)
        if on_gpu:
            torch.cuda.current_stream(self.device).synchronize()
        tg = time.perf_counter() - tg
        parents = self.env.ga_select_mate(self.seed, self.generation, self.agent_base)  # chooseAndMateAgents
        self.env.sync()
        t2 = time.perf_counter()
        stats = torch.stack([local.max(), local.mean(), colony.max(), colony.mean()]).tolist()  # four scalars leave the device
        # (outside the timed parts) who ended the generation where: agents outside the raycast grid's box tunnelled through both
        # boundaries and cannot crash any more -- if one of them is alive, it is what ran the loop into the step cap
        off_alive, off_all = self.env.off_grid_count() if hasattr(self.env, "off_grid_count") else (0, 0)
        rec = {"generation": self.generation, "steps": steps, "live_agent_steps": int(self.live_agent_steps), "rollout_s": t1 - t0,
               "select_mate_s": t2 - t1, "all_gather_s": tg, "alive_at_end": self.alive_at_end, "off_grid_alive": off_alive,
               "off_grid_agents": off_all,
               "island_best": stats[0], "island_mean": stats[1], "colony_best": stats[2], "colony_mean": stats[3],
               "parents": [int(v) for v in parents]}
        self.history.append(rec)
        self.generation += 1
        return rec
