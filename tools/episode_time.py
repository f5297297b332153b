"""Wall time of C3 generations / C5 episodes through the product's drivers (episodes: work follows the live agents)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openkitchen_amd as ok
from openkitchen_amd.evolution import EvolutionaryRacer
from openkitchen_amd.qlearning import QLearningRacers

spl = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t = ok.Track("Monza")
env = ok.BatchedEnvironment.from_track(t, 8192, 32)
ga = EvolutionaryRacer(env, t, hidden=30, seed=1234, agent_base=0, max_steps=4000, steps_per_launch=spl)
for g in range(6):
    env.sync(); t0 = time.perf_counter()
    steps = ga.rollout()
    env.sync(); dt = time.perf_counter() - t0
    print("c3 spl %d gen %d: steps %d wall %.2f ms nominal %.3e/s live %.3e/s (live frac %.3f)" %
          (spl, g, steps, dt * 1e3, 8192 * steps / dt, ga.live_agent_steps / dt, ga.live_agent_steps / (8192.0 * steps)))
    env.ga_scores(); env.ga_select_mate(1234, g, 0); ga.generation += 1
env.close()
t = ok.Track("Silverstone")
env = ok.BatchedEnvironment.from_track(t, 16384, 16)
ql = QLearningRacers(env, t, seed=1234, agent_base=0, steps_per_launch=spl)
for e in range(6):
    r = ql.run_episode()
    print("c5 spl %d ep %d: steps %d wall %.2f ms nominal %.3e/s live %.3e/s (live frac %.3f)" %
          (spl, e, r["steps"], r["wall_s"] * 1e3, 16384 * r["steps"] / r["wall_s"], r["live_agent_steps"] / r["wall_s"],
           r["live_agent_steps"] / (16384.0 * r["steps"])))
env.close()
