"""How much of a C3 generation / C5 episode is spent on agents that have already crashed: alive counts after every launch of
`spl` steps.  Prints per generation: steps, sum of alive agent-steps (upper bound from launch-boundary counts), wall time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openkitchen_amd as ok
from openkitchen_amd.evolution import EvolutionaryRacer
from openkitchen_amd.qlearning import QLearningRacers

spl = int(sys.argv[1]) if len(sys.argv) > 1 else 20

def c3(gens=6):
    t = ok.Track("Monza")
    env = ok.BatchedEnvironment.from_track(t, 8192, 32)
    ga = EvolutionaryRacer(env, t, hidden=30, seed=1234, agent_base=0, max_steps=4000, steps_per_launch=spl)
    for g in range(gens):
        env.reset_all(*ga.start); env.step(1)
        steps, curve = 1, []
        env.sync(); t0 = time.perf_counter()
        while steps < 4000:
            env.rollout_policy(spl); steps += spl
            a = env.alive_count(); curve.append(a)
            if a == 0: break
        env.sync(); dt = time.perf_counter() - t0
        live = sum(curve) * spl
        print("c3 gen %d: steps %d wall %.1f ms live<=%.3e nominal %.3e live_frac %.3f; alive@[1,2,5,10,20,50]xspl=%s" %
              (g, steps, dt * 1e3, live, 8192 * steps, live / (8192 * steps), [curve[i - 1] if i <= len(curve) else 0 for i in (1, 2, 5, 10, 20, 50)]))
        local = env.ga_scores(); env.ga_select_mate(1234, g, 0)
        ga.generation += 1
    env.close()

def c5(eps=6):
    t = ok.Track("Silverstone")
    env = ok.BatchedEnvironment.from_track(t, 16384, 16)
    ql = QLearningRacers(env, t, seed=1234, agent_base=0, steps_per_launch=spl)
    for e in range(eps):
        env.q_begin_episode(ql.reset_idx)
        steps, curve = 0, []
        env.sync(); t0 = time.perf_counter()
        while steps < 4000:
            env.rollout_q(spl, float(ql.epsilon), ql.seed, 0, ql.steps_total + steps); steps += spl
            a = env.alive_count(); curve.append(a)
            if a == 0: break
        env.sync(); dt = time.perf_counter() - t0
        ql.steps_total += steps
        ql.epsilon = ql.epsilon - np.float32(0.05) if ql.epsilon > np.float32(0.05) else np.float32(0.0)
        ql.reset_idx = int(ql._rng.integers(0, t.P))
        live = sum(curve) * spl
        print("c5 ep %d: steps %d wall %.1f ms live<=%.3e nominal %.3e live_frac %.3f; alive@[1,2,5,10,20,50]xspl=%s" %
              (e, steps, dt * 1e3, live, 16384 * steps, live / (16384 * steps), [curve[i - 1] if i <= len(curve) else 0 for i in (1, 2, 5, 10, 20, 50)]))
    env.close()

c3(); c5()
