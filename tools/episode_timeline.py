"""Per launch of a C3 generation / C5 episode: agents listed, steps asked, wall time (each launch synchronised)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openkitchen_amd as ok

spl = int(sys.argv[1]) if len(sys.argv) > 1 else 100


def episode(e, launch, tag, budget=3999):
    e.episode_begin()
    tail = e.episode_tail_limit()
    steps, listed, alive = 0, e.N, e.N
    while steps < budget:
        n = budget - steps if listed <= tail else min(spl, budget - steps)
        e.sync(); t0 = time.perf_counter()
        launch(n, steps)
        e.sync(); dt = time.perf_counter() - t0
        steps += n
        a2, l2 = e.episode_compact()
        print("%s launch: listed %5d alive-before %5d asked %4d  %8.1f us%s -> alive %d listed %d" %
              (tag, listed, alive, n, dt * 1e6, "" if n > spl else "  (%.2f us/step)" % (dt * 1e6 / n), a2, l2))
        alive, listed = a2, l2
        if alive == 0:
            break
    T, live = e.episode_end()
    print("%s: T %d live %d tail limit %d" % (tag, T, live, tail))


t = ok.Track("Monza")
env = ok.BatchedEnvironment.from_track(t, 8192, 32)
env.set(ok.capi.F_MODE, np.ones(8192, dtype=np.uint8))
env.policy_mlp_create(30, 1234, 0)
for g in range(4):
    env.reset_all(float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    env.step(1)
    episode(env, lambda n, s: env.rollout_policy(n), "c3 gen %d" % g)
    env.ga_scores(); env.ga_select_mate(1234, g, 0)
env.close()
t = ok.Track("Silverstone")
env = ok.BatchedEnvironment.from_track(t, 16384, 16)
env.q_create()
tot = 0
for ep in range(2):
    env.q_begin_episode(3)
    episode(env, lambda n, s: env.rollout_q(n, 0.9 - 0.05 * ep, 1234, 0, tot + s), "c5 ep %d" % ep)
env.close()
