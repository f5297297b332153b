"""Who is still alive when an EvolutionaryRacer generation on Spa reaches its step cap?  (the crash test is
lidar-only and a step can be 1.6 px long, so an agent can tunnel through both boundary polylines, 3 px apart, and then drives
on outside the track with nothing in range.)  Prints, for the agents with crashed_ == false at the cap: where they are relative to
the track (distance to the nearest boundary point, nearest centre-line index), how many of their rays see anything, speed."""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
import openkitchen_amd as ok
from openkitchen_amd.evolution import EvolutionaryRacer

track_name = sys.argv[1] if len(sys.argv) > 1 else "Spa"
gens = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t = ok.Track(track_name)
N, R = 8192, 32
env = ok.BatchedEnvironment.from_track(t, N, R, device=0)
ga = EvolutionaryRacer(env, t, hidden=30, seed=1234, agent_base=0, max_steps=4000, steps_per_launch=100, device=torch.device("cuda", 0))
seg = t.segments.reshape(-1, 4)
lo = np.minimum(seg[:, :2], seg[:, 2:]).min(axis=0)
hi = np.maximum(seg[:, :2], seg[:, 2:]).max(axis=0)
for g in range(gens):
    rec = ga.run_generation()  # (scoring and mating change weights only: the population's state is the rollout's end state)
    steps = rec["steps"]
    s = env.snapshot()
    alive = np.flatnonzero(s["crashed"] == 0)
    print("generation %d: %d steps, %d alive at the end, %d timed out" % (g, steps, alive.size, int(s["timed_out"].sum())))
    if alive.size:
        x, y = s["pos_x"][alive], s["pos_y"][alive]
        bd, _ = t.queries(x, y)
        d = s["dist"].reshape(N, R)[alive]
        sees = (d < 200.0).sum(axis=1)
        outside_box = (x < lo[0]) | (x > hi[0]) | (y < lo[1]) | (y > hi[1])
        far = np.hypot(np.maximum(np.maximum(lo[0] - x, x - hi[0]), 0), np.maximum(np.maximum(lo[1] - y, y - hi[1]), 0))
        print("  outside the track's bounding box: %d of %d; rays that see a segment: min %d median %d max %d; agents seeing nothing: %d" %
              (int(outside_box.sum()), alive.size, sees.min(), int(np.median(sees)), sees.max(), int((sees == 0).sum())))
        print("  distance to the nearest boundary point: min %.1f median %.1f max %.1f px; beyond the box by up to %.0f px; speed min %.1f max %.1f" %
              (bd.min(), np.median(bd), bd.max(), far.max(), s["speed"][alive].min(), s["speed"][alive].max()))
        print("  positions (first 5):", [(round(float(a), 1), round(float(b), 1)) for a, b in zip(x[:5], y[:5])], "track box", lo, hi)
