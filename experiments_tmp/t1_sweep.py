import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import openkitchen_amd as ok
t = ok.Track("Silverstone")
for N, R in ((4096, 5), (1024, 15)):
  for t1 in (8, 16, 24, 32, 48, 64):
    os.environ["OKENV_PHASE1_RANGE"] = str(t1)
    env = ok.BatchedEnvironment.from_track(t, N, num_rays=R)
    env.init_bench_state(0, 0)
    env.rollout_random(50, 1, 0, 0); env.sync()
    env.set_timing(True)
    for s in range(20):
        env.rollout_random(50, 1, 0, 50 + s * 50)
    ms, n = env.get_timing()
    print("N %5d R %2d T1 %3d: %.1f us/step" % (N, R, t1, ms / n / 50 * 1e3), flush=True)
    env.close()
