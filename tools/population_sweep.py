"""Throughput of the step kernel against population size (run on the GPU box): kernel-only agent-steps/s for 100-step
launches of the C2 recipe on Silverstone, R = 64, 16 and 5 (the applications' five-ray fan)."""
import sys

sys.path.insert(0, ".")
import openkitchen_amd as ok

t = ok.Track("Silverstone")
print("%8s %4s %12s %14s %10s" % ("agents", "rays", "us/step", "agent-steps/s", "rays/s"))
import numpy as np
for R in (64, 16, 5):
    for N in (1, 16, 64, 256, 1024, 4096, 16384, 65536, 262144):
        fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32) if R == 5 else ok.default_ray_fan(R)
        env = ok.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
        env.init_bench_state(0, 0)
        env.rollout_random(100, 1234, 0, 0)
        env.sync()
        env.set_timing(True)
        reps = 3 if N <= 65536 else 1
        for c in range(reps):
            env.rollout_random(100, 1234, 0, 100 + c * 100)
        ms, n = env.get_timing()
        us = ms * 1e3 / (100 * reps)
        print("%8d %4d %12.2f %14.3e %10.3e" % (N, R, us, N / (us * 1e-6), N * R / (us * 1e-6)), flush=True)
        env.close()
