"""Diagnostic: per-launch fixed cost vs per-step cost of the step kernel for a few population shapes (run on the GPU box)."""
import sys
sys.path.insert(0, ".")
import openkitchen_amd as ok
t = ok.Track("Silverstone")
for N, R in ((4096, 5), (64, 16), (4096, 64)):
    env = ok.BatchedEnvironment.from_track(t, N, num_rays=R)
    env.init_bench_state(0, 0)
    env.rollout_random(50, 1, 0, 0); env.sync()
    for spl in (1, 2, 4, 100):
        env.set_timing(True)
        for s in range(40):
            env.rollout_random(spl, 1, 0, 50 + s * spl)
        ms, n = env.get_timing()
        print("N %5d R %2d steps/launch %3d: %.1f us/launch, %.1f us/step" % (N, R, spl, ms / n * 1e3, ms / n / spl * 1e3), flush=True)
    env.close()
