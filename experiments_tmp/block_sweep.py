import os, sys, time
import numpy as np
sys.path.insert(0, ".")
import openkitchen_amd as ok
t = ok.Track("Silverstone")
for N, R in ((4096, 5), (1024, 15), (64, 16), (512, 64), (16384, 16)):
    for bt in (0, 128, 256, 512, 1024):
        if bt: os.environ["OKENV_BLOCK_THREADS"] = str(bt)
        else: os.environ.pop("OKENV_BLOCK_THREADS", None)
        env = ok.BatchedEnvironment.from_track(t, N, num_rays=R)
        env.init_bench_state(0, 0)
        info = env.info()
        env.rollout_random(50, 1, 0, 0); env.sync()
        env.set_timing(True)
        for s in range(200):
            env.rollout_random(1, 1, 0, 50 + s)
        ms, n = env.get_timing()
        print("N %5d R %2d bt %4d -> block %4d x %3d blocks: %.1f us/step (kernel only)" % (N, R, bt, info["block_threads"], info["grid_blocks"], ms / n * 1e3), flush=True)
        env.close()
