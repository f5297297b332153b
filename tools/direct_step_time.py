"""Development aid: kernel time per step of multi-step launches for the small five-ray populations the C++ facade serves (direct
dealing of ray pieces to lanes, no phase 1).  OKENV_LIB=<variant .so> to time a variant (tools/build_variant.sh)."""
import os, sys
sys.path.insert(0, ".")
import numpy as np
if os.environ.get("OKENV_LIB"):
    import openkitchen_amd.buildlib as bl
    bl.LIB_PATH = os.path.abspath(os.environ["OKENV_LIB"]); bl.needs_build = lambda: False
import openkitchen_amd as ok
t = ok.Track("Silverstone")
fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32)
for N in (1, 15, 50, 256):
    env = ok.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    env.init_bench_state(0, 0)
    env.rollout_random(200, 1234, 0, 0); env.sync()
    env.set_timing(True)
    for i in range(5):
        env.rollout_random(200, 1234, 0, 200 + 200 * i)
    ms, n = env.get_timing()
    print("N %4d x 5 rays: %.2f us per step (G %d)" % (N, ms * 1e3 / (200 * n), env.info()["lanes_per_agent"]), flush=True)
    env.close()
