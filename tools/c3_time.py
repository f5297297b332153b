import os, sys, time
sys.path.insert(0, "/root/repo")
import openkitchen_amd.buildlib as bl
v = os.environ.get("OKENV_VARIANT")
if v:
    bl.LIB_PATH = "/root/repo/tools/_build/libokenv_%s.so" % v
    bl.needs_build = lambda: False
import numpy as np
import openkitchen_amd as ok
t = ok.Track("Monza")
env = ok.BatchedEnvironment.from_track(t, 8192, 32)
env.set(ok.capi.F_MODE, np.ones(8192, dtype=np.uint8))
env.policy_mlp_create(30, 1234, 0)
env.reset_all(float(t.x[3]), float(t.y[3]), float(t.heading[0]))
env.step(1)
env.rollout_policy(100); env.sync()
env.set_timing(True)
for _ in range(3): env.rollout_policy(100)
ms, n = env.get_timing()
print(v or "main", "C3 fused MLP rollout: %.2f us/step, alive %d" % (ms * 1e3 / 300, env.alive_count()))
