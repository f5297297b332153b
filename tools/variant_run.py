"""Development aid for tools/pmc_variants.sh: runs a few 100-step C2 launches of the library variant named by OKENV_VARIANT."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import openkitchen_amd.buildlib as bl
bl.LIB_PATH = os.path.join(ROOT, "tools", "_build", "libokenv_%s.so" % os.environ["OKENV_VARIANT"])
bl.needs_build = lambda: False
import openkitchen_amd as ok
t = ok.Track("Silverstone")
env = ok.BatchedEnvironment.from_track(t, 4096, 64)
env.init_bench_state(0, 0)
for c in range(4):
    env.rollout_random(100, 1234, 0, c * 100)
env.sync()
env.close()
