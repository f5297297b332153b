"""Quick C2 timing probe (development aid; bench.py is the judged harness)."""
import sys
import time


sys.path.insert(0, ".")
import openkitchen_amd as ok

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
R = int(sys.argv[2]) if len(sys.argv) > 2 else 64
track = sys.argv[3] if len(sys.argv) > 3 else "Silverstone"
cell = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
flags = int(sys.argv[5]) if len(sys.argv) > 5 else 0
t = ok.Track(track)
env = ok.BatchedEnvironment.from_track(t, N, R, grid_cell=cell, flags=flags)
print(env.info())
env.init_bench_state(0, 0)
env.rollout_random(100, 1234, 0, 0)
env.sync()
for spl in (1, 10, 100):
    steps = 400 if flags != 2 else 20
    env.set_timing(True)
    t0 = time.perf_counter()
    for c in range(steps // spl):
        env.rollout_random(spl, 1234, 0, 100 + c * spl)
    env.sync()
    dt = time.perf_counter() - t0
    ms, n = env.get_timing()
    env.set_timing(False)
    print("steps/launch %4d: wall %.3f s -> %.3e agent-steps/s ; kernel time %.3f ms over %d launches -> %.3e agent-steps/s (kernel only), %.1f us/step"
          % (spl, dt, N * steps / dt, ms, n, N * steps / (ms * 1e-3), ms * 1e3 / steps))
s = env.snapshot()
print("crashed frac", s["crashed"].mean(), "mean dist", s["dist"].mean())
