"""Development aid: kernel time (HIP events) and host wall time of single-step launches for small populations."""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import openkitchen_amd as ok
t = ok.Track("Austin")
for N, R in ((1, 5), (15, 5), (50, 15), (200, 5), (4096, 5)):
    fan = ok.default_ray_fan(R)
    env = ok.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    env.init_bench_state(0, 0)
    env.set_actions(np.full(N, 30, dtype=np.float32), np.zeros(N, dtype=np.float32))
    env.step(50); env.sync()
    env.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(300):
        env.step(1)
        env.sync()
    wall = (time.perf_counter() - t0) / 300 * 1e6
    ms, n = env.get_timing()
    print("N %5d R %2d: kernel %.2f us per single-step launch, host wall (launch + sync) %.2f us ; info %s" % (N, R, ms * 1e3 / n, wall, {k: env.info()[k] for k in ("block_threads", "grid_blocks", "lanes_per_agent")}))
    env.close()
