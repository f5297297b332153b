"""Upper bound for an "inner boundaries first" broad phase (development aid, not a product path): the C2 workload stepped against
ALL of the track's segments (as shipped) and against the two inner boundary polylines only (LI and RI runs + their closers:
TrackSegments order, Environment/TrackSegments.cu:11-39).  An agent between the inner boundaries can only ever hit those first, so
if the outer polylines (3 px further out, 50 % of the points in every cell) were looked at only when needed, a step would cost
at most what the second line shows.  Results of the second run are NOT the reference's for agents that tunnel."""
import sys
sys.path.insert(0, ".")
import numpy as np
import openkitchen_amd as ok

track = sys.argv[1] if len(sys.argv) > 1 else "Silverstone"
N, R = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (4096, 64)
t = ok.Track(track)
P = t.P
seg = t.segments.reshape(-1, 4)
run = P - 1
inner = np.concatenate([seg[0:run], seg[2 * run:3 * run], seg[4 * run:4 * run + 2]])  # LI, RI, closers LI, RI
for name, s in (("all %d segments" % seg.shape[0], seg), ("inner boundaries only (%d segments)" % inner.shape[0], inner)):
    env = ok.BatchedEnvironment(np.ascontiguousarray(s), N, ok.default_ray_fan(R), centerline=(t.x, t.y, t.heading))
    env.init_bench_state(0, 0)
    env.rollout_random(300, 1234, 0, 0)
    env.sync()
    env.set_timing(True)
    for c in range(10):
        env.rollout_random(100, 1234, 0, 300 + 100 * c)
    ms, n = env.get_timing()
    ws = env.work_stats()
    print("%-44s %.2f us/step; per ray: %.2f exact tests, %.2f cells, %.1f points; LDS image %d B, cell %.0f" %
          (name, ms * 1e3 / 1000, ws["tests"] / ws["rays"], ws["cells"] / ws["rays"], ws["points"] / ws["rays"], env.info()["lds_bytes"], env.info()["grid_cell"]), flush=True)
    env.close()
