"""Prints what okenv_create chose for the BASELINE shapes: lane-group width, cell edge, LDS image sizes (combined, front + back)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import openkitchen_amd as ok

for name, track, N, R in (("C1", "Austin", 64, 16), ("C2", "Silverstone", 4096, 64), ("C3", "Monza", 8192, 32), ("C4 island", "Spa", 8192, 32),
                          ("C5", "Silverstone", 16384, 16), ("five rays", "Silverstone", 4096, 5), ("one agent", "Silverstone", 1, 5)):
    t = ok.Track(track)
    env = ok.BatchedEnvironment.from_track(t, N, R)
    i = env.info()
    print("%-10s %-11s %5d x %2d: %s" % (name, track, N, R, {k: i[k] for k in sorted(i) if k not in ("device_name",)}))
    print("           tail limit %d" % env.episode_tail_limit() if hasattr(env, "episode_tail_limit") else "")
    env.close()
