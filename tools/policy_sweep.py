"""Development aid: kernel time per step of the fused policy kernels (C3: 8192 x 32 Monza MLP; C5: 16384 x 16 Silverstone Q) for
phase-1 ranges / cell sizes / populations given on the command line:  python tools/policy_sweep.py c3|c5 N T1[,T1...] [cell,...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

cfg, N = sys.argv[1], int(sys.argv[2])
t1s = sys.argv[3].split(",")
cells = [float(c) for c in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0.0]
lanes = sys.argv[5] if len(sys.argv) > 5 else None
import openkitchen_amd as ok
for cell in cells:
    for t1 in t1s:
        if t1 == "-":
            os.environ.pop("OKENV_PHASE1_RANGE", None)
        else:
            os.environ["OKENV_PHASE1_RANGE"] = t1
        if lanes:
            os.environ["OKENV_LANES_PER_AGENT"] = lanes
        if cfg == "c3":
            t = ok.Track("Monza")
            env = ok.BatchedEnvironment.from_track(t, N, 32, grid_cell=cell)
            env.set(ok.capi.F_MODE, np.ones(N, dtype=np.uint8))
            env.policy_mlp_create(30, 1234, 0)
            env.reset_all(float(t.x[3]), float(t.y[3]), float(t.heading[0]))
            env.step(1); env.rollout_policy(20); env.sync()
            env.set_timing(True)
            for _ in range(3): env.rollout_policy(60)
        else:
            t = ok.Track("Silverstone")
            env = ok.BatchedEnvironment.from_track(t, N, 16, grid_cell=cell)
            env.q_create(); env.q_begin_episode(3)
            env.rollout_q(20, 0.9, 1234, 0, 0); env.sync()
            env.set_timing(True)
            for i in range(3): env.rollout_q(60, 0.9, 1234, 0, 20 + 60 * i)
        ms, n = env.get_timing()
        print("%s N=%d G=%d cell=%s T1=%s: %.2f us/step, alive %d" % (cfg, N, env.info()["lanes_per_agent"], env.info()["grid_cell"], t1, ms * 1e3 / 180, env.alive_count()), flush=True)
        env.close()
