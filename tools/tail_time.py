"""Development aid: per-step kernel time of a short episode list, tail kernel (one agent per workgroup) against the cooperative
kernel (OKENV_TAIL_MAX_AGENTS=0), for the C3 (MLP, 32 rays, Monza) and C5 (Q-learning, 16 rays, Silverstone) shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
if os.environ.get("OKENV_STAMPS_LIB") or os.environ.get("OKENV_LIB"):  # a variant from tools/build_variant.sh (the former: built with -DOKENV_STAMPS)
    import openkitchen_amd.buildlib as bl
    bl.LIB_PATH = os.path.abspath(os.environ.get("OKENV_STAMPS_LIB") or os.environ["OKENV_LIB"]); bl.needs_build = lambda: False
import openkitchen_amd as ok

def run(cfg, N, tail):
    os.environ["OKENV_TAIL_MAX_AGENTS"] = "100000" if tail else "0"
    if cfg == "c3":
        t = ok.Track("Monza")
        env = ok.BatchedEnvironment.from_track(t, 8192, 32)   # created like the full population: 32-lane groups
        env.set(ok.capi.F_MODE, np.ones(8192, dtype=np.uint8))
        env.policy_mlp_create(30, 1234, 0)
        env.reset_all(float(t.x[3]), float(t.y[3]), float(t.heading[0])); env.step(1)
        crashed = np.ones(8192, dtype=np.uint8); crashed[:N] = 0   # N agents alive, the rest out of the way
        env.set(ok.capi.F_CRASHED, crashed)
        env.episode_begin(); env.rollout_policy(1); alive, listed = env.episode_compact()
        env.rollout_policy(20); env.sync(); env.set_timing(True)
        for _ in range(3): env.rollout_policy(40)
    else:
        t = ok.Track("Silverstone")
        env = ok.BatchedEnvironment.from_track(t, 16384, 16)
        env.q_create(); env.q_begin_episode(3)
        crashed = np.ones(16384, dtype=np.uint8); crashed[:N] = 0
        env.set(ok.capi.F_CRASHED, crashed)
        env.episode_begin(); env.rollout_q(1, 0.9, 1234, 0, 0); alive, listed = env.episode_compact()
        env.rollout_q(20, 0.9, 1234, 0, 1); env.sync(); env.set_timing(True)
        for i in range(3): env.rollout_q(40, 0.9, 1234, 0, 21 + 40 * i)
    ms, n = env.get_timing()
    a2, l2 = env.episode_compact()
    print("%s listed %d (alive at the end %d) %s: %.2f us/step" % (cfg, listed, a2, "tail kernel" if tail else "coop kernel", ms * 1e3 / 120), flush=True)
    if os.environ.get("OKENV_STAMPS_LIB") and tail:
        L = ok.capi.load(); L.okenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        out = np.zeros((8192, 24), dtype=np.uint64)
        n = L.okenv_debug_stamps(env._h, out.ctypes.data_as(C.c_void_p), 8192)
        o = out[:n].astype(np.float64) / 40.0
        print("    stamps (cycles per wave-step, mean over %d waves): policy %.0f pre-step %.0f walk %.0f epilogue %.0f barrier %.0f crash+Q %.0f ; sum %.0f"
              % (n, o[:, 0].mean(), o[:, 1].mean(), o[:, 3].mean(), o[:, 5].mean(), o[:, 6].mean(), o[:, 7].mean(), o[:, [0, 1, 3, 5, 6, 7]].sum(axis=1).mean()))
        print("    inside the walk: set-up %.0f cell entry %.0f point loop %.0f exact loop %.0f leaving %.0f" % tuple(o[:, 8 + i].mean() for i in range(5)))
    env.close()

sizes = [int(v) for v in sys.argv[1].split(",")] if len(sys.argv) > 1 else [8, 64, 256, 512, 768, 1024]
for cfg in ("c3", "c5"):
    for N in sizes:
        for tail in (False, True):
            run(cfg, N, tail)
