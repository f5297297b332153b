"""Diagnostic (OKENV_STAMPS build): where a step of the fused policy kernels goes, per phase, for a full and for a nearly empty
machine (the tail of a C3 generation / C5 episode is a few agents whose steps are one wave's dependent chain).

  tools/build_variant.sh stamps -DOKENV_STAMPS && python tools/policy_stamps.py"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
import openkitchen_amd.buildlib as bl
bl.LIB_PATH = os.path.abspath(os.environ.get("OKENV_STAMPS_LIB", "tools/_build/libokenv_stamps.so"))
bl.needs_build = lambda: False
import openkitchen_amd as ok
from openkitchen_amd import capi
L = capi.load(build_if_missing=False)
L.okenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
NAMES = ["policy", "pre-step", "phase1", "phase2", "epilogue"]
IDX = [0, 1, 3, 5, 6]
WN = ["setup", "cell-entry", "points", "exact", "cell-leave"]

def report(tag, env, spl, ms):
    out = np.zeros((65536, 24), dtype=np.uint64)
    n = L.okenv_debug_stamps(env._h, out.ctypes.data_as(C.c_void_p), 65536)
    out = out[:n]
    out = out[out[:, 4] > 0]  # waves that ran the loop
    cyc = out[:, IDX].astype(np.float64) / spl
    life = (out[:, 4] - out[:, 2]).astype(np.float64) * 10e-3
    print("%s: %.2f us/step (events), %d waves; wave lifetime/step mean %.2f max %.2f us; cycles per wave-step: %s ; total %.0f" %
          (tag, ms * 1e3 / spl, len(out), life.mean() / spl, life.max() / spl,
           ", ".join("%s %.0f" % (NAMES[i], cyc[:, i].mean()) for i in range(5)), cyc.sum(axis=1).mean()))
    wp = out[:, 8:18].astype(np.float64) / spl
    for ph, off in (("phase1", 0), ("phase2", 5)):
        print("    %s walk internals (busiest lane): %s" % (ph, ", ".join("%s %.0f" % (WN[i], wp[:, off + i].mean()) for i in range(5))))

def c3(N, spl=50, lanes=None):
    if lanes:
        os.environ["OKENV_LANES_PER_AGENT"] = str(lanes)
    else:
        os.environ.pop("OKENV_LANES_PER_AGENT", None)
    t = ok.Track("Monza")
    env = ok.BatchedEnvironment.from_track(t, N, 32)
    env.set(ok.capi.F_MODE, np.ones(N, dtype=np.uint8))
    env.policy_mlp_create(30, 1234, 0)
    env.reset_all(float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    env.step(1); env.rollout_policy(20); env.sync()
    env.set_timing(True)
    env.rollout_policy(spl)
    ms, n = env.get_timing()
    report("c3 N=%d G=%d alive %d" % (N, env.info()["lanes_per_agent"], env.alive_count()), env, spl, ms)
    env.close()

def c5(N, spl=50, lanes=None):
    if lanes:
        os.environ["OKENV_LANES_PER_AGENT"] = str(lanes)
    else:
        os.environ.pop("OKENV_LANES_PER_AGENT", None)
    t = ok.Track("Silverstone")
    env = ok.BatchedEnvironment.from_track(t, N, 16)
    env.q_create()
    env.q_begin_episode(3)
    env.rollout_q(20, 0.9, 1234, 0, 0); env.sync()
    env.set_timing(True)
    env.rollout_q(spl, 0.9, 1234, 0, 20)
    ms, n = env.get_timing()
    report("c5 N=%d G=%d alive %d" % (N, env.info()["lanes_per_agent"], env.alive_count()), env, spl, ms)
    env.close()

def ctrl(N, spl=50, lanes=None):
    """The CMA-ES racers' fused rollout: five rays, controller 5-16-8-2, index-progress fitness."""
    if lanes:
        os.environ["OKENV_LANES_PER_AGENT"] = str(lanes)
    else:
        os.environ.pop("OKENV_LANES_PER_AGENT", None)
    t = ok.Track("Silverstone")
    fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32)
    env = ok.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    rng = np.random.default_rng(1)
    idx = rng.integers(0, t.P, N)
    env.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    env.step(1)
    n_params = env.controller_create(16)
    env.controller_set_params(rng.normal(0, 0.5, (N, n_params)).astype(np.float32))
    env.tracker_create(1); env.tracker_begin()
    env.rollout_controller(20); env.sync()
    env.set_timing(True)
    env.rollout_controller(spl)
    ms, n = env.get_timing()
    report("ctrl N=%d G=%d alive %d" % (N, env.info()["lanes_per_agent"], env.alive_count()), env, spl, ms)
    env.close()

def setp1(v):
    if v is None:
        os.environ.pop("OKENV_PHASE1_RANGE", None)
    else:
        os.environ["OKENV_PHASE1_RANGE"] = str(v)

if len(sys.argv) > 1:  # e.g. c3:16:64:48 c5:256:64:0  (config:agents:lanes per agent:phase-1 range, '-' = default)
    for spec in sys.argv[1:]:
        cfg, N, lanes, p1 = spec.split(":")
        setp1(None if p1 == "-" else p1)
        print("[%s]" % spec, end=" ")
        {"c3": c3, "c5": c5, "ctrl": ctrl}[cfg](int(N), lanes=None if lanes == "-" else int(lanes))
else:
    for N, lanes in ((8192, None), (256, 32), (256, 64), (16, 32), (16, 64)):
        c3(N, lanes=lanes)
    for N, lanes in ((16384, None), (256, 16), (256, 64), (16, 16), (16, 64)):
        c5(N, lanes=lanes)
