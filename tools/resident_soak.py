"""Soak test of the resident packed-step kernel (run on the GPU box): the same random steps through a handle that keeps the
kernel resident (OKENV_RESIDENT=1) and through one that launches per step (OKENV_RESIDENT=0); every output must agree bit
for bit.  usage: python tools/resident_soak.py [steps] [agents] [rays]"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import openkitchen_amd as ok  # noqa: E402
from test_gpu_packed_step import REC, WITH_STATS  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
N = int(sys.argv[2]) if len(sys.argv) > 2 else 15
R = int(sys.argv[3]) if len(sys.argv) > 3 else 5
t = ok.Track("Silverstone")
fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32) if R == 5 else ok.default_ray_fan(R)
envs = []
for mode in ("1", "0"):
    os.environ["OKENV_RESIDENT"] = mode
    envs.append(ok.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan))
rng = np.random.default_rng(7)
idx = rng.integers(0, t.P, N)
rec = np.zeros(N, dtype=REC)
rec["pos_x"], rec["pos_y"], rec["rot"] = t.x[idx], t.y[idx], t.heading[idx]
recs = [rec.copy(), rec.copy()]
hits = [np.zeros((N, R, 2), dtype=np.float32) for _ in range(2)]
t0 = time.time()
block = 1000
for s0 in range(0, steps, block):
    thr = rng.uniform(30, 100, (block, N)).astype(np.float32)
    steer = rng.uniform(-5, 5, (block, N)).astype(np.float32)
    where = rng.integers(0, t.P, (block, N))
    for i in range(block):
        crashed = np.flatnonzero(recs[0]["crashed"])
        for e in range(2):
            r = recs[e]
            if crashed.size:
                j = where[i, crashed]
                r["pos_x"][crashed], r["pos_y"][crashed], r["rot"][crashed] = t.x[j], t.y[j], t.heading[j]
                r["speed"][crashed] = r["acc"][crashed] = 0
                r["crashed"][crashed] = r["timed_out"][crashed] = 0
            r["throttle"], r["steer"] = thr[i], steer[i]
            ok.capi.check(envs[e]._L.okenv_step_packed(envs[e]._h, r.ctypes.data_as(C.c_void_p), r.ctypes.data_as(C.c_void_p),
                                                      hits[e].ctypes.data_as(C.c_void_p), WITH_STATS), envs[e]._h)
        if recs[0].tobytes() != recs[1].tobytes() or hits[0].tobytes() != hits[1].tobytes():
            print("MISMATCH at step", s0 + i)
            sys.exit(1)
    if (s0 // block) % 20 == 0:
        info = envs[0].info()
        print("step %d: %.0f s, resident steps %d, fallbacks %d" % (s0 + block, time.time() - t0, info["packed_resident_steps"], info["packed_fallbacks"]), flush=True)
info = envs[0].info()
print("soak ok: %d steps x %d agents x %d rays, resident steps %d, fallbacks %d, %.0f s" % (steps, N, R, info["packed_resident_steps"], info["packed_fallbacks"], time.time() - t0))
