"""Diagnostic: in-kernel phase timing of the cooperative step kernel (s_memtime stamps, OKENV_STAMPS build).

Build the instrumented library first (not the product build):
  hipcc -O3 -std=c++17 -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt --offload-arch=gfx950 -fPIC -shared \
        -fvisibility=hidden -Wno-unused-function -DOKENV_STAMPS -I include -I openkitchen_amd/csrc \
        -o tools/_build/libokenv_stamps.so openkitchen_amd/csrc/okenv_capi.hip openkitchen_amd/csrc/facade/*.cpp
Stamps: 0 pre-step, 1 phase 1, 2 barrier, 3 phase 2, 4 barrier, 5 epilogue (shader clock cycles per wave and step).
"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
import openkitchen_amd.buildlib as bl
bl.LIB_PATH = os.path.abspath("tools/_build/libokenv_stamps.so")
import openkitchen_amd as ok
from openkitchen_amd import capi
L = capi.load(build_if_missing=False)
L.okenv_debug_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
t = ok.Track("Silverstone")
for N, R in ((4096, 5), (4096, 64)):
    env = ok.BatchedEnvironment.from_track(t, N, num_rays=R)
    info = env.info()
    env.init_bench_state(0, 0)
    env.rollout_random(100, 1, 0, 0); env.sync()
    env.set_timing(True)
    env.rollout_random(100, 1, 0, 100)
    ms, n = env.get_timing()
    waves = N * info["lanes_per_agent"] // 64
    out = np.zeros((waves, 6), dtype=np.uint64)
    L.okenv_debug_stamps(env._h, out.ctypes.data_as(C.c_void_p), waves)
    tot = out.sum(axis=1).astype(np.float64)
    m = out.mean(axis=0)
    print("N %d R %d: %.1f us/step; stamp ticks per step (mean over waves): %s ; fractions %s ; total ticks/step %.1f" % (
        N, R, ms / n / 100 * 1e3, np.round(m / 100, 1), np.round(m / m.sum(), 3), m.sum() / 100))
    env.close()
