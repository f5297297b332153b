"""Diagnostic (OKENV_STAMPS build): how unevenly the waves of one C2 launch finish.

  tools/build_variant.sh stamps -DOKENV_STAMPS && python tools/wave_imbalance.py [steps_per_launch]
Per wave: in-loop shader cycles by phase (stamps 0 + 1 policy + pre-step, 3 phase 1, 5 phase 2, 6 epilogue) and start / end on the
100 MHz real-time clock.  Prints the spread of the per-wave totals and of the per-SIMD sums (waves w, w+4, w+8, w+12 of a
workgroup share a SIMD), i.e. how much of the launch is tail."""
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
spl = int(sys.argv[1]) if len(sys.argv) > 1 else 100
t = ok.Track("Silverstone")
N, R = 4096, 64
env = ok.BatchedEnvironment.from_track(t, N, num_rays=R)
env.init_bench_state(0, 0)
env.rollout_random(200, 1234, 0, 0); env.sync()
env.set_timing(True)
env.rollout_random(spl, 1234, 0, 200)
ms, n = env.get_timing()
waves = N
out = np.zeros((waves, 24), dtype=np.uint64)
assert L.okenv_debug_stamps(env._h, out.ctypes.data_as(C.c_void_p), waves) == waves
cyc = np.stack([out[:, 0] + out[:, 1], out[:, 3], out[:, 5], out[:, 6]], axis=1).astype(np.float64)
tot = cyc.sum(axis=1) / spl
start, end = out[:, 2].astype(np.float64), out[:, 4].astype(np.float64)
t0 = start.min()
life = (end - start) * 10e-3  # us
print("launch %.1f us (%.2f us/step, events); waves: first start %.1f us spread, last end %.1f us after first start" % (ms * 1e3, ms * 1e3 / spl, (start.max() - t0) * 10e-3, (end.max() - t0) * 10e-3))
print("wave lifetime us: mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f  -> mean/max %.3f" % (life.mean(), np.percentile(life, 50), np.percentile(life, 90), np.percentile(life, 99), life.max(), life.mean() / life.max()))
print("in-loop cycles per wave-step: mean %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f ; phase means %s" % (tot.mean(), np.percentile(tot, 50), np.percentile(tot, 90), np.percentile(tot, 99), tot.max(), np.round(cyc.mean(axis=0) / spl)))
wg = life.reshape(-1, 16)
print("per-workgroup end (max wave lifetime): mean %.1f max %.1f us" % (wg.max(axis=1).mean(), wg.max(axis=1).max()))
simd = np.stack([wg[:, i::4].max(axis=1) for i in range(4)], axis=1)
print("per-SIMD last finisher: mean %.1f p90 %.1f max %.1f us" % (simd.mean(), np.percentile(simd, 90), simd.max()))
for name, arr in (("phase1", cyc[:, 1] / spl), ("phase2", cyc[:, 2] / spl), ("pre", cyc[:, 0] / spl)):
    print("  %s cycles per step: mean %.0f p10 %.0f p90 %.0f max %.0f" % (name, arr.mean(), np.percentile(arr, 10), np.percentile(arr, 90), arr.max()))
wp = out[:, 8:18].astype(np.float64) / spl
names = ["setup", "cell-entry", "points", "exact", "cell-leave"]
for ph, off in (("phase1", 0), ("phase2", 5)):
    print("  %s walk internals (busiest lane), cycles per step: %s ; sum %.0f" % (ph, ", ".join("%s %.0f" % (names[i], wp[:, off + i].mean()) for i in range(5)), wp[:, off:off + 5].sum(axis=1).mean()))
env.close()
