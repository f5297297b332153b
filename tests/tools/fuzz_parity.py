"""Randomised differential test of the device path against the CPU oracle (run on the GPU box):
random track, population, fan (including non-monotone and duplicate angles), movement mode, sensor offset, cell size,
lane-group width and phase-1 range, with the bench driver loop or host actions + auto-reset.  Every state field is
compared bit for bit.  usage: python tests/tools/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import _oracle as O  # noqa: E402
import openkitchen_amd as ok  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build_oracle(with_ref=False)
KEYS = ["pos_x", "pos_y", "rot", "speed", "acc", "thr", "steer", "crashed", "timed_out", "disp_ctr", "disp_x", "disp_y", "disp_to",
        "hit_x", "hit_y", "rel_x", "rel_y", "dist"]
tracks = {n: ok.Track(n) for n in ("Austin", "Silverstone", "Monza", "Spa")}
t0, cases, steps_total = time.time(), 0, 0
while time.time() - t0 < budget:
    name = rng.choice(list(tracks))
    t = tracks[name]
    N = int(rng.choice([1, 7, 33, 64, 150, 256]))
    R = int(rng.choice([1, 2, 5, 9, 15, 16, 31, 32, 64, 70]))
    fan = np.sort(rng.uniform(-110, 110, R)).astype(np.float32) if rng.random() < 0.5 else ok.default_ray_fan(R)
    if rng.random() < 0.2:
        fan = rng.permutation(fan).astype(np.float32)
    mode = int(rng.integers(0, 2))
    cell = float(rng.choice([0.0, 0.0, 9.0, 14.0, 33.0]))
    for key, choices in (("OKENV_LANES_PER_AGENT", [None, None, "8", "64"]), ("OKENV_PHASE1_RANGE", [None, None, "0", "2", "20", "90"]), ("OKENV_AGENTS_PER_BLOCK", [None, None, "0", "1", "2"])):
        v = choices[int(rng.integers(0, len(choices)))]
        if v is None or (key == "OKENV_LANES_PER_AGENT" and int(v) < R and R > 64):
            os.environ.pop(key, None)
        else:
            os.environ[key] = v
    dev = ok.BatchedEnvironment(t.segments, N, fan, grid_cell=cell, centerline=(t.x, t.y, t.heading))
    dev.set_lane_bounds(t.li, t.ri)
    orc = O.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    orc.set_lane_bounds(t.li, t.ri)
    off = float(rng.choice([0.0, 0.0, 3.5, -2.0]))
    dev.set_sensor_offset(off)
    O.lib().oracle_env_set_sensor_offset(orc.h, off)
    seed = int(rng.integers(0, 2 ** 31))
    what = "track %s N %d R %d mode %d cell %g offset %g seed %d env %s" % (
        name, N, R, mode, cell, off, seed, {k: os.environ.get(k) for k in ("OKENV_LANES_PER_AGENT", "OKENV_PHASE1_RANGE")})
    if rng.random() < 0.5:
        dev.init_bench_state(5, mode)
        orc.init_bench_state(5, mode)
        for chunk in range(4):
            n = int(rng.integers(1, 120))
            dev.rollout_random(n, seed, 5, steps_total % 100000)
            orc.rollout_random(n, seed, 5, steps_total % 100000, threads=8)
            steps_total += n
    else:
        flags = int(rng.integers(0, 8))
        for env in (dev, orc):
            env.reset_random(None, 3, seed, 0, 0)
            env.set(O.F_MODE, np.full(N, mode, dtype=np.uint8))
            env.set_auto_reset(True, flags, seed, 11)
        for chunk in range(int(rng.integers(20, 120))):
            thr = (rng.uniform(0, 100, N) if mode == 0 else rng.uniform(-0.3, 0.6, N)).astype(np.float32)
            steer = rng.uniform(-6, 6, N).astype(np.float32)
            n = int(rng.integers(1, 4))
            for env in (dev, orc):
                env.set(O.F_THR, thr)
                env.set(O.F_STEER, steer)
                env.step(n)
            steps_total += n
    d, o = dev.snapshot(), orc.snapshot()
    for k in KEYS:
        a, b = np.ascontiguousarray(d[k]), np.ascontiguousarray(o[k])
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            print("MISMATCH in %s (%d places, first %s): %s" % (k, len(bad), bad[0], what), flush=True)
            sys.exit(1)
    cases += 1
    dev.close()
    if cases % 20 == 0:
        print("%d cases, %d steps, %.0f s" % (cases, steps_total, time.time() - t0), flush=True)
print("fuzz ok: %d cases, %d environment steps compared bit for bit in %.0f s" % (cases, steps_total, time.time() - t0))
