"""okenv_step_packed (the facade's one-copy-each-way exchange) through the C ABI against the oracle: full steps with
and without the DisplacementStats members, and the collision-only form CollisionChecker::checkCollision uses."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

REC = np.dtype([("pos_x", "<f4"), ("pos_y", "<f4"), ("rot", "<f4"), ("speed", "<f4"), ("acc", "<f4"), ("throttle", "<f4"),
                ("steer", "<f4"), ("disp_x", "<f4"), ("disp_y", "<f4"), ("disp_ctr", "<u4"), ("mode", "u1"), ("crashed", "u1"),
                ("timed_out", "u1"), ("disp_timed_out", "u1")])
WITH_STATS, COLLIDE_ONLY = 1, 2
ORACLE_KEYS = {"pos_x": "pos_x", "pos_y": "pos_y", "rot": "rot", "speed": "speed", "acc": "acc", "crashed": "crashed",
               "timed_out": "timed_out", "disp_x": "disp_x", "disp_y": "disp_y", "disp_ctr": "disp_ctr", "disp_timed_out": "disp_to"}


def packed_step(gpu, dev, rec, flags):
    hits = np.zeros((dev.N, dev.R, 2), dtype=np.float32)
    gpu.capi.check(dev._L.okenv_step_packed(dev._h, rec.ctypes.data_as(C.c_void_p), rec.ctypes.data_as(C.c_void_p),
                                            hits.ctypes.data_as(C.c_void_p), flags), dev._h)
    return hits


def test_record_layout():
    assert REC.itemsize == 44


@pytest.mark.parametrize("mode", [0, 1])
def test_packed_steps_match_oracle(gpu, oracle, mode):
    N, R = 300, 9
    t = gpu.Track("Austin")
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    rng = np.random.default_rng(mode)
    idx = rng.integers(0, t.P, N)
    rec = np.zeros(N, dtype=REC)
    rec["pos_x"], rec["pos_y"], rec["rot"], rec["mode"] = t.x[idx], t.y[idx], t.heading[idx], mode
    orc.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    orc.set(oracle.F_MODE, np.full(N, mode, dtype=np.uint8))
    crashes = 0
    for it in range(260):
        thr = rng.uniform(30, 100, N).astype(np.float32) if mode == 0 else rng.uniform(-0.2, 0.6, N).astype(np.float32)
        steer = rng.uniform(-5, 5, N).astype(np.float32)
        rec["throttle"], rec["steer"] = thr, steer
        orc.set(oracle.F_THR, thr)
        orc.set(oracle.F_STEER, steer)
        hits = packed_step(gpu, dev, rec, WITH_STATS)
        orc.step(1)
        if it % 20 == 0 or it == 259:
            o = orc.snapshot()
            for k, ok_ in ORACLE_KEYS.items():
                a, b = np.ascontiguousarray(rec[k]), np.ascontiguousarray(o[ok_])
                assert a.tobytes() == b.astype(a.dtype).tobytes(), (it, k)
            assert np.array_equal(hits[..., 0].view(np.uint32), o["rel_x"].view(np.uint32)), it
            assert np.array_equal(hits[..., 1].view(np.uint32), o["rel_y"].view(np.uint32)), it
            crashes = int(o["crashed"].sum())
    assert crashes > 0 and rec["disp_ctr"].max() > 0


def test_collide_only_and_untouched_stats(gpu, oracle):
    N, R = 64, 15
    t = gpu.Track("Monza")
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    rng = np.random.default_rng(4)
    idx = rng.integers(0, t.P, N)
    rec = np.zeros(N, dtype=REC)
    rec["pos_x"], rec["pos_y"], rec["rot"] = t.x[idx] + rng.normal(0, 3, N), t.y[idx] + rng.normal(0, 3, N), t.heading[idx]
    rec["throttle"], rec["disp_ctr"], rec["disp_x"] = 50.0, 77, 1.5   # must come back untouched without WITH_STATS
    orc.reset_agents(np.arange(N), rec["pos_x"], rec["pos_y"], rec["rot"])
    before = rec.copy()
    hits = packed_step(gpu, dev, rec, COLLIDE_ONLY)
    orc.collide()
    o = orc.snapshot()
    assert np.array_equal(rec["pos_x"], before["pos_x"]) and np.array_equal(rec["rot"], before["rot"])  # no kinematics
    assert (rec["disp_ctr"] == 77).all() and (rec["disp_x"] == 1.5).all()
    assert np.array_equal(rec["crashed"], o["crashed"])
    assert np.array_equal(hits[..., 0].view(np.uint32), o["rel_x"].view(np.uint32))
    assert dev.step_count == 0
    packed_step(gpu, dev, rec, 0)  # a full step, standstill bookkeeping kept on the device
    assert dev.step_count == 1 and (rec["disp_ctr"] == 77).all()
    assert (dev.get(gpu.capi.F_DISP_CTR) <= 1).all()
