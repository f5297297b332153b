"""Known-answer tests of the oracle's raycast / epilogue / standstill restatement (the part of the reference
that exists only as CUDA, Environment/CollisionChecker.cu, and cannot be run here): simple geometries whose
answers follow from the reference's formulas by hand."""
import numpy as np

import _oracle as O


def cast(ox, oy, ang, segs):
    s = np.ascontiguousarray(np.asarray(segs, dtype=np.float32).reshape(-1))
    return O.lib().oracle_cast_ray(ox, oy, ang, s, s.size // 4)


def test_axis_aligned_hits(oracle):
    wall = [[10, -5, 10, 5]]
    assert cast(0, 0, 0.0, wall) == 10.0
    assert cast(0, 0, np.pi, wall) == 200.0  # pointing away: sensor range (Agent.h:10)
    assert cast(0, 0, 0.0, [[10, 1, 10, 5]]) == 200.0  # s outside [0,1]
    assert cast(0, 0, 0.0, [[10, 0, 10, 5]]) == 10.0  # s == 0 is a hit (s >= 0)
    assert cast(0, 0, 0.0, [[10, -5, 10, 0]]) == 10.0  # s == 1 is a hit (s <= 1)
    assert cast(0, 0, 0.0, [[250, -5, 250, 5]]) == 200.0  # beyond range
    assert cast(0, 0, 0.0, [[200, -5, 200, 5]]) == 200.0  # exactly at range: t <= range accepted, same value


def test_first_hit_is_minimum_and_order_independent(oracle):
    segs = [[50, -5, 50, 5], [20, -5, 20, 5], [35, -5, 35, 5]]
    assert cast(0, 0, 0.0, segs) == 20.0
    assert cast(0, 0, 0.0, segs[::-1]) == 20.0


def test_parallel_rejected(oracle):
    # |denom| < 1e-8 -> rejected (CollisionChecker.cu:23): a ray along a collinear segment sees nothing
    assert cast(0, 0, 0.0, [[5, 0, 15, 0]]) == 200.0


def test_hit_behind_origin_rejected(oracle):
    assert cast(0, 0, 0.0, [[-10, -5, -10, 5]]) == 200.0  # t < 0


def test_epilogue_rotation_quirk_and_crash(oracle):
    """rel = R(+rot) * (hit - origin) (CollisionChecker.cu:157-158): with rot = 90 deg and a
    single ray at 0 deg the world hit is straight 'up' (0,+t) and the 'robot frame' hit is (-t, ~0)."""
    segs = np.array([[-50, 30, 50, 30]], dtype=np.float32)
    env = O.OracleEnv(segs, 1, 1, np.zeros(1, dtype=np.float32))
    env.reset_agents([0], [0.0], [0.0], [90.0])
    env.collide()
    s = env.snapshot()
    assert abs(s["hit_y"][0, 0] - 30.0) < 1e-4 and abs(s["hit_x"][0, 0]) < 1e-4
    assert abs(s["rel_x"][0, 0] + 30.0) < 1e-4 and abs(s["rel_y"][0, 0]) < 1e-4
    assert abs(s["dist"][0, 0] - 30.0) < 1e-4 and s["crashed"][0] == 0
    # closer than sqrt(2): crash (min_dist2 < 2.0f, CollisionChecker.cu:167-171)
    env.reset_agents([0], [0.0], [28.7], [90.0])
    env.collide()
    assert env.get(O.F_CRASHED)[0] == 1
    env.reset_agents([0], [0.0], [28.5], [90.0])  # 1.5 px away: 2.25 >= 2 -> no crash
    env.collide()
    assert env.get(O.F_CRASHED)[0] == 0


def test_crashed_agent_keeps_stale_hits_and_does_not_move(oracle):
    segs = np.array([[-50, 30, 50, 30]], dtype=np.float32)
    env = O.OracleEnv(segs, 1, 1, np.zeros(1, dtype=np.float32))
    env.reset_agents([0], [0.0], [0.0], [90.0])
    env.set(O.F_THR, np.array([50.0], dtype=np.float32))
    env.step(1)
    before = env.snapshot()
    env.set(O.F_CRASHED, np.array([1], dtype=np.uint8))
    env.step(5)
    after = env.snapshot()
    for k in ("pos_x", "pos_y", "hit_x", "hit_y", "rel_x", "rel_y", "dist"):
        assert np.array_equal(before[k], after[k]), k


def test_standstill_timeout_fsm(oracle):
    """Environment.cpp:16-39: latch at ctr 0, evaluate at ctr >= 200 (i.e. on the 201st tick), < 20 px moved -> timed
    out; the counters survive Agent::reset."""
    segs = np.array([[1000, 1000, 1001, 1000]], dtype=np.float32)
    env = O.OracleEnv(segs, 2, 1, np.zeros(1, dtype=np.float32))
    env.reset_agents([0, 1], [100.0, 100.0], [100.0, 300.0], [0.0, 0.0])
    env.set(O.F_THR, np.array([0.0, 50.0], dtype=np.float32))  # agent 0 parked, agent 1 drives 0.8 px/step
    env.step(200)
    s = env.snapshot()
    assert s["crashed"].tolist() == [0, 0] and s["disp_ctr"].tolist() == [200, 200]
    env.step(1)
    s = env.snapshot()
    assert s["crashed"].tolist() == [1, 0] and s["timed_out"].tolist() == [1, 0] and s["disp_ctr"].tolist() == [0, 0]
    env.reset_agents([0], [100.0], [100.0], [0.0])
    s = env.snapshot()
    assert s["crashed"][0] == 0 and s["timed_out"][0] == 0
    env.step(1)
    assert env.get(O.F_DISP_CTR)[0] == 1 and env.get(O.F_DISP_X)[0] == 100.0


def test_acceleration_mode_clamps_speed(oracle):
    segs = np.array([[1000, 1000, 1001, 1000]], dtype=np.float32)
    env = O.OracleEnv(segs, 1, 1, np.zeros(1, dtype=np.float32))
    env.reset_agents([0], [100.0], [100.0], [0.0])
    env.set(O.F_MODE, np.array([1], dtype=np.uint8))
    env.set(O.F_THR, np.array([500.0], dtype=np.float32))
    env.step(60)
    assert env.get(O.F_SPEED)[0] == 100.0  # kSpeedLimit
    env.set(O.F_THR, np.array([-5000.0], dtype=np.float32))
    env.step(60)
    assert env.get(O.F_SPEED)[0] == 0.0


def test_segment_order_of_track_segments(oracle):
    t = O.Track("Austin")
    P = t.P
    assert t.S == 4 * P
    seg = t.segments
    assert np.array_equal(seg[0], [t.li[0], t.li[1], t.li[2], t.li[3]])  # LI run first
    assert np.array_equal(seg[P - 1], [t.lo[0], t.lo[1], t.lo[2], t.lo[3]])  # then LO
    assert np.array_equal(seg[2 * (P - 1)], [t.ri[0], t.ri[1], t.ri[2], t.ri[3]])  # then RI
    assert np.array_equal(seg[3 * (P - 1)], [t.ro[0], t.ro[1], t.ro[2], t.ro[3]])  # then RO
    closers = seg[4 * (P - 1):]
    for row, poly in zip(closers, (t.li, t.ri, t.lo, t.ro)):  # closers LI, RI, LO, RO
        assert np.array_equal(row, [poly[2 * (P - 1)], poly[2 * (P - 1) + 1], poly[0], poly[1]])
