"""Pins the CPU oracle against the reference's own compilable translation units (oracle/_ref/libokref.so =
reference Environment/Agent.cpp + Environment/RaceTrack.cpp built from /root/reference)
and the product's host-side track builder against both.  Runs only where the reference tree exists (the
development container); the committed fixtures in tests/golden/ carry the same pin to the GPU box."""
import ctypes as C
import os

import numpy as np
import pytest

import _oracle as O

needs_ref = pytest.mark.skipif(not os.path.isdir("/root/reference/Environment"), reason="reference tree not present")
TRACKS = ["Austin", "Silverstone", "Monza", "Spa"]


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


@needs_ref
@pytest.mark.parametrize("name", TRACKS)
def test_track_geometry_bit_equal_to_reference(oracle, name):
    a, b = O.Track(name, "oracle"), O.Track(name, "ref")
    assert a.P == b.P
    for k in O.Track.KEYS:
        assert np.array_equal(bits(getattr(a, k)), bits(getattr(b, k))), k


@needs_ref
@pytest.mark.parametrize("name", TRACKS)
def test_product_track_builder_bit_equal_to_reference(oracle, ok, name):
    a, b = ok.Track(name), O.Track(name, "ref")
    assert a.P == b.P and a.S == 4 * a.P
    for k in O.Track.KEYS:
        assert np.array_equal(bits(getattr(a, k)), bits(getattr(b, k))), k


@needs_ref
def test_layout_facts(oracle):
    f = np.zeros(4, dtype=np.uint32)
    O.ref().ref_layout_facts(f)
    assert f[0] == 24 and f[1] == 16  # sizeof(Ray_), sizeof(Segment2d)
    assert f[2] == 0x3C8EFA35  # kDeg2Rad
    assert f[3] == 3  # RaceTrack::kStartingIdx
    rays = np.zeros(64, dtype=np.float32)
    n = O.ref().ref_agent_default_rays(rays, 64)
    assert n == 15 and np.array_equal(rays[:15], np.arange(-70, 71, 10, dtype=np.float32))


@needs_ref
@pytest.mark.parametrize("name", ["Austin", "Spa"])
def test_nearest_index_equal_to_reference(oracle, name):
    t = O.Track(name, "oracle")
    rng = np.random.default_rng(1)
    qx = rng.uniform(0, 1600, 3000).astype(np.float32)
    qy = rng.uniform(0, 1400, 3000).astype(np.float32)
    qx[:t.P], qy[:t.P] = t.x, t.y
    want = np.zeros(3000, dtype=np.int32)
    h = O.ref().ref_track_load(O.track_path(name).encode())
    O.ref().ref_nearest_track_idx(h, qx, qy, 3000, want)
    O.ref().ref_track_free(h)
    got = np.zeros(3000, dtype=np.int32)
    O.lib().oracle_nearest_track_idx(t.x, t.y, t.P, qx, qy, 3000, got)
    assert np.array_equal(got, want)


@needs_ref
@pytest.mark.parametrize("name", ["Austin", "Silverstone", "Monza", "Spa"])
def test_track_queries_equal_to_reference(oracle, name):
    """RaceTrack::getNearestDistanceToTrackBoundary / getDistanceToLaneCenter: oracle restatement and the product's own
    RaceTrack (through the C ABI) against the reference's compiled RaceTrack.cpp, bit for bit, 3000 probes."""
    import openkitchen_amd as ok
    t = O.Track(name, "oracle")
    rng = np.random.default_rng(2)
    qx = rng.uniform(0, 1600, 3000).astype(np.float32)
    qy = rng.uniform(0, 1400, 3000).astype(np.float32)
    qx[:t.P], qy[:t.P] = t.x, t.y
    qx[t.P:2 * t.P], qy[t.P:2 * t.P] = t.li[0::2], t.li[1::2]
    wb, wl = np.zeros(3000, dtype=np.float32), np.zeros(3000, dtype=np.float32)
    h = O.ref().ref_track_load(O.track_path(name).encode())
    O.ref().ref_track_queries(h, qx, qy, 3000, wb, wl)
    O.ref().ref_track_free(h)
    gb, gl = np.zeros(3000, dtype=np.float32), np.zeros(3000, dtype=np.float32)
    O.lib().oracle_boundary_distance(t.li, t.ri, t.P, qx, qy, 3000, gb)
    O.lib().oracle_lane_center_distance(t.x, t.y, t.wl, t.wr, t.P, qx, qy, 3000, gl)
    assert np.array_equal(gb.view(np.uint32), wb.view(np.uint32))
    assert np.array_equal(gl.view(np.uint32), wl.view(np.uint32))
    pb, pl = ok.Track(name).queries(qx, qy)
    assert np.array_equal(pb.view(np.uint32), wb.view(np.uint32))
    assert np.array_equal(pl.view(np.uint32), wl.view(np.uint32))


@needs_ref
@pytest.mark.parametrize("mode", [0, 1])
def test_kinematics_bit_equal_to_reference_agent(oracle, mode):
    """Oracle kinematics in glibc-trig mode == the reference's Agent::move, bit for bit, over 3000 steps with
    random actions (rot drifts to large unwrapped angles)."""
    rng = np.random.default_rng(2 + mode)
    n = 3000
    thr = (rng.uniform(0, 100, n) if mode == 0 else rng.uniform(-0.4, 0.6, n)).astype(np.float32)
    steer = rng.uniform(-5, 6, n).astype(np.float32)
    out = [np.zeros(n, dtype=np.float32) for _ in range(5)]
    O.ref().ref_agent_rollout(mode, 700.0, 500.0, 33.0, thr, steer, n, *out)
    seg = np.array([[0, 0, 1, 0]], dtype=np.float32)
    env = O.OracleEnv(seg, 1, 1, np.zeros(1, dtype=np.float32))
    env.reset_agents([0], [700.0], [500.0], [33.0])
    env.set(O.F_MODE, np.array([mode], dtype=np.uint8))
    O.lib().oracle_set_trig_mode(1)
    try:
        for s in range(n):
            env.set(O.F_THR, thr[s:s + 1])
            env.set(O.F_STEER, steer[s:s + 1])
            env.set(O.F_DISP_CTR, np.array([1], dtype=np.uint32))  # keep the standstill FSM from firing
            env.move_only()
            got = [env.get(f)[0] for f in (O.F_POS_X, O.F_POS_Y, O.F_ROT, O.F_SPEED, O.F_ACC)]
            for g, w in zip(got, out):
                assert np.float32(g).view(np.uint32) == w[s].view(np.uint32), (s, g, w[s])
    finally:
        O.lib().oracle_set_trig_mode(0)


@needs_ref
def test_agent_reset_semantics(oracle):
    out = np.zeros(9, dtype=np.float32)
    O.ref().ref_agent_reset_probe(11.0, 22.0, 33.0, out)
    assert list(out) == [11.0, 22.0, 33.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0]
    seg = np.array([[0, 0, 1, 0]], dtype=np.float32)
    env = O.OracleEnv(seg, 1, 1, np.zeros(1, dtype=np.float32))
    for f, v in ((O.F_SPEED, 5), (O.F_ACC, 6), (O.F_THR, 7), (O.F_STEER, 8)):
        env.set(f, np.array([v], dtype=np.float32))
    env.set(O.F_CRASHED, np.array([1], dtype=np.uint8))
    env.set(O.F_TIMED_OUT, np.array([1], dtype=np.uint8))
    env.set(O.F_DISP_CTR, np.array([77], dtype=np.uint32))
    env.reset_agents([0], [11.0], [22.0], [33.0])
    s = env.snapshot()
    got = [s[k][0] for k in ("pos_x", "pos_y", "rot", "speed", "acc", "crashed", "timed_out", "thr", "steer")]
    assert got == [11.0, 22.0, 33.0, 0.0, 0.0, 0, 0, 0.0, 0.0]
    assert s["disp_ctr"][0] == 77  # DisplacementStats survive resets
