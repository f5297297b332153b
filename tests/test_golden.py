"""Committed golden fixtures (tests/golden/, generator make_golden.py): reference-derived pins that travel to the
GPU box, where /root/reference does not exist."""
import hashlib
import os

import numpy as np
import pytest

import _oracle as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TRACKS = ["Austin", "Silverstone", "Monza", "Spa"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def check_track(t, g, name):
    assert t.P == int(g[name + "_P"])
    for k in O.Track.KEYS:
        a = getattr(t, k)
        assert sha(a) == str(g["%s_sha_%s" % (name, k)]), (name, k)
        assert np.array_equal(bits(a[:16]), bits(g["%s_head_%s" % (name, k)]))
        assert np.array_equal(bits(a[-16:]), bits(g["%s_tail_%s" % (name, k)]))


@pytest.mark.parametrize("name", TRACKS)
def test_oracle_track_matches_reference_fixture(oracle, name):
    g = np.load(os.path.join(GOLD, "ref_tracks.npz"))
    t = O.Track(name, "oracle")
    check_track(t, g, name)
    got = np.zeros(256, dtype=np.int32)
    O.lib().oracle_nearest_track_idx(t.x, t.y, t.P, g[name + "_probe_x"], g[name + "_probe_y"], 256, got)
    assert np.array_equal(got, g[name + "_probe_idx"])
    # RaceTrack::getNearestDistanceToTrackBoundary / getDistanceToLaneCenter (RaceTrack.cpp:33-72)
    bd, lc = np.zeros(256, dtype=np.float32), np.zeros(256, dtype=np.float32)
    O.lib().oracle_boundary_distance(t.li, t.ri, t.P, g[name + "_probe_x"], g[name + "_probe_y"], 256, bd)
    O.lib().oracle_lane_center_distance(t.x, t.y, t.wl, t.wr, t.P, g[name + "_probe_x"], g[name + "_probe_y"], 256, lc)
    assert np.array_equal(bits(bd), bits(g[name + "_probe_boundary"]))
    assert np.array_equal(bits(lc), bits(g[name + "_probe_lane"]))


@pytest.mark.parametrize("name", TRACKS)
def test_product_track_matches_reference_fixture(ok, name):
    g = np.load(os.path.join(GOLD, "ref_tracks.npz"))
    t = ok.Track(name)
    check_track(t, g, name)
    # the product's RaceTrack query methods (C ABI okenv_track_queries) against the reference's answers
    bd, lc = t.queries(g[name + "_probe_x"], g[name + "_probe_y"])
    assert np.array_equal(bits(bd), bits(g[name + "_probe_boundary"]))
    assert np.array_equal(bits(lc), bits(g[name + "_probe_lane"]))


@pytest.mark.parametrize("mode", [0, 1])
def test_oracle_kinematics_match_reference_fixture(oracle, mode):
    """glibc-trig mode reproduces the reference's Agent::move bit for bit; the parity definition (ok_sincosf) stays
    within 1e-5 of it on pose, the tolerance BASELINE.json states."""
    g = np.load(os.path.join(GOLD, "ref_kinematics.npz"))
    thr, steer = g["m%d_thr" % mode], g["m%d_steer" % mode]
    seg = np.array([[0, 0, 1, 0]], dtype=np.float32)
    for trig_mode in (1, 0):
        env = O.OracleEnv(seg, 1, 1, np.zeros(1, dtype=np.float32))
        env.reset_agents([0], [700.0], [500.0], [33.0])
        env.set(O.F_MODE, np.array([mode], dtype=np.uint8))
        O.lib().oracle_set_trig_mode(trig_mode)
        try:
            for s in range(thr.size):
                env.set(O.F_THR, thr[s:s + 1])
                env.set(O.F_STEER, steer[s:s + 1])
                env.set(O.F_DISP_CTR, np.array([1], dtype=np.uint32))
                env.move_only()
                got = {"x": env.get(O.F_POS_X)[0], "y": env.get(O.F_POS_Y)[0], "rot": env.get(O.F_ROT)[0],
                       "speed": env.get(O.F_SPEED)[0], "acc": env.get(O.F_ACC)[0]}
                for k, v in got.items():
                    w = g["m%d_%s" % (mode, k)][s]
                    if trig_mode == 1:
                        assert np.float32(v).view(np.uint32) == w.view(np.uint32), (s, k)
                    else:
                        assert abs(float(v) - float(w)) <= 1e-5 * max(1.0, abs(float(w))), (s, k, v, w)
        finally:
            O.lib().oracle_set_trig_mode(0)
    assert np.array_equal(g["default_rays"], np.arange(-70, 71, 10, dtype=np.float32))
    assert list(g["reset_probe"]) == [11.0, 22.0, 33.0, 0, 0, 0, 0, 0, 0]


def run_c1(env_factory):
    g = np.load(os.path.join(GOLD, "c1_trajectory.npz"))
    env = env_factory()
    env.init_bench_state(0, 0)
    for chunk in range(9):
        env.rollout_random(50, int(g["seed"]), 0, chunk * 50)
        s = env.snapshot()
        for k in ("pos_x", "pos_y", "rot", "speed", "crashed", "timed_out", "disp_ctr", "dist", "rel_x", "rel_y"):
            want = g["s%03d_%s" % ((chunk + 1) * 50, k)]
            assert np.array_equal(bits(s[k]), bits(want)), (chunk, k)


def test_oracle_reproduces_c1_trajectory_fixture(oracle):
    t = O.Track("Austin", "oracle")
    fan = O.default_ray_fan(16)
    run_c1(lambda: O.OracleEnv(t.segments, 64, 16, fan, (t.x, t.y, t.heading)))


@pytest.mark.gpu
def test_gpu_reproduces_c1_trajectory_fixture(gpu):
    t = gpu.Track("Austin")
    run_c1(lambda: gpu.BatchedEnvironment.from_track(t, 64, 16))
