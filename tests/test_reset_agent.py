"""Environment::resetAgent (Environment/Environment.cpp:79-122): the oracle's restatement against an independent
Python reading of the same lines (CPU), and the device-side reset / per-agent auto-reset against the oracle (GPU).

The reference draws from raylib's GetRandomValue (un-vendored, global state), so nothing here can be pinned to the
reference's numbers: what is pinned is the arithmetic around the draws (inclusive integer ranges, the +-(45 + draw)
heading offset with alternating sign, alpha = draw / 100, the lane interpolation's operation order, kStartingIdx).
"""
import numpy as np
import pytest

from test_math import philox_ref

RANDOM_POINT, RANDOM_LANE, RANDOM_HEADING, ONLY_DONE = 1, 2, 4, 8
f32 = np.float32


def expected_reset(t, seed, agent, epoch, ctr, flags):
    """Environment.cpp:83-121 with GetRandomValue(lo, hi) := lo + floor(word * (hi - lo + 1) / 2^32)."""
    w = philox_ref([agent, epoch, 1, 0], [seed, 0x6F6B656E])
    rnd = lambda word, lo, hi: lo + ((word * (hi - lo + 1)) >> 32)
    pick = bool(flags & RANDOM_POINT)
    idx = rnd(w[0], 0, t.P - 1) if pick else 3
    off = f32(0)
    if pick and flags & RANDOM_HEADING:
        off = f32(rnd(w[1], 0, 45))
        off = f32(f32(off + f32(45)) * f32(-1)) if ctr % 2 == 0 else f32(off + f32(45))
    if pick and flags & RANDOM_LANE:
        alpha = f32(f32(rnd(w[2], 10, 90)) / f32(100))
        lx, ly, rx, ry = t.li[2 * idx], t.li[2 * idx + 1], t.ri[2 * idx], t.ri[2 * idx + 1]
        x = f32(f32(lx * alpha) + f32(rx * f32(f32(1) - alpha)))
        y = f32(f32(ly * alpha) + f32(ry * f32(f32(1) - alpha)))
    else:
        x, y = t.x[idx], t.y[idx]
    return idx, f32(x), f32(y), f32(t.heading[idx] + off), off


def make_oracle_env(oracle, track_name, N, R=5):
    t = oracle.Track(track_name)
    fan = oracle.default_ray_fan(R)
    env = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    env.set_lane_bounds(t.li, t.ri)
    return t, env


@pytest.mark.parametrize("flags", [0, 1, 3, 5, 7, 2, 6])
def test_oracle_reset_matches_independent_restatement(oracle, flags):
    N = 257
    t, env = make_oracle_env(oracle, "Silverstone", N)
    rng = np.random.default_rng(flags)
    env.set(oracle.F_SPEED, rng.uniform(1, 50, N))
    env.set(oracle.F_ACC, rng.uniform(1, 50, N))
    env.set(oracle.F_THR, rng.uniform(1, 50, N))
    env.set(oracle.F_STEER, rng.uniform(1, 5, N))
    env.set(oracle.F_CRASHED, np.ones(N))
    env.set(oracle.F_TIMED_OUT, np.ones(N))
    env.set(oracle.F_DISP_CTR, np.full(N, 17))
    seed, epoch, base = 99, 1000, 4096
    env.reset_random(None, flags, seed, epoch, base)
    s = env.snapshot()
    offs = []
    for a in range(N):
        idx, x, y, rot, off = expected_reset(t, seed, base + a, epoch, epoch + a, flags)
        assert s["pos_x"][a].tobytes() == x.tobytes() and s["pos_y"][a].tobytes() == y.tobytes(), (a, idx)
        assert s["rot"][a].tobytes() == rot.tobytes()
        offs.append(float(off))
    # Agent::reset (Agent.cpp:123-135): motion state and flags cleared, DisplacementStats untouched
    for k in ("speed", "acc", "thr", "steer", "crashed", "timed_out"):
        assert not s[k].any(), k
    assert (s["disp_ctr"] == 17).all()
    offs = np.array(offs)
    if flags & RANDOM_POINT and flags & RANDOM_HEADING:
        assert (np.abs(offs) >= 45).all() and (np.abs(offs) <= 90).all()
        assert (offs[0::2] < 0).all() and (offs[1::2] > 0).all()  # epoch is even: even entries turn negative
        assert len(np.unique(np.abs(offs))) > 30
    else:
        assert not offs.any()
    if not flags & RANDOM_POINT:
        assert (s["pos_x"] == t.x[3]).all() and (s["rot"] == t.heading[3]).all()


def test_oracle_reset_lane_points_lie_between_the_inner_boundaries(oracle):
    N = 4000
    t, env = make_oracle_env(oracle, "Monza", N)
    env.reset_random(None, RANDOM_POINT | RANDOM_LANE, 5, 0, 0)
    s = env.snapshot()
    l, r = t.li.reshape(-1, 2), t.ri.reshape(-1, 2)
    # the point is l*a + r*(1-a) for some track index and a in {0.10 .. 0.90}: recover both
    d = np.hypot(s["pos_x"][:, None] - t.x[None, :], s["pos_y"][:, None] - t.y[None, :])
    assert (d.min(axis=1) < np.hypot(*(l - r).T).max()).all()
    alphas = set()
    for a in range(0, N, 40):
        w = philox_ref([a, 0, 1, 0], [5, 0x6F6B656E])
        idx = (w[0] * t.P) >> 32
        den = l[idx] - r[idx]
        k = int(np.argmax(np.abs(den)))
        alpha = (np.array([s["pos_x"][a], s["pos_y"][a]])[k] - r[idx][k]) / den[k]
        assert 0.0999 < alpha < 0.9001
        alphas.add(round(float(alpha), 2))
    assert len(alphas) > 40


def test_oracle_reset_subset_and_only_done(oracle):
    N = 64
    t, env = make_oracle_env(oracle, "Austin", N)
    env.reset_random(None, 0, 0, 0, 0)
    before = env.snapshot()
    crashed = np.zeros(N, dtype=np.uint8)
    crashed[[3, 10, 11, 40]] = 1
    env.set(oracle.F_CRASHED, crashed)
    idx = np.array([40, 3, 5, 10], dtype=np.int32)
    env.reset_random(idx, RANDOM_POINT | RANDOM_HEADING | ONLY_DONE, 7, 12, 0)
    s = env.snapshot()
    moved = np.flatnonzero(s["pos_x"] != before["pos_x"])
    assert set(moved) <= {40, 3, 10} and len(moved) >= 2
    assert s["crashed"][11] == 1 and not s["crashed"][[40, 3, 10]].any()
    for j, a in enumerate(idx):
        if a == 5:
            continue  # not crashed: skipped, but it still occupies call slot j
        _, x, y, rot, _ = expected_reset(t, 7, int(a), 12, 12 + j, RANDOM_POINT | RANDOM_HEADING)
        assert s["pos_x"][a] == x and s["rot"][a] == rot


def test_oracle_auto_reset_restarts_crashed_agents(oracle):
    N, R = 48, 9
    t, env = make_oracle_env(oracle, "Austin", N, R)
    env.reset_random(None, RANDOM_POINT, 1, 0, 0)
    env.set(oracle.F_MODE, np.zeros(N))
    env.set_auto_reset(True, RANDOM_POINT | RANDOM_LANE | RANDOM_HEADING, 31, 0)
    rng = np.random.default_rng(3)
    restarts = 0
    for step in range(260):
        env.set(oracle.F_THR, rng.uniform(40, 100, N))
        env.set(oracle.F_STEER, rng.uniform(-5, 5, N))
        was_crashed = env.get(oracle.F_CRASHED).astype(bool)
        env.step(1)
        s = env.snapshot()
        if was_crashed.any():
            a = int(np.flatnonzero(was_crashed)[0])
            # the reset step is the initial-observation step: zero action, zero speed, pose = reset pose
            _, x, y, rot, _ = expected_reset(t, 31, a, step, a + step, 7)
            assert s["pos_x"][a] == x and s["pos_y"][a] == y and s["rot"][a] == rot
            assert s["thr"][a] == 0 and s["speed"][a] == 0
            restarts += int(was_crashed.sum())
    assert env.step_count == 260
    assert restarts > 10


# ---- device -----------------------------------------------------------------------------------------------------


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same(dev, orc, where):
    d, o = dev.snapshot(), orc.snapshot()
    for k in o:
        assert np.array_equal(bits(d[k]), bits(o[k])), "%s: field %s differs at %s" % (
            where, k, np.argwhere(bits(d[k]) != bits(o[k]))[:3].tolist())
    return o


def make_pair(gpu, oracle, track_name, N, R):
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    orc.set_lane_bounds(t.li, t.ri)
    return t, dev, orc


@pytest.mark.gpu
@pytest.mark.parametrize("flags", [0, 1, 3, 5, 7])
def test_device_reset_random_bit_exact(gpu, oracle, flags):
    N, R = 1000, 8
    t, dev, orc = make_pair(gpu, oracle, "Spa", N, R)
    rng = np.random.default_rng(flags)
    for env in (dev, orc):
        env.reset_random(None, RANDOM_POINT, 3, 0, 0)
    speed, crashed = rng.uniform(1, 50, N).astype(f32), (rng.random(N) < 0.4).astype(np.uint8)
    for env in (dev, orc):
        env.set(oracle.F_SPEED, speed)
        env.set(oracle.F_CRASHED, crashed)
    dev.reset_random(None, flags, 11, 6, 100)
    orc.reset_random(None, flags, 11, 6, 100)
    assert_same(dev, orc, "all agents")
    idx = rng.permutation(N)[:300].astype(np.int32)
    for env in (dev, orc):
        env.set(oracle.F_CRASHED, crashed)
    dev.reset_random(idx, flags | ONLY_DONE, 12, 7, 100)
    orc.reset_random(idx, flags | ONLY_DONE, 12, 7, 100)
    o = assert_same(dev, orc, "subset, only done")
    assert o["crashed"].sum() > 0  # crashed agents outside idx stay crashed
    dev.step(1)
    orc.step(1)
    assert_same(dev, orc, "step after reset")


@pytest.mark.gpu
@pytest.mark.parametrize("track_name,N,R,mode,flags", [("Austin", 96, 16, 0, 7), ("Silverstone", 70, 64, 0, 1),
                                                       ("Monza", 64, 32, 1, 3), ("Spa", 50, 5, 0, 5)])
def test_device_auto_reset_bit_exact(gpu, oracle, track_name, N, R, mode, flags):
    """Continuous stepping with host-written actions: crashed agents are re-placed by the step kernel itself."""
    t, dev, orc = make_pair(gpu, oracle, track_name, N, R)
    rng = np.random.default_rng(17)
    for env in (dev, orc):
        env.reset_random(None, RANDOM_POINT | RANDOM_LANE, 2, 0, 0)
        env.set(oracle.F_MODE, np.full(N, mode, dtype=np.uint8))
        env.set_auto_reset(True, flags, 77, 5000)
    restarts, crashed_before = 0, np.zeros(N, dtype=bool)
    for it in range(240):
        thr = rng.uniform(30, 100, N).astype(f32) if mode == 0 else rng.uniform(-0.3, 0.6, N).astype(f32)
        steer = rng.uniform(-5, 5, N).astype(f32)
        n = 1 if it % 7 else 3  # a few multi-step launches: a reset agent keeps the zeroed action inside the launch
        for env in (dev, orc):
            env.set(oracle.F_THR, thr)
            env.set(oracle.F_STEER, steer)
            env.step(n)
        if it % 20 == 0 or it == 239:
            o = assert_same(dev, orc, "iteration %d" % it)
            restarts += int((crashed_before & (o["crashed"] == 0)).sum())
            crashed_before = o["crashed"].astype(bool)
    assert dev.step_count == orc.step_count == 240 + 2 * 35
    assert restarts > 0
    # turning it off leaves crashed agents crashed again
    for env in (dev, orc):
        env.set_auto_reset(False)
        env.set(oracle.F_CRASHED, np.ones(N, dtype=np.uint8))
        env.step(2)
    o = assert_same(dev, orc, "auto-reset off")
    assert o["crashed"].all()


@pytest.mark.gpu
def test_device_auto_reset_with_fused_policy(gpu, oracle):
    N, R = 64, 32
    t, dev, orc = make_pair(gpu, oracle, "Monza", N, R)
    dev.policy_mlp_create(30, 1234, 0)
    ga = oracle.OracleGA(orc, 30, 1234, 0)
    for env in (dev, orc):
        env.reset_random(None, RANDOM_POINT, 9, 0, 0)
        env.set(oracle.F_MODE, np.ones(N, dtype=np.uint8))
        env.set_auto_reset(True, RANDOM_POINT | RANDOM_HEADING, 5, 0)
        env.step(1)
    for chunk in range(4):
        dev.rollout_policy(150)
        ga.rollout_policy(150)
        assert_same(dev, orc, "chunk %d" % chunk)
    assert dev.step_count == 601


@pytest.mark.gpu
def test_device_reset_preconditions(gpu):
    t = gpu.Track("Austin")
    fan = gpu.default_ray_fan(5)
    env = gpu.BatchedEnvironment(t.segments, 8, fan)
    with pytest.raises(gpu.capi.OkenvError) as e:
        env.reset_random(None, RANDOM_POINT, 0, 0, 0)
    assert e.value.code == -5 and "okenv_set_centerline" in str(e.value)
    env.set_centerline(t.x, t.y, t.heading)
    env.reset_random(None, RANDOM_POINT, 0, 0, 0)
    with pytest.raises(gpu.capi.OkenvError) as e:
        env.set_auto_reset(True, RANDOM_POINT | RANDOM_LANE, 0, 0)
    assert e.value.code == -5 and "okenv_set_lane_bounds" in str(e.value)
    env.set_lane_bounds(t.li, t.ri)
    env.set_auto_reset(True, RANDOM_POINT | RANDOM_LANE, 0, 0)
    env.step(3)
    assert env.step_count == 3
