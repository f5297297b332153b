"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bar (BASELINE.json north_star): crash/done flags bit-exact; ray distances and pose within 1e-5.  Because the
kernels and the oracle share one sincos and run the reference's fp32 operation order without FMA
contraction, every field is in fact compared BIT FOR BIT (np.array_equal on the uint32 views), which is
stronger than the stated tolerance.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

FIELDS_EXACT = ["pos_x", "pos_y", "rot", "speed", "acc", "thr", "steer", "crashed", "timed_out", "disp_ctr", "disp_x",
                "disp_y", "disp_to", "hit_x", "hit_y", "rel_x", "rel_y", "dist"]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def assert_same_state(dev, orc, where=""):
    for k in FIELDS_EXACT:
        a, b = bits(dev[k]), bits(orc[k])
        if not np.array_equal(a, b):
            bad = np.argwhere(a != b)
            raise AssertionError("%s field %s differs at %d places, first %s: dev=%r oracle=%r" %
                                 (where, k, bad.shape[0], bad[0], dev[k][tuple(bad[0])], orc[k][tuple(bad[0])]))


def make_pair(gpu, oracle, track_name, N, R, flags=0, grid_cell=0.0, rays=None):
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R) if rays is None else np.asarray(rays, dtype=np.float32)
    dev = gpu.BatchedEnvironment(t.segments, N, fan, flags=flags, grid_cell=grid_cell, centerline=(t.x, t.y, t.heading))
    orc = oracle.OracleEnv(t.segments, N, fan.size, fan, (t.x, t.y, t.heading))
    return t, dev, orc


def test_sincos_bit_exact(gpu, oracle):
    rng = np.random.default_rng(7)
    x = np.concatenate([
        rng.uniform(-10, 10, 200000), rng.uniform(-1e4, 1e4, 200000), rng.uniform(-2e6, 2e6, 100000),
        rng.standard_normal(1000) * 1e-6, np.array([0.0, -0.0, np.pi / 2, np.pi, 1e-30, 1e-40, 3.4e38, -3.4e38, 1e9]),
        np.deg2rad(np.arange(-720, 721, 5, dtype=np.float64)),
    ]).astype(np.float32)
    s_d, c_d = gpu.debug_sincos(x)
    s_o, c_o = np.zeros_like(x), np.zeros_like(x)
    oracle.lib().oracle_sincosf(x, s_o, c_o, x.size)
    assert np.array_equal(bits(s_d), bits(s_o))
    assert np.array_equal(bits(c_d), bits(c_o))


@pytest.mark.parametrize("track_name", ["Silverstone", "Spa"])
@pytest.mark.parametrize("flags", [0, 1])
def test_cast_rays_vs_brute_force(gpu, oracle, track_name, flags):
    """Arbitrary rays through the grid (LDS and global forms) against the oracle's sweep over all segments."""
    t, dev, orc = make_pair(gpu, oracle, track_name, 4, 8, flags=flags)
    rng = np.random.default_rng(11)
    n = 60000
    # a mix: on the centre line, on boundary points exactly, anywhere in the window, far outside the grid
    idx = rng.integers(0, t.P, n)
    ox = t.x[idx] + rng.normal(0, 8, n).astype(np.float32)
    oy = t.y[idx] + rng.normal(0, 8, n).astype(np.float32)
    k = n // 6
    ox[:k], oy[:k] = t.segments[rng.integers(0, t.S, k), 0], t.segments[rng.integers(0, t.S, k), 1]  # near-endpoints
    sel = rng.integers(0, t.S, k)
    ox[k:2 * k], oy[k:2 * k] = t.segments[sel, 0], t.segments[sel, 1]  # exactly on a segment start point
    ox[2 * k:3 * k] = rng.uniform(-300, 1900, k)
    oy[2 * k:3 * k] = rng.uniform(-300, 1700, k)
    ang = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    # rays along the segment they start on (degenerate, near-parallel)
    seg = t.segments[sel]
    ang[k:2 * k] = np.arctan2(seg[:, 3] - seg[:, 1], seg[:, 2] - seg[:, 0]).astype(np.float32)
    ang[:64] = np.float32(0.0)
    ang[64:128] = np.float32(np.pi / 2)
    ox, oy, ang = ox.astype(np.float32), oy.astype(np.float32), ang.astype(np.float32)
    got = dev.debug_cast_rays(ox, oy, ang)
    L = oracle.lib()
    segflat = np.ascontiguousarray(t.segments.reshape(-1))
    want = np.array([L.oracle_cast_ray(float(ox[i]), float(oy[i]), float(ang[i]), segflat, t.S) for i in range(n)], dtype=np.float32)
    assert np.array_equal(bits(got), bits(want)), "first-hit t differs for %d rays" % int((bits(got) != bits(want)).sum())
    assert (want < 200).mean() > 0.5


@pytest.mark.parametrize("track_name,N,R,mode", [("Austin", 64, 16, 0), ("Silverstone", 96, 64, 0), ("Monza", 80, 32, 1),
                                                  ("Spa", 40, 15, 0), ("Austin", 33, 5, 1), ("Silverstone", 12, 100, 0)])
def test_rollout_random_bit_exact(gpu, oracle, track_name, N, R, mode):
    """Bench driver loop (Philox actions + auto reset) for 450 steps, compared every 75 steps, including
    wall crashes, standstill bookkeeping and stale rays."""
    t, dev, orc = make_pair(gpu, oracle, track_name, N, R)
    dev.init_bench_state(0, mode)
    orc.init_bench_state(0, mode)
    crashes = 0
    for chunk in range(6):
        dev.rollout_random(75, 1234, 0, chunk * 75)
        orc.rollout_random(75, 1234, 0, chunk * 75, threads=8)
        d, o = dev.snapshot(), orc.snapshot()
        assert_same_state(d, o, "chunk %d" % chunk)
        crashes += int(o["crashed"].sum())
    if N >= 32:
        assert crashes > 0, "the trajectory must exercise crashes"


def test_long_trajectory_bit_exact(gpu, oracle):
    """BASELINE config 1 (64 agents x 16 rays, Austin) for 3000 steps of the bench driver loop: thousands of crashes and
    resets and many standstill periods later the state is still the oracle's, bit for bit."""
    t, dev, orc = make_pair(gpu, oracle, "Austin", 64, 16)
    dev.init_bench_state(0, 0)
    orc.init_bench_state(0, 0)
    for chunk in range(6):
        dev.rollout_random(500, 4321, 0, chunk * 500)
        orc.rollout_random(500, 4321, 0, chunk * 500, threads=8)
        assert_same_state(dev.snapshot(), orc.snapshot(), "after %d steps" % ((chunk + 1) * 500))


def test_host_actions_step_by_step(gpu, oracle):
    """The Environment::step surface proper: host writes actions, one launch per step; agents crash and are
    left crashed (stale rays), one agent is reset by the host mid-way, standstill timeouts fire."""
    N, R = 48, 16
    t, dev, orc = make_pair(gpu, oracle, "Austin", N, R)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, t.P, N)
    for e in (dev, orc):
        e.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    mode = (np.arange(N) % 2).astype(np.uint8)
    dev.set(gpu.capi.F_MODE, mode)
    orc.set(oracle.F_MODE, mode)
    saw_timeout = False
    for s in range(460):
        thr = rng.uniform(0, 60, N).astype(np.float32)
        thr[:8] = 0.0  # these never move: standstill timeout after 200 ticks
        steer = rng.uniform(-3, 3, N).astype(np.float32)
        dev.set_actions(thr, steer)
        orc.set(oracle.F_THR, thr)
        orc.set(oracle.F_STEER, steer)
        dev.step(1)
        orc.step(1)
        if s == 230:
            for e in (dev, orc):
                e.reset_agents([0, 9], [t.x[3], t.x[40]], [t.y[3], t.y[40]], [t.heading[3], t.heading[40]])
        if s % 23 == 0 or s > 440:
            d, o = dev.snapshot(), orc.snapshot()
            assert_same_state(d, o, "step %d" % s)
            saw_timeout |= bool(o["timed_out"].any())
    assert saw_timeout
    f = dev.flags()
    o = orc.snapshot()
    assert np.array_equal(f & 1, o["crashed"]) and np.array_equal((f >> 1) & 1, o["timed_out"])
    h = dev.hits()
    assert np.array_equal(bits(h[..., 0]), bits(o["rel_x"])) and np.array_equal(bits(h[..., 1]), bits(o["rel_y"]))
    assert np.array_equal(bits(dev.distances()), bits(o["dist"]))


def test_collide_only_and_sensor_offset(gpu, oracle):
    N, R = 20, 15
    t, dev, orc = make_pair(gpu, oracle, "Monza", N, R)
    rng = np.random.default_rng(9)
    idx = rng.integers(0, t.P, N)
    rot = rng.uniform(-180, 180, N).astype(np.float32)
    for e in (dev, orc):
        e.reset_agents(np.arange(N), t.x[idx], t.y[idx], rot)
    dev.set_sensor_offset(4.5)
    oracle.lib().oracle_env_set_sensor_offset(orc.h, 4.5)
    dev.collide()
    orc.collide()
    assert_same_state(dev.snapshot(), orc.snapshot(), "collide")


@pytest.mark.parametrize("flags", [1, 2])
def test_grid_forms_agree(gpu, oracle, flags):
    """Global-memory grid (flag 1) and the reference-style brute-force sweep (flag 2) against the oracle."""
    N, R = 32, 16
    t, dev, orc = make_pair(gpu, oracle, "Silverstone", N, R, flags=flags)
    info = dev.info()
    assert info["grid_in_lds"] == 0
    dev.init_bench_state(0, 0)
    orc.init_bench_state(0, 0)
    dev.rollout_random(120, 99, 0, 0)
    orc.rollout_random(120, 99, 0, 0, threads=8)
    assert_same_state(dev.snapshot(), orc.snapshot(), "flags %d" % flags)


@pytest.mark.parametrize("cell", [6.0, 11.0, 37.0, 400.0])
def test_cell_size_does_not_change_results(gpu, oracle, cell):
    N, R = 24, 32
    t, dev, orc = make_pair(gpu, oracle, "Austin", N, R, grid_cell=cell)
    dev.init_bench_state(0, 0)
    orc.init_bench_state(0, 0)
    dev.rollout_random(150, 5, 0, 0)
    orc.rollout_random(150, 5, 0, 0, threads=8)
    assert_same_state(dev.snapshot(), orc.snapshot(), "cell %g" % cell)


def test_sharded_population_matches_unsharded(gpu, oracle):
    """Agents [0,N) split over two handles with agent_base offsets reproduce the single-handle run bit for bit
    (what one-GPU-per-shard data parallelism relies on)."""
    N, R = 64, 16
    t = gpu.Track("Spa")
    fan = gpu.default_ray_fan(R)
    whole = gpu.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    halves = [gpu.BatchedEnvironment(t.segments, N // 2, fan, centerline=(t.x, t.y, t.heading)) for _ in range(2)]
    whole.init_bench_state(0, 0)
    whole.rollout_random(200, 77, 0, 0)
    w = whole.snapshot()
    for r, e in enumerate(halves):
        e.init_bench_state(r * N // 2, 0)
        e.rollout_random(200, 77, r * N // 2, 0)
        s = e.snapshot()
        for k in FIELDS_EXACT:
            assert np.array_equal(bits(s[k]), bits(w[k][r * N // 2:(r + 1) * N // 2])), k


def test_nearest_track_idx(gpu, oracle):
    t, dev, orc = make_pair(gpu, oracle, "Spa", 16, 8)
    rng = np.random.default_rng(3)
    qx = rng.uniform(300, 1300, 4096).astype(np.float32)
    qy = rng.uniform(0, 1400, 4096).astype(np.float32)
    qx[:t.P], qy[:t.P] = t.x, t.y  # exact centre-line points
    got = dev.nearest_track_idx(qx, qy)
    want = np.zeros(qx.size, dtype=np.int32)
    oracle.lib().oracle_nearest_track_idx(t.x, t.y, t.P, qx, qy, qx.size, want)
    assert np.array_equal(got, want)
    dev.init_bench_state(0, 0)
    pos = dev.snapshot()
    want2 = np.zeros(16, dtype=np.int32)
    oracle.lib().oracle_nearest_track_idx(t.x, t.y, t.P, pos["pos_x"], pos["pos_y"], 16, want2)
    assert np.array_equal(dev.nearest_track_idx(), want2)


def test_full_size_c2_properties(gpu):
    """BASELINE config 2 at full size (4096 x 64, Silverstone): size-independent properties.
    determinism (two runs agree bit for bit), chunking invariance (1x200 steps == 4x50 steps), every
    distance in [0, 200], crashed agents have a ray under sqrt(2) or timed out."""
    t = gpu.Track("Silverstone")
    fan = gpu.default_ray_fan(64)
    a = gpu.BatchedEnvironment(t.segments, 4096, fan, centerline=(t.x, t.y, t.heading))
    b = gpu.BatchedEnvironment(t.segments, 4096, fan, centerline=(t.x, t.y, t.heading))
    a.init_bench_state(0, 0)
    b.init_bench_state(0, 0)
    a.rollout_random(200, 1234, 0, 0)
    for c in range(4):
        b.rollout_random(50, 1234, 0, c * 50)
    sa, sb = a.snapshot(), b.snapshot()
    for k in FIELDS_EXACT:
        assert np.array_equal(bits(sa[k]), bits(sb[k])), k
    d = sa["dist"]
    live = sa["crashed"] == 0
    assert np.all(d[live] >= 0) and np.all(d[live] <= 200.0 + 1e-3)
    crashed = sa["crashed"] == 1
    assert crashed.any()
    near = (sa["rel_x"] ** 2 + sa["rel_y"] ** 2).min(axis=1) < 2.0
    assert np.all(near[crashed] | (sa["timed_out"][crashed] == 1))
