"""Edge cases of the HIP path against the oracle: tiny and ragged populations, one-ray fans, agents off the map or with
non-finite poses, degenerate segment sets, populations beyond the reference's uint16 agent index."""
import numpy as np
import pytest

from test_gpu_parity import assert_same_state, bits, make_pair

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,R", [(1, 1), (1, 64), (3, 2), (17, 7), (5000, 4), (257, 33), (2, 130)])
def test_ragged_shapes(gpu, oracle, N, R):
    t, dev, orc = make_pair(gpu, oracle, "Austin", N, R)
    dev.init_bench_state(0, 0)
    orc.init_bench_state(0, 0)
    steps = 40 if N > 1000 else 150
    dev.rollout_random(steps, 11, 0, 0)
    orc.rollout_random(steps, 11, 0, 0, threads=8 if N > 64 else 1)
    assert_same_state(dev.snapshot(), orc.snapshot(), "N=%d R=%d" % (N, R))


def test_agents_off_the_map_and_non_finite_poses(gpu, oracle):
    N, R = 16, 16
    t, dev, orc = make_pair(gpu, oracle, "Monza", N, R)
    x = np.array([-5000, 5000, 800, 800, 1e6, -1e6, 800, 800, np.nan, 800, np.inf, 800, t.x[5], t.x[50], 0, 1599], dtype=np.float32)
    y = np.array([700, 700, -5000, 5000, 1e6, 1e6, 1e7, -1e7, 700, np.nan, 700, -np.inf, t.y[5], t.y[50], 0, 1399], dtype=np.float32)
    rot = np.array([0, 180, 90, -90, 45, 135, 1e7, -1e7, 0, 0, 0, 0, np.nan, 3.0e38, 7, 11], dtype=np.float32)
    for e in (dev, orc):
        e.reset_agents(np.arange(N), x, y, rot)
    thr = np.full(N, 40.0, dtype=np.float32)
    steer = np.full(N, 1.5, dtype=np.float32)
    dev.set_actions(thr, steer)
    orc.set(oracle.F_THR, thr)
    orc.set(oracle.F_STEER, steer)
    for s in range(30):
        dev.step(1)
        orc.step(1)
    d, o = dev.snapshot(), orc.snapshot()
    for k in ("crashed", "timed_out", "disp_ctr"):
        assert np.array_equal(d[k], o[k]), k
    for k in ("pos_x", "pos_y", "rot", "hit_x", "hit_y", "rel_x", "rel_y", "dist"):
        a, b = d[k], o[k]
        same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))  # NaN payloads may differ between libm and the GPU
        assert same.all(), k


def test_degenerate_segment_sets(gpu, oracle):
    rng = np.random.default_rng(4)
    fan = gpu.default_ray_fan(12)
    sets = [
        np.array([[100, 50, 100, 150]], dtype=np.float32),  # one wall
        np.array([[100, 50, 100, 150], [30, 30, 30, 30], [np.nan, 0, 1, 1], [np.inf, 5, 6, 7], [10, 10, 10, 10]], dtype=np.float32),
        np.concatenate([rng.uniform(0, 200, (300, 4)), np.tile([[50, 50, 50.5, 50.5]], (200, 1))]).astype(np.float32),  # crowded cell
        (rng.uniform(0, 150, (64, 4)) + 3.0e4).astype(np.float32),  # far from the origin
    ]
    for segs in sets:
        lo = float(np.nanmin(np.where(np.isfinite(segs), segs, np.nan)))
        hi = float(np.nanmax(np.where(np.isfinite(segs), segs, np.nan)))
        N = 24
        dev = gpu.BatchedEnvironment(segs, N, fan)
        orc = oracle.OracleEnv(segs, N, fan.size, fan)
        x = rng.uniform(lo - 20, hi + 20, N).astype(np.float32)
        y = rng.uniform(lo - 20, hi + 20, N).astype(np.float32)
        rot = rng.uniform(-180, 180, N).astype(np.float32)
        for e in (dev, orc):
            e.reset_agents(np.arange(N), x, y, rot)
        thr = rng.uniform(0, 80, N).astype(np.float32)
        steer = rng.uniform(-4, 4, N).astype(np.float32)
        dev.set_actions(thr, steer)
        orc.set(oracle.F_THR, thr)
        orc.set(oracle.F_STEER, steer)
        dev.step(25)
        orc.step(25)
        assert_same_state(dev.snapshot(), orc.snapshot(), "segment set with %d segments" % segs.shape[0])


def test_population_beyond_uint16(gpu, oracle):
    """The reference's step loop indexes agents with uint16 (Environment.cpp:128); here 70000 agents are one launch."""
    N, R = 70000, 8
    t, dev, orc = make_pair(gpu, oracle, "Silverstone", N, R)
    dev.init_bench_state(0, 0)
    orc.init_bench_state(0, 0)
    dev.rollout_random(12, 5, 0, 0)
    orc.rollout_random(12, 5, 0, 0, threads=16)
    d, o = dev.snapshot(), orc.snapshot()
    for k in ("pos_x", "pos_y", "crashed", "dist"):
        assert np.array_equal(bits(d[k]), bits(o[k])), k


def test_pybind_compat_single_agent(gpu):
    """The `open_kitchen_pybind.Environment` surface (reference Pybind/bindings.cpp:70-78) over the device environment."""
    import openkitchen_amd.pybind_compat as okpy

    env = okpy.Environment(gpu.track_path("Austin"), draw_rays=False, hidden_window=True, seed=5)
    x0, y0, _ = env.pose
    for _ in range(25):
        env.step()
        env.set_action(30.0, 0.0)
    x1, y1, _ = env.pose
    assert (x1 - x0) ** 2 + (y1 - y0) ** 2 > 25.0  # ~0.48 px per step at speed 30
    assert env.distances.shape == (15,) and (env.distances > 0).all() and (env.distances <= 200.0 + 1e-3).all()
    info = env.get_render_target_info()
    assert (info.height, info.width, info.channels) == (1400, 1600, 4) and len(env.get_render_target()) == info.height * info.row_bytes()
    assert isinstance(env.crashed, bool)
