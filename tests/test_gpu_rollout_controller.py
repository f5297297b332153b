"""okenv_rollout_controller: the CMA-ES racers' inner loop (main_eigen.cpp:135-160 -- CmaEsAgent::updateAction, Environment::step,
fitness bookkeeping) fused into the step kernel, against the oracle running the same loop one call at a time: every agent field,
every ray, the actions and every bookkeeping field, bit for bit; in chunks of any length, as an episode with the reference loop's
own length, and with the device-side resetAgent switched on."""
import numpy as np
import pytest

from test_gpu_parity import assert_same_state

pytestmark = pytest.mark.gpu

RAYS = np.array([-70, -30, 0, 30, 70], dtype=np.float32)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def make(gpu, oracle, track, N, fan, hidden, kind, seed):
    t = gpu.Track(track)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, fan.size, fan, (t.x, t.y, t.heading))
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, t.P, N)
    for e in (dev, orc):
        e.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    dev.step(1)
    orc.step(1)  # initial observation
    n_params = dev.controller_create(hidden)
    params = rng.normal(0, 0.8, (N, n_params)).astype(np.float32)
    dev.controller_set_params(params)
    for e in (dev, orc):
        e.tracker_create(kind)
        e.tracker_begin()
    return t, dev, orc, params, rng


def oracle_iteration(oracle, orc, params, hidden, throttle=100.0, scale=5.0):
    assert oracle.lib().oracle_env_controller_act(orc.h, params, hidden, throttle, scale) == 0
    orc.step(1)
    orc.tracker_update()


def same_everything(gpu, oracle, dev, orc, where):
    assert_same_state(dev.snapshot(), orc.snapshot(), where)
    for f_dev, f_orc in ((gpu.capi.F_THROTTLE, oracle.F_THR), (gpu.capi.F_STEER, oracle.F_STEER)):
        assert np.array_equal(bits(dev.get(f_dev)), bits(orc.get(f_orc))), (where, "action")
    d, o = dev.tracker_snapshot(), orc.tracker_snapshot()
    for k in o:
        assert np.array_equal(bits(d[k]), bits(o[k])), (where, k)


@pytest.mark.parametrize("track,N,R,hidden,kind", [("Silverstone", 300, 5, 16, 1), ("Monza", 64, 15, 32, 1), ("Austin", 33, 9, 6, 0),
                                                   ("Spa", 40, 64, 64, 1), ("Silverstone", 1100, 5, 16, 1)])
def test_rollout_controller_equals_the_three_calls(gpu, oracle, track, N, R, hidden, kind):
    fan = RAYS if R == 5 else gpu.default_ray_fan(R)
    t, dev, orc, params, rng = make(gpu, oracle, track, N, fan, hidden, kind, seed=hidden + R)
    done = 0
    for n in (1, 7, 30, 110, 70, 45):  # crashes from the first dozens of steps on, standstill timeouts at step 201
        dev.rollout_controller(n, 100.0, 5.0)
        for _ in range(n):
            oracle_iteration(oracle, orc, params, hidden)
        done += n
        same_everything(gpu, oracle, dev, orc, "%s N=%d R=%d hidden=%d after %d steps" % (track, N, R, hidden, done))
    crashed = dev.get(gpu.capi.F_CRASHED)
    assert crashed.any() and dev.tracker_snapshot()["fitness"].max() > 0
    # ... and the three separate calls go on from the fused rollout's state (and back)
    for _ in range(5):
        dev.controller_act(100.0, 5.0)
        dev.step(1)
        dev.tracker_update()
        oracle_iteration(oracle, orc, params, hidden)
    dev.rollout_controller(9, 100.0, 5.0)
    for _ in range(9):
        oracle_iteration(oracle, orc, params, hidden)
    same_everything(gpu, oracle, dev, orc, "mixed")
    dev.close()


@pytest.mark.parametrize("track,N,spl", [("Austin", 96, 16), ("Silverstone", 700, 50), ("Monza", 40, 3)])
def test_rollout_controller_episode_equals_reference_loop(gpu, oracle, track, N, spl):
    """main_eigen.cpp:135-160 as an episode: launches cover the agents that can still change, okenv_episode_end returns the step in
    which the last candidate crashed and leaves every candidate as the per-step loop does."""
    hidden, cap = 16, 1500
    t, dev, orc, params, rng = make(gpu, oracle, track, N, RAYS, hidden, 1, seed=N)
    T = 0
    while T < cap:  # the reference's loop
        oracle_iteration(oracle, orc, params, hidden)
        T += 1
        if orc.get(oracle.F_CRASHED).all():
            break
    count0 = dev.step_count
    dev.episode_begin()
    assert dev.episode_tail_limit() == 0
    taken, listed_min = 0, N
    while taken < cap:
        n = min(spl, cap - taken)
        dev.rollout_controller(n, 100.0, 5.0)
        taken += n
        alive, listed = dev.episode_compact()
        listed_min = min(listed_min, listed)
        if alive == 0:
            break
    steps, live = dev.episode_end()
    assert steps == T and N <= live <= N * T
    assert dev.step_count == count0 + T
    if T < cap:
        assert listed_min < N
    same_everything(gpu, oracle, dev, orc, "%s N=%d spl=%d episode of %d steps" % (track, N, spl, T))
    dev.close()


def test_rollout_controller_with_auto_reset(gpu, oracle):
    """Device-side resetAgent at the start of every step: a re-placed candidate's bookkeeping restarts, as okenv_tracker_update has it."""
    N, hidden = 200, 16
    t, dev, orc, params, rng = make(gpu, oracle, "Silverstone", N, RAYS, hidden, 1, seed=77)
    orc.set_lane_bounds(t.li, t.ri)  # (from_track has given the device its copy)
    for e in (dev, orc):
        e.set_auto_reset(True, 7, 123, 0)
    for n in (40, 200, 60):
        dev.rollout_controller(n, 100.0, 5.0)
        for _ in range(n):
            oracle_iteration(oracle, orc, params, hidden)
        same_everything(gpu, oracle, dev, orc, "auto-reset after another %d steps" % n)
    assert dev.tracker_snapshot()["episode_return"].max() > 0
    dev.close()


def test_rollout_controller_refusals(gpu, oracle):
    t = gpu.Track("Austin")
    dev = gpu.BatchedEnvironment.from_track(t, 16, ray_angles_deg=RAYS)
    with pytest.raises(RuntimeError, match="okenv_controller_create"):
        dev.rollout_controller(1)
    dev.controller_create(64)
    with pytest.raises(RuntimeError, match="okenv_tracker_create"):
        dev.rollout_controller(1)
    dev.tracker_create(0)
    dev.episode_begin()
    if dev.info()["lanes_per_agent"] * 4 >= 64:
        with pytest.raises(RuntimeError, match="OKENV_REWARD_PROGRESS"):  # the +1 reward keeps counting for crashed agents
            dev.rollout_controller(1)
    dev.episode_end()
    dev.tracker_create(1)
    if dev.info()["lanes_per_agent"] * 4 < 64:
        with pytest.raises(RuntimeError, match="too wide"):
            dev.rollout_controller(1)
    dev.close()
    # while a controller episode is running its bookkeeping and parameters are the rollout's: outside calls are refused
    dev = gpu.BatchedEnvironment.from_track(t, 16, ray_angles_deg=RAYS)
    n_params = dev.controller_create(16)
    dev.tracker_create(1)
    params = np.zeros((16, n_params), dtype=np.float32)
    dev.controller_set_params(params)
    dev.step(1)
    dev.tracker_begin()
    dev.episode_begin()
    dev.tracker_update()  # no controller rollout yet: still the caller's loop
    dev.rollout_controller(3)
    for call, name in ((dev.tracker_begin, "okenv_tracker_begin"), (dev.tracker_update, "okenv_tracker_update"),
                       (lambda: dev.controller_set_params(params), "okenv_controller_set_params")):
        with pytest.raises(RuntimeError, match=name + ": a controller episode is running"):
            call()
    dev.episode_end()
    dev.tracker_update()
    dev.controller_set_params(params)
    dev.close()
