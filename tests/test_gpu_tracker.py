"""Rollout bookkeeping on the device (okenv_tracker_*) and the callers built on it (CmaEsRacers, collect_episode)
against the oracle's restatement of CovarianceMatrixAdaptationEvolution/main_eigen.cpp:113-171 and
RLRacers/PPO/ppo_sim.cpp:46-92, fed the same actions.  Rewards and fitness are integer-valued floats: bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

f32 = np.float32


def pair(gpu, oracle, track_name, N, R):
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    orc.set_lane_bounds(t.li, t.ri)
    return t, dev, orc


def same_tracker(dev, orc, where):
    d, o = dev.tracker_snapshot(), orc.tracker_snapshot()
    for k in o:
        assert np.array_equal(np.ascontiguousarray(d[k]).view(np.uint32), np.ascontiguousarray(o[k]).view(np.uint32)), (where, k)
    return o


@pytest.mark.parametrize("kind,auto_reset", [(1, False), (1, True), (0, False), (0, True)])
def test_tracker_bit_exact(gpu, oracle, kind, auto_reset):
    N, R = 200, 9
    t, dev, orc = pair(gpu, oracle, "Silverstone", N, R)
    rng = np.random.default_rng(kind * 2 + auto_reset)
    with pytest.raises(gpu.capi.OkenvError):
        dev.tracker_update()
    with pytest.raises(gpu.capi.OkenvError):
        dev.get(gpu.capi.F_FITNESS)
    for env in (dev, orc):
        env.reset_random(None, 3, 4, 0, 0)
        env.set_auto_reset(auto_reset, 7, 9, 0)
        env.tracker_create(kind)
        env.step(1)
        env.tracker_begin()
    same_tracker(dev, orc, "begin")
    finished = 0
    for it in range(500):
        thr = rng.uniform(40, 100, N).astype(f32)
        steer = rng.uniform(-5, 5, N).astype(f32)
        for env in (dev, orc):
            env.set(oracle.F_THR, thr)
            env.set(oracle.F_STEER, steer)
            env.step(1)
            env.tracker_update()
        if it % 50 == 49:
            o = same_tracker(dev, orc, "iteration %d" % it)
            finished = int((o["episode_return"] > 0).sum())
    assert finished > 10
    o = orc.tracker_snapshot()
    if kind == 0 and not auto_reset:
        assert (o["fitness"] == 500).all() and (o["reward"] == 1).all()  # ppo_sim.cpp:77-80: +1 for crashed agents too
    if kind == 1:
        assert o["fitness"].max() > 20 and (o["fitness"] == np.round(o["fitness"])).all()
    if auto_reset:
        assert o["episode_steps"].max() < 500  # episodes restarted


def test_cmaes_generation_fitness_matches_oracle_replay(gpu, oracle):
    from openkitchen_amd.cmaes import CmaEsRacers
    N = 96
    racers = CmaEsRacers("Austin", N, seed=3, max_steps=900, rollout=False)  # the three calls per iteration: env.step is visible
    venv, log = racers.venv, []
    inner = venv.step

    def recording_step(actions=None, n_steps=1):
        log.append((venv.throttle.cpu().numpy().copy(), venv.steering.cpu().numpy().copy()))
        return inner(actions, n_steps)

    venv.step = recording_step
    best, steps = racers.run_generation(check_every=16, use_graph=False)
    assert steps == len(log) and steps >= 16
    fitness = venv.fitness.cpu().numpy()
    assert best == fitness.max() and best > 0
    # main_eigen.cpp:113-171 on the oracle with the recorded actions
    t = venv.track
    fan = venv.env.ray_angles_deg
    orc = oracle.OracleEnv(t.segments, N, fan.size, fan, (t.x, t.y, t.heading))
    orc.reset_random(None, 0, 3, 0, 0)           # env.resetAgent(agent, kResetAgentsRandomly = false)
    orc.step(1)                                  # initial observation
    orc.tracker_create(1)
    orc.tracker_begin()                          # prev_track_idx_
    for thr, steer in log:
        assert (thr == 100).all() and (np.abs(steer) <= 5).all()
        orc.set(oracle.F_THR, thr)
        orc.set(oracle.F_STEER, steer)
        orc.step(1)
        orc.tracker_update()
    o = orc.tracker_snapshot()
    assert np.array_equal(fitness.view(np.uint32), o["fitness"].view(np.uint32))
    assert np.array_equal(venv.done.cpu().numpy(), orc.get(oracle.F_CRASHED).astype(bool))
    if steps < 900:
        assert venv.done.all()
    # the solver consumed that fitness: the mean moved, sigma changed
    assert np.abs(racers.solver.mean).max() > 0 and racers.solver.sigma != 0.5
    venv.step = inner
    best2, _ = racers.run_generation()  # from here on as a replayed HIP graph
    assert racers.generation == 2 and best2 >= 0


def test_cmaes_graph_replay_equals_eager(gpu):
    """The captured iteration (controller forward + env.step + bookkeeping) gives the same fitness as eager launches,
    generation after generation; capturing has no side effect on the simulation state."""
    from openkitchen_amd.cmaes import CmaEsRacers
    a = CmaEsRacers("Monza", 64, seed=11, max_steps=500, rollout=False)
    b = CmaEsRacers("Monza", 64, seed=11, max_steps=500, rollout=False)
    c = CmaEsRacers("Monza", 64, seed=11, max_steps=500)  # the fused rollout, as an episode
    for g in range(3):
        ba, sa = a.run_generation(use_graph=False)
        bb, sb = b.run_generation(use_graph=True)
        bc, sc = c.run_generation()
        assert (ba, sa) == (bb, sb), g
        assert np.array_equal(a.venv.fitness.cpu().numpy(), b.venv.fitness.cpu().numpy()), g
        assert np.array_equal(a.venv.disp_ctr.cpu().numpy(), b.venv.disp_ctr.cpu().numpy()), g
        # ... and as the fused rollout: the same fitness (hence the same next population), the loop's exact length
        assert bc == ba and sa - 16 < sc <= sa, (g, sa, sc)
        assert np.array_equal(a.venv.fitness.cpu().numpy(), c.venv.fitness.cpu().numpy()), g


def test_collect_episode_shapes_and_semantics(gpu):
    from openkitchen_amd.rollout import collect_episode, discounted_returns
    from openkitchen_amd.torch_env import VectorEnvironment
    N = 128
    venv = VectorEnvironment("Monza", N, ray_angles_deg=np.array([-70, -30, 0, 30, 70], dtype=np.float32),
                             auto_reset=False, seed=2, reward="step")
    torch.manual_seed(0)
    actor = torch.nn.Sequential(torch.nn.Linear(5, 128), torch.nn.ReLU(), torch.nn.Linear(128, 3), torch.nn.Softmax(dim=1)).cuda()
    ep = collect_episode(venv, actor, max_steps=1500)
    T = ep["states"].shape[0]
    assert ep["states"].shape == (T, N, 5) and ep["actions"].shape == (T, N) and ep["rewards"].shape == (T, N)
    assert (ep["rewards"] == 1).all()
    assert ep["alive"][0].sum() >= N - 5 and (ep["alive"].sum(dim=0) <= T).all()
    # alive is monotone per agent (no resets inside an episode) and crashed agents freeze their observation
    a = ep["alive"].int()
    assert (a[1:] <= a[:-1]).all()
    i = int(torch.argmin(ep["alive"].sum(dim=0)))
    t0 = int(ep["alive"][:, i].sum())
    if t0 + 2 < T:
        assert torch.equal(ep["states"][t0 + 1, i], ep["states"][t0 + 2, i])
    assert ((ep["states"] >= 0) & (ep["states"] <= 1.0001)).all()  # a 200 px miss can round to 200.00002
    ret = discounted_returns(ep["rewards"])
    assert ret.shape == (T, N) and abs(float(ret.mean())) < 1e-3
    assert (venv.fitness == T).all()
