"""The CMA-ES candidates' controllers on the device (okenv_controller_*, CovarianceMatrixAdaptationEvolution/Controller.cpp +
CmaEsAgent::updateAction): bit-equal to the oracle's restatement, close to the same network evaluated by PyTorch, and a whole
generation loop (controller -> Environment::step -> fitness bookkeeping) against the oracle."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RAYS = np.array([-70, -30, 0, 30, 70], dtype=np.float32)


def make(gpu, oracle, N, fan, track="Silverstone", seed=0):
    t = gpu.Track(track)
    dev = gpu.BatchedEnvironment.from_track(t, N, ray_angles_deg=fan)
    orc = oracle.OracleEnv(t.segments, N, fan.size, fan, (t.x, t.y, t.heading))
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, t.P, N)
    for e in (dev, orc):
        e.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    dev.step(1)
    orc.step(1)   # initial observation
    return t, dev, orc, rng


@pytest.mark.parametrize("N,R,hidden", [(300, 5, 16), (64, 15, 32), (33, 9, 6), (20, 64, 64)])
def test_controller_actions_match_oracle(gpu, oracle, N, R, hidden):
    fan = RAYS if R == 5 else gpu.default_ray_fan(R)
    t, dev, orc, rng = make(gpu, oracle, N, fan, seed=hidden)
    n_params = dev.controller_create(hidden)
    assert n_params == hidden * R + hidden + (hidden // 2) * hidden + hidden // 2 + 2 * (hidden // 2) + 2
    for it in range(3):
        params = rng.normal(0, 0.8, (N, n_params)).astype(np.float32)
        dev.controller_set_params(params)
        for step in range(30):
            dev.controller_act(100.0, 5.0)
            assert oracle.lib().oracle_env_controller_act(orc.h, params, hidden, 100.0, 5.0) == 0
            if step % 10 == 0:
                assert np.array_equal(dev.get(gpu.capi.F_STEER).view(np.uint32), orc.get(oracle.F_STEER).view(np.uint32)), (it, step)
                assert np.array_equal(dev.get(gpu.capi.F_THROTTLE), orc.get(oracle.F_THR))
            dev.step(1)
            orc.step(1)
        d, o = dev.snapshot(), orc.snapshot()
        for k in ("pos_x", "pos_y", "rot", "crashed", "dist"):
            assert np.array_equal(np.ascontiguousarray(d[k]).view(np.uint8), np.ascontiguousarray(o[k]).view(np.uint8)), (it, k)
    dev.close()


def test_controller_close_to_the_torch_module(gpu, oracle):
    """Same parameters through torch.nn.Linear + tanh (what the reference runs): equal to a few ulps."""
    N, R, hidden = 256, 5, 16
    t, dev, orc, rng = make(gpu, oracle, N, RAYS, seed=5)
    n_params = dev.controller_create(hidden)
    params = rng.normal(0, 0.7, (N, n_params)).astype(np.float32)
    dev.controller_set_params(torch.from_numpy(params).cuda())   # a device tensor works as well
    dev.controller_act(100.0, 5.0)
    steer = dev.get(gpu.capi.F_STEER)
    x = torch.from_numpy(np.asarray(dev.distances(), dtype=np.float32).reshape(N, R) / 200.0)
    want = np.zeros(N, dtype=np.float32)
    for a in range(N):
        p, off, h = torch.from_numpy(params[a]), 0, x[a]
        for o, i in ((hidden, R), (hidden // 2, hidden), (2, hidden // 2)):
            w = p[off:off + o * i].view(o, i)
            off += o * i
            b = p[off:off + o]
            off += o
            h = torch.tanh(torch.nn.functional.linear(h, w, b))
        want[a] = float(h[0]) * 5.0
    assert np.abs(steer - want).max() < 5e-6
    dev.close()


@pytest.mark.parametrize("rollout", [False, True])
def test_cmaes_generation_against_the_oracle_running_the_same_controllers(gpu, oracle, rollout):
    """main_eigen.cpp:113-171 for 96 candidates: the device runs controller -> Environment::step -> fitness bookkeeping as a
    replayed HIP graph of the three calls (rollout=False) or fused into the step kernel as an episode (rollout=True); the oracle
    runs the same loop with its own controller, step and bookkeeping.  Fitness and flags agree bit for bit."""
    from openkitchen_amd.cmaes import CmaEsRacers
    N = 96
    racers = CmaEsRacers("Austin", N, seed=3, max_steps=640, rollout=rollout)
    assert racers.fused and racers.rollout == rollout
    seen, inner = [], racers.set_params

    def spy(population):
        seen.append(population.detach().cpu().numpy().copy() if torch.is_tensor(population) else np.array(population, dtype=np.float32))
        inner(population)

    racers.set_params = spy
    best, steps = racers.run_generation(check_every=16, use_graph=True)
    venv = racers.venv
    fitness = venv.fitness.cpu().numpy()
    assert best == fitness.max() and best > 0 and steps >= 16
    t, fan = venv.track, venv.env.ray_angles_deg
    orc = oracle.OracleEnv(t.segments, N, fan.size, fan, (t.x, t.y, t.heading))
    orc.reset_random(None, 0, 3, 0, 0)
    orc.step(1)
    orc.tracker_create(1)
    orc.tracker_begin()
    params = np.ascontiguousarray(seen[0], dtype=np.float32)
    for _ in range(steps):
        assert oracle.lib().oracle_env_controller_act(orc.h, params, 16, 100.0, 5.0) == 0
        orc.step(1)
        orc.tracker_update()
    assert np.array_equal(fitness.view(np.uint32), orc.tracker_snapshot()["fitness"].view(np.uint32))
    assert np.array_equal(venv.done.cpu().numpy(), orc.get(oracle.F_CRASHED).astype(bool))
    assert np.array_equal(venv.env.get(gpu.capi.F_POS_X).view(np.uint32), orc.get(oracle.F_POS_X).view(np.uint32))


def test_fused_and_torch_controllers_learn_alike(gpu):
    """Not bit-equal (torch's tanh and summation order are its own), but the same algorithm: after a few generations both
    variants have candidates that get well past the start."""
    from openkitchen_amd.cmaes import CmaEsRacers
    best = {}
    for fused in (True, False):
        racers = CmaEsRacers("Silverstone", 128, seed=5, max_steps=300, fused=fused)
        best[fused] = max(racers.run_generation()[0] for _ in range(4))
    assert best[True] > 30 and best[False] > 30
