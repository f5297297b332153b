"""Batched Python binding on device tensors (openkitchen_amd/torch_env.py): the tensors are
zero-copy views of the library's buffers, and `step(actions) -> obs, done` driven from torch matches the oracle bit for
bit when both are fed the same actions."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

RANDOM_POINT, RANDOM_LANE, RANDOM_HEADING = 1, 2, 4


def oracle_twin(oracle, venv, flags):
    t = venv.track
    fan = venv.env.ray_angles_deg
    orc = oracle.OracleEnv(t.segments, venv.num_envs, fan.size, fan, (t.x, t.y, t.heading))
    orc.set_lane_bounds(t.li, t.ri)
    orc.reset_random(None, RANDOM_POINT if flags & RANDOM_POINT else 0, venv.seed, 0xFFFFFFFF, venv.agent_base)
    orc.set_auto_reset(True, flags, venv.seed, venv.agent_base)
    return orc


def test_views_are_zero_copy(gpu):
    from openkitchen_amd.torch_env import VectorEnvironment
    venv = VectorEnvironment("Austin", 32, num_rays=5, seed=3)
    for name, f in VectorEnvironment.FIELDS.items():
        address, nbytes = venv.env.field_device_ptr(f)
        t = getattr(venv, name)
        assert t.data_ptr() == address and t.numel() * t.element_size() == nbytes and t.is_cuda
    assert venv.done.dtype == torch.bool and venv.done.data_ptr() == venv.crashed.data_ptr()
    obs, done = venv.step(torch.tensor([[60.0, 0.0]], device="cuda").expand(32, 2))
    assert obs.data_ptr() == venv.distances.data_ptr()
    torch.cuda.synchronize()
    assert np.array_equal(obs.cpu().numpy(), venv.env.distances())
    assert np.array_equal(venv.speed.cpu().numpy(), np.full(32, 60.0, dtype=np.float32))
    # DLPack round trip of a view
    again = torch.from_dlpack(venv.distances)
    assert again.data_ptr() == venv.distances.data_ptr()


@pytest.mark.parametrize("track_name,N,R,flags", [("Silverstone", 256, 15, 1), ("Monza", 100, 5, 7)])
def test_step_from_torch_matches_oracle(gpu, oracle, track_name, N, R, flags):
    from openkitchen_amd.torch_env import VectorEnvironment
    venv = VectorEnvironment(track_name, N, num_rays=R, seed=21, agent_base=1000, randomize_lane=bool(flags & 2),
                             randomize_heading=bool(flags & 4))
    orc = oracle_twin(oracle, venv, flags)
    gen = torch.Generator(device="cuda").manual_seed(5)
    episodes = 0
    obs, done = venv.reset()
    orc.reset_random(None, flags, venv.seed, 0, venv.agent_base)
    orc.step(1)
    for it in range(400):
        # a torch "policy": steer away from the nearer side, full throttle; plus noise -- all on the device
        left, right = obs[:, : R // 2].mean(dim=1), obs[:, R - R // 2:].mean(dim=1)
        steer = torch.clamp((right - left) * 0.05, -2, 2) + (torch.rand(N, device="cuda", generator=gen) - 0.5) * 10
        thr = 40 + 60 * torch.rand(N, device="cuda", generator=gen)
        actions = torch.stack([thr, steer], dim=1)
        obs, done = venv.step(actions)
        a = actions.cpu().numpy()
        orc.set(oracle.F_THR, a[:, 0])
        orc.set(oracle.F_STEER, a[:, 1])
        orc.step(1)
        if it % 25 == 0 or it == 399:
            o = orc.snapshot()
            assert np.array_equal(obs.cpu().numpy().view(np.uint32), o["dist"].view(np.uint32)), it
            assert np.array_equal(done.cpu().numpy(), o["crashed"].astype(bool)), it
            assert np.array_equal(venv.pos_x.cpu().numpy().view(np.uint32), o["pos_x"].view(np.uint32)), it
            assert np.array_equal(venv.timed_out.cpu().numpy(), o["timed_out"]), it
        episodes += int(done.sum())
    assert episodes > 0
    assert np.array_equal(venv.nearest_track_idx().cpu().numpy(),
                          venv.env.nearest_track_idx())


def test_masked_reset_and_scalar_actions(gpu, oracle):
    from openkitchen_amd.torch_env import VectorEnvironment
    venv = VectorEnvironment("Spa", 64, num_rays=5, auto_reset=False, seed=8)
    venv.set_action(30.0, 0.0)
    for _ in range(150):
        _, done = venv.step()
    assert done.any() and not done.all()
    before = venv.pos_x.clone()
    crashed = done.clone()
    venv.reset(mask=crashed)
    torch.cuda.synchronize()
    # (a reset agent may crash again on its first observation step, so `done` is not asserted here)
    moved = venv.pos_x != before
    assert moved[crashed].all()
    assert (venv.throttle[crashed] == 0).all() and (venv.throttle[~crashed] == 30).all()


def test_graph_replay_matches_eager_and_oracle(gpu, oracle):
    """policy + step + bookkeeping captured into one HIP graph: replays advance the device-side step counter (the
    auto-reset epoch) and give the same bits as eager stepping and as the oracle fed the same actions."""
    from openkitchen_amd.torch_env import VectorEnvironment
    N, R, flags = 192, 5, 7
    venv = VectorEnvironment("Austin", N, num_rays=R, seed=33, randomize_lane=True, randomize_heading=True, reward="progress")
    orc = oracle_twin(oracle, venv, flags)
    orc.tracker_create(1)
    venv.reset()
    orc.reset_random(None, flags, venv.seed, 0, venv.agent_base)
    orc.step(1)
    orc.tracker_begin()
    w = torch.randn(R, 2, device="cuda")
    log_thr = torch.zeros(64, N, device="cuda")
    log_steer = torch.zeros(64, N, device="cuda")
    it = torch.zeros((), dtype=torch.long, device="cuda")

    def body():
        out = torch.tanh(venv.observation() @ w)
        thr, steer = 70 + 30 * out[:, 0], 5 * out[:, 1]
        log_thr.index_copy_(0, it.view(1) % 64, thr.unsqueeze(0))
        log_steer.index_copy_(0, it.view(1) % 64, steer.unsqueeze(0))
        it.add_(1)
        venv.step(torch.stack([thr, steer], dim=1))

    def replay_oracle(n):
        torch.cuda.synchronize()
        lt, ls = log_thr.cpu().numpy(), log_steer.cpu().numpy()
        for k in range(n):
            orc.set(oracle.F_THR, lt[k])
            orc.set(oracle.F_STEER, ls[k])
            orc.step(1)
            orc.tracker_update()

    before = venv.pos_x.clone()
    prev_before = venv.prev_crashed.clone()
    # many warm-up iterations, so that agents do crash during them: everything they changed is restored afterwards,
    # including the tracker's memory of crashed_ (an agent crashed on the last warm-up step must not be taken for a
    # re-placed one by the first update after the capture: the oracle comparison below would see its fitness zeroed)
    graph = venv.capture(body, warmup=40)
    assert venv.env.step_count == orc.step_count == 1 and torch.equal(venv.pos_x, before)
    assert torch.equal(venv.prev_crashed, prev_before)
    it.zero_()
    restarts = 0
    for chunk in range(6):
        it.zero_()
        for _ in range(64):
            graph.replay()
        replay_oracle(64)
        o = orc.snapshot()
        assert np.array_equal(venv.distances.cpu().numpy().view(np.uint32), o["dist"].view(np.uint32)), chunk
        assert np.array_equal(venv.pos_x.cpu().numpy().view(np.uint32), o["pos_x"].view(np.uint32)), chunk
        assert np.array_equal(venv.crashed.cpu().numpy(), o["crashed"]), chunk
        ot = orc.tracker_snapshot()
        assert np.array_equal(venv.fitness.cpu().numpy().view(np.uint32), ot["fitness"].view(np.uint32)), chunk
        restarts += int((ot["episode_steps"] < 64).sum())
    assert venv.env.step_count == orc.step_count == 1 + 6 * 64
    assert restarts > 0
