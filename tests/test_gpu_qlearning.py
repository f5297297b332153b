"""RLRacers/Q_Learning on the device (BASELINE config 5 shape) against the oracle's
restatement: epsilon-greedy actions, rewards from centre-line progress, table updates, episode bookkeeping."""
import numpy as np
import pytest

from test_gpu_parity import assert_same_state, bits

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("track_name,N,R", [("Silverstone", 128, 16), ("Austin", 60, 5), ("Spa", 32, 64)])
def test_q_learning_episodes_bit_exact(gpu, oracle, track_name, N, R):
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R) if R != 5 else np.array([-70, -30, 0, 30, 70], dtype=np.float32)  # QAgent.hpp:56-62
    dev = gpu.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    dev.q_create()
    oq = oracle.OracleQ(orc)
    eps = np.float32(0.9)  # QAgent.hpp:26
    step_base = 0
    rng = np.random.default_rng(3)
    reset_idx = 3  # q_racer_sim.cpp:114
    saw_learned = False
    for episode in range(4):
        dev.q_begin_episode(reset_idx)
        oq.begin_episode(reset_idx)
        for chunk in range(8):
            dev.rollout_q(60, float(eps), 77, 0, step_base)
            oq.rollout(60, float(eps), 77, 0, step_base)
            step_base += 60
            assert_same_state(dev.snapshot(), orc.snapshot(), "episode %d chunk %d" % (episode, chunk))
            sd, so = dev.q_state(), oq.state()
            for x, y, name in zip(sd, so, ("state", "action", "prev_idx")):
                assert np.array_equal(x, y), (episode, chunk, name)
            td, to = dev.q_table(), oq.table()
            assert np.array_equal(bits(td), bits(to)), (episode, chunk)
            if dev.alive_count() == 0:
                break
        saw_learned |= bool((to > np.float32(-1e30)).any())
        eps = eps - np.float32(0.05) if eps > np.float32(0.05) else np.float32(0.0)  # q_racer_sim.cpp:194-205
        reset_idx = int(rng.integers(0, t.P))  # pickResetPosition
    assert saw_learned
    table = dev.q_table()
    valid = table > np.float32(-1e30)
    assert valid.any() and (table[valid] >= -1200.0).all()


def test_q_state_rays_for_wide_fans(gpu):
    t = gpu.Track("Austin")
    env = gpu.BatchedEnvironment(t.segments, 4, gpu.default_ray_fan(16), centerline=(t.x, t.y, t.heading))
    env.q_create()
    env.q_begin_episode(3)
    s, a, p = env.q_state()
    assert (s >= 0).all() and (s < 243).all() and (p == p[0]).all()
    small = gpu.BatchedEnvironment(t.segments, 4, gpu.default_ray_fan(3), centerline=(t.x, t.y, t.heading))
    with pytest.raises(gpu.capi.OkenvError):
        small.q_create()


def test_share_cumulative_knowledge(gpu):
    """shareCumulativeKnowledge (q_racer_sim.cpp:24-75): every table becomes the agent-order mean of the valid entries;
    entries nobody has learnt stay invalid.  Checked against a numpy restatement of the reference's loop."""
    t = gpu.Track("Silverstone")
    N = 96
    env = gpu.BatchedEnvironment(t.segments, N, gpu.default_ray_fan(16), centerline=(t.x, t.y, t.heading))
    env.q_create()
    env.q_begin_episode(3)
    env.rollout_q(200, 0.9, 5, 0, 0)
    before = env.q_table().reshape(N, -1)
    invalid = np.float32(np.finfo(np.float32).min)
    want = np.full(before.shape[1], invalid, dtype=np.float32)
    for e in range(before.shape[1]):
        total, cnt = invalid, np.float32(0)
        for a in range(N):
            v = before[a, e]
            if v != invalid:
                total = np.float32(0) if total == invalid else total
                total = np.float32(total + v)
                cnt = np.float32(cnt + 1)
        want[e] = np.float32(total / cnt) if cnt > 0 else total
    sums, counts = env.q_table_sums()
    assert (counts > 0).sum() > 5
    env.q_share_knowledge()
    after = env.q_table().reshape(N, -1)
    assert np.array_equal(after.view(np.uint32), np.tile(want.view(np.uint32), (N, 1)))
    # the two-call form a multi-GPU caller uses around its all-reduce
    env.set_q_table(before.reshape(N, 243, 3))
    s2, c2 = env.q_table_sums()
    assert np.array_equal(s2.view(np.uint32), sums.view(np.uint32)) and np.array_equal(c2, counts)
    env.q_assign_mean(s2, c2)
    assert np.array_equal(env.q_table().reshape(N, -1).view(np.uint32), after.view(np.uint32))
