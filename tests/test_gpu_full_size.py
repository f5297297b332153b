"""Size-independent properties at BASELINE.json's full configuration sizes for the caller configurations: C3/C4
(EvolutionaryRacer, 8192 agents x 32 rays on Monza / Spa) and C5 (Q-learning, 16384 agents x 16 rays).  The oracle cannot
run these sizes in seconds, so the checks are: a window of the full population equals a small population created with
that window's global agent ids (which the oracle tests pin bit for bit), launch chunking does not change results, and
repeated runs are deterministic."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint32) if a.dtype == np.float32 else a


def same(a, b, keys, sl=None):
    for k in keys:
        x = a[k] if sl is None else a[k][sl]
        assert np.array_equal(bits(x), bits(b[k])), k


KEYS = ["pos_x", "pos_y", "rot", "speed", "acc", "crashed", "timed_out", "disp_ctr", "dist", "hit_x"]


@pytest.mark.parametrize("track_name", ["Monza", "Spa"])
def test_c3_c4_generation_window_and_chunking(gpu, track_name):
    N, R, W, BASE = 8192, 32, 128, 4096 + 64
    t = gpu.Track(track_name)
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    big = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    big2 = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    win = gpu.BatchedEnvironment.from_track(t, W, num_rays=R)
    for env, base in ((big, 0), (big2, 0), (win, BASE)):
        env.policy_mlp_create(30, 1234, base)
        env.set(gpu.capi.F_MODE, np.ones(env.N, dtype=np.uint8))
        env.reset_all(*start)
        env.step(1)
    assert np.array_equal(bits(big.policy_weights()[BASE:BASE + W]), bits(win.policy_weights()))
    big.rollout_policy(600)
    win.rollout_policy(600)
    for _ in range(6):
        big2.rollout_policy(100)
    a, b, w = big.snapshot(), big2.snapshot(), win.snapshot()
    same(a, b, KEYS)                              # 1 x 600 steps == 6 x 100 steps
    same(a, w, KEYS, slice(BASE, BASE + W))       # a window of the population == that window alone
    assert 0 < a["crashed"].mean() < 1 or a["crashed"].all()
    # scoring and mating: deterministic, offspring 0 is the best parent's clone
    s1, s2 = big.ga_scores(), big2.ga_scores()
    assert np.array_equal(s1, s2) and s1.max() > 3
    w0 = big.policy_weights()
    p1, p2 = big.ga_select_mate(7, 0), big2.ga_select_mate(7, 0)
    assert np.array_equal(p1, p2) and s1[p1[0]] == s1.max()
    w1 = big.policy_weights()
    assert np.array_equal(bits(w1), bits(big2.policy_weights()))
    assert np.array_equal(bits(w1[0]), bits(w0[p1[0]]))


def test_c5_q_learning_window_and_chunking(gpu):
    N, R, W, BASE = 16384, 16, 256, 10000
    t = gpu.Track("Silverstone")
    big = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    big2 = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    win = gpu.BatchedEnvironment.from_track(t, W, num_rays=R)
    for env in (big, big2, win):
        env.q_create()
        env.q_begin_episode(3)
    big.rollout_q(400, 0.9, 77, 0, 0)
    win.rollout_q(400, 0.9, 77, BASE, 0)
    for c in range(4):
        big2.rollout_q(100, 0.9, 77, 0, c * 100)
    a, b, w = big.snapshot(), big2.snapshot(), win.snapshot()
    same(a, b, KEYS)
    same(a, w, KEYS, slice(BASE, BASE + W))
    ta, tb, tw = big.q_table(), big2.q_table(), win.q_table()
    assert np.array_equal(bits(ta), bits(tb))
    assert np.array_equal(bits(ta[BASE:BASE + W]), bits(tw))
    valid = ta > np.float32(-1e30)
    assert valid.any() and (ta[valid] >= -1200.0).all() and (ta[valid] <= 1200.0).all()
    for x, y in zip(big.q_state(), win.q_state()):
        assert np.array_equal(x[BASE:BASE + W], y)
