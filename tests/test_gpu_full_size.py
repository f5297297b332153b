"""BASELINE.json's configurations at their FULL sizes, where the machine is full (four waves per SIMD, the wave priorities and the
inter-wave timing live) and the instantiations the benchmark runs are the ones under test.

C2 (4096 agents x 64 rays, Silverstone, the bench driver's loop) is compared with the oracle directly, every field of every
agent, bit for bit: the oracle spreads its agents over the host's threads and needs a few seconds for a few dozen steps.
C3/C4 (EvolutionaryRacer, 8192 x 32 rays) and C5 (Q-learning, 16384 x 16 rays) run whole episodes at full size; the oracle
replays a WINDOW of the population (agents are independent: a window of the population is the population of that window's
global agent ids) one Environment::step at a time for as many steps as the device says the full population's loop took.
Plus the size-independent properties: chunking invariance, determinism, windows against small device populations."""
import os

import numpy as np
import pytest

from test_gpu_parity import FIELDS_EXACT, assert_same_state

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


def window(snapshot, lo, n):
    return {k: v[lo:lo + n] for k, v in snapshot.items()}


def test_c2_full_size_equals_oracle(gpu, oracle):
    """The headline instantiation (okStepCoopKernel<0, false, false, false, 64, true>) with every SIMD holding four waves, against
    the oracle: from the recipe's initial state through the driver's launches (5 warm-up steps, a 20-step region, ...), and
    again from the state 400 steps later (the population spread over the track, crashes, re-placements, standstill counters
    and stale rays in play)."""
    N, R, seed = 4096, 64, 1234
    threads = min(os.cpu_count() or 1, 64)
    t = gpu.Track("Silverstone")
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    info = dev.info()
    assert info["lanes_per_agent"] == 64 and info["grid_in_lds"] == 1 and info["grid_blocks"] * info["block_threads"] == N * 64
    dev.init_bench_state(0, 0)
    orc.init_bench_state(0, 0)
    done = 0
    for n in (5, 20, 10):  # bench.py --steps 20 --warmup 5: one launch per region
        dev.rollout_random(n, seed, 0, done)
        orc.rollout_random(n, seed, 0, done, threads=threads)
        done += n
        assert_same_state(dev.snapshot(), orc.snapshot(), "C2 full size after %d steps" % done)
    for _ in range(4):
        dev.rollout_random(100, seed, 0, done)
        done += 100
    snap = dev.snapshot()
    assert snap["disp_ctr"].max() > 0 and (snap["crashed"] == 1).any() and np.abs(snap["hit_x"]).max() > 0
    for f, name in enumerate(gpu.capi.FIELD_NAMES[:19]):
        orc.set(f, snap[name])
    for n in (20, 5):
        dev.rollout_random(n, seed, 0, done)
        orc.rollout_random(n, seed, 0, done, threads=threads)
        done += n
        assert_same_state(dev.snapshot(), orc.snapshot(), "C2 full size after %d steps" % done)
    dev.close()


@pytest.mark.parametrize("track_name", ["Monza", "Spa"])
def test_c3_c4_full_size_episode_window_equals_oracle(gpu, oracle, track_name):
    """A whole EvolutionaryRacer rollout of 8192 agents as the product runs it (episode: launches of 100 steps over the agents
    that can still change), then a window of 96 agents replayed on the oracle one step at a time for the loop's length."""
    N, R, W, BASE, seed = 8192, 32, 96, 4096 + 64, 1234
    t = gpu.Track(track_name)
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    dev = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    dev.set(gpu.capi.F_MODE, np.ones(N, dtype=np.uint8))
    dev.policy_mlp_create(30, seed, 0)
    fan = gpu.default_ray_fan(R)
    orc = oracle.OracleEnv(t.segments, W, R, fan, (t.x, t.y, t.heading))
    orc.set(oracle.F_MODE, np.ones(W, dtype=np.uint8))
    ga = oracle.OracleGA(orc, 30, seed, BASE)
    assert np.array_equal(dev.policy_weights()[BASE:BASE + W].view(np.uint32), ga.weights().view(np.uint32))
    dev.reset_all(*start)
    ga.reset_all(*start)
    dev.step(1)
    orc.step(1)
    dev.episode_begin()
    taken, listed_min = 0, N
    while taken < 3999:
        n = min(100, 3999 - taken)
        dev.rollout_policy(n)
        taken += n
        alive, listed = dev.episode_compact()
        listed_min = min(listed_min, listed)
        if alive == 0:
            break
    steps, live = dev.episode_end()
    assert 100 < steps <= taken and N < live < N * steps and listed_min < N // 8
    for _ in range(steps):  # the reference's loop: every agent of the window, crashed or not, until the LAST of the 8192 is done
        ga.rollout_policy(1)
    assert_same_state(window(dev.snapshot(), BASE, W), orc.snapshot(), "C3/C4 window after the episode (%d steps)" % steps)
    assert np.array_equal(dev.ga_scores()[BASE:BASE + W], ga.scores())
    dev.close()


def test_c5_full_size_episode_window_equals_oracle(gpu, oracle):
    N, R, W, BASE, seed = 16384, 16, 128, 10000, 77
    t = gpu.Track("Silverstone")
    dev = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    dev.q_create()
    fan = gpu.default_ray_fan(R)
    orc = oracle.OracleEnv(t.segments, W, R, fan, (t.x, t.y, t.heading))
    oq = oracle.OracleQ(orc)
    eps, total = np.float32(0.9), 0
    for episode, reset_idx in enumerate((3, 511)):
        dev.q_begin_episode(reset_idx)
        oq.begin_episode(reset_idx)
        dev.episode_begin()
        taken = 0
        while taken < 4000:
            dev.rollout_q(100, float(eps), seed, 0, total + taken)
            taken += 100
            alive, listed = dev.episode_compact()
            if alive == 0:
                break
        steps, live = dev.episode_end()
        assert 50 < steps <= taken
        for s in range(steps):
            oq.rollout(1, float(eps), seed, BASE, total + s)
        total += steps
        assert_same_state(window(dev.snapshot(), BASE, W), orc.snapshot(), "C5 window after episode %d (%d steps)" % (episode, steps))
        assert np.array_equal(dev.q_table()[BASE:BASE + W].view(np.uint32), oq.table().view(np.uint32))
        for got, want in zip(dev.q_state(), oq.state()):
            assert np.array_equal(got[BASE:BASE + W], want)
        eps = eps - np.float32(0.05)
    dev.close()


def test_c4_island_generation_that_runs_into_the_step_cap(gpu, oracle):
    """BASELINE config 4's island (8192 x 32 rays, Spa, the benchmark's seed) as the benchmark runs it: the second timed generation
    takes all 4000 steps -- not because anybody laps the track but because ONE agent has tunnelled through both boundary polylines
    (the crash test is lidar-only and a step can be 1.6 px long: SURVEY.md appendix A.4, CollisionChecker.cu:167-171) and drives
    on outside at full speed, where nothing is within sensor range and the standstill timeout never fires.  That regime -- thousands
    of steps of an agent off the grid, thousands of pixels away -- is replayed on the oracle for the survivor and the generation's
    best scorers, from the device's state at the generation's start (DisplacementStats carry over from the generations before), one
    Environment::step at a time to the cap, and compared bit for bit."""
    import torch
    from openkitchen_amd.evolution import EvolutionaryRacer
    N, R, seed = 8192, 32, 1234
    t = gpu.Track("Spa")
    env = gpu.BatchedEnvironment.from_track(t, N, R, device=0)
    ga = EvolutionaryRacer(env, t, hidden=30, seed=seed, agent_base=0, max_steps=4000, steps_per_launch=100, device=torch.device("cuda", 0))
    recs = [ga.run_generation() for _ in range(2)]
    assert [r["steps"] for r in recs] == [804, 1005] and all(r["alive_at_end"] == 0 and r["off_grid_alive"] == 0 for r in recs)
    env.reset_all(*ga.start)            # the state generation 2 starts from (its own reset_all repeats this one)
    s0, w0 = env.snapshot(), env.policy_weights().copy()
    rec = ga.run_generation()
    s1 = env.snapshot()
    assert rec["steps"] == 4000 and rec["alive_at_end"] == 1 and rec["off_grid_alive"] == 1 and rec["off_grid_agents"] >= 1
    survivors = np.flatnonzero(s1["crashed"] == 0)
    assert survivors.size == 1
    a = int(survivors[0])
    seg = t.segments.reshape(-1, 4)
    far = max(float(s1["pos_x"][a]) - seg[:, [0, 2]].max(), seg[:, [0, 2]].min() - float(s1["pos_x"][a]),
              float(s1["pos_y"][a]) - seg[:, [1, 3]].max(), seg[:, [1, 3]].min() - float(s1["pos_y"][a]))
    assert far > 1000 and s1["speed"][a] == 100.0      # thousands of pixels beyond the track's box, flat out
    score = ga._fitness.cpu().numpy()
    ids = np.unique(np.concatenate([survivors, np.argsort(-score, kind="stable")[:31], np.arange(8)]))
    K = ids.size
    fan = gpu.default_ray_fan(R)
    orc = oracle.OracleEnv(t.segments, K, R, fan, (t.x, t.y, t.heading))
    names = gpu.capi.FIELD_NAMES
    for f in range(19):
        v = s0[names[f]]
        orc.set(f, v.reshape(N, R)[ids] if f in gpu.capi.PER_RAY else v[ids])
    og = oracle.OracleGA(orc, 30, seed, 0)
    og.set_weights(w0[ids])
    orc.step(1)                          # the initial observation (genetic_learner_sim.cpp:75)
    for _ in range(rec["steps"] - 1):    # the reference's loop to the cap, crashed agents included
        og.rollout_policy(1)
    want = orc.snapshot()
    got = {k: (v.reshape(N, R)[ids] if v.size == N * R else v[ids]) for k, v in s1.items()}
    assert_same_state(got, want, "the survivor and the best scorers after 4000 steps")
    assert np.array_equal(score[ids], og.scores())
    assert (want["crashed"] == 0).sum() == 1
    env.close()


@pytest.mark.parametrize("N,R,track_name", [(10007, 64, "Silverstone"), (20011, 16, "Austin"), (9001, 32, "Monza")])
def test_populations_beyond_one_round_of_workgroups_equal_oracle_windows(gpu, oracle, N, R, track_name):
    """More agents than the machine holds at once (1.1 - 2.4 rounds of workgroups: the later ones queue behind the first) and a last
    workgroup that is only partly filled: windows of the population -- its start, the seam between the first round and the
    second, its very end -- against the oracle stepping those windows' global agent ids (agents are independent; the bench
    driver's draws are keyed by the global id)."""
    seed, W = 4321, 40
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment.from_track(t, N, num_rays=R)
    info = dev.info()
    lanes_per_round = info["compute_units"] * 1024
    assert N * info["lanes_per_agent"] > lanes_per_round and info["grid_blocks"] > info["compute_units"]
    assert (N * info["lanes_per_agent"]) % info["block_threads"] != 0  # the last workgroup is partly filled
    seam = lanes_per_round // info["lanes_per_agent"]                   # first agent of the second round
    bases = [0, seam - W // 2, N - W]
    orcs = [oracle.OracleEnv(t.segments, W, R, fan, (t.x, t.y, t.heading)) for _ in bases]
    dev.init_bench_state(0, 0)
    for o, b in zip(orcs, bases):
        o.init_bench_state(b, 0)
    done = 0
    for n in (1, 37, 100, 20):
        dev.rollout_random(n, seed, 0, done)
        for o, b in zip(orcs, bases):
            o.rollout_random(n, seed, b, done, threads=8)
        done += n
        snap = dev.snapshot()
        for o, b in zip(orcs, bases):
            want = o.snapshot()
            for k in FIELDS_EXACT:
                assert np.array_equal(bits(snap[k][b:b + W]), bits(want[k])), (k, b, done)
    assert (snap["crashed"] == 1).any() and snap["disp_ctr"].max() > 0
    dev.close()
