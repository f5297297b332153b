"""Episodes -- "step everybody until every agent has crashed" (include/okenv.h: okenv_episode_begin / _compact / _end) -- against
the reference's own loop shape replayed on the oracle ONE STEP AT A TIME (genetic_learner_sim.cpp:76-95,
q_racer_sim.cpp:156-190: policy for every agent, crashed ones included; Environment::step; leave after the step in which the last
agent crashes).  Whatever the launches' lengths, the device must end with the oracle's step count, every state field, every Q
table entry and the oracle's number of live agent-steps -- while stepping only the agents that can still change."""
import numpy as np
import pytest

from test_gpu_parity import assert_same_state, bits

pytestmark = pytest.mark.gpu


def make_ga(gpu, oracle, track_name, N, R, hidden=30, seed=4321, flags=0):
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading), flags=flags)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    mode = np.ones(N, dtype=np.uint8)
    dev.set(gpu.capi.F_MODE, mode)
    orc.set(oracle.F_MODE, mode)
    dev.policy_mlp_create(hidden, seed, 0)
    ga = oracle.OracleGA(orc, hidden, seed, 0)
    return t, dev, orc, ga


def oracle_ga_loop(orc, ga, cap):
    """the reference's loop: returns (iterations, live agent-steps)"""
    it, live = 0, 0
    while it < cap:
        live += ga.alive_count()
        ga.rollout_policy(1)
        it += 1
        if ga.alive_count() == 0:
            break
    return it, live


def device_ga_loop(dev, spl, cap):
    dev.episode_begin()
    taken, listed_trace = 0, []
    while taken < cap:
        n = min(spl, cap - taken)
        dev.rollout_policy(n)
        taken += n
        alive, listed = dev.episode_compact()
        listed_trace.append(listed)
        assert alive <= listed <= dev.N
        if alive == 0:
            break
    steps, live = dev.episode_end()
    return steps, live, listed_trace


@pytest.mark.parametrize("track_name,N,R,spl", [("Monza", 96, 32, 50), ("Monza", 96, 32, 7), ("Austin", 40, 15, 1000), ("Silverstone", 24, 64, 33),
                                                ("Spa", 200, 32, 20), ("Monza", 3, 8, 16)])
def test_ga_episode_equals_reference_loop(gpu, oracle, track_name, N, R, spl):
    t, dev, orc, ga = make_ga(gpu, oracle, track_name, N, R)
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    for generation in range(2):
        dev.reset_all(*start)
        ga.reset_all(*start)
        dev.step(1)
        orc.step(1)
        count0 = dev.step_count
        want_steps, want_live = oracle_ga_loop(orc, ga, 1500)
        steps, live, listed = device_ga_loop(dev, spl, 1500)
        assert (steps, live) == (want_steps, want_live), (generation, steps, want_steps, live, want_live)
        assert dev.step_count == count0 + steps
        assert_same_state(dev.snapshot(), orc.snapshot(), "generation %d" % generation)
        assert listed == sorted(listed, reverse=True)  # the list only shrinks
        assert np.array_equal(dev.ga_scores(), ga.scores())
        assert np.array_equal(dev.ga_select_mate(5, generation), ga.select_mate(5, generation))
        assert np.array_equal(bits(dev.policy_weights()), bits(ga.weights()))
    dev.close()


@pytest.mark.parametrize("tail", ["0", "40"])
def test_ga_episode_lists_on_the_cooperative_kernel(gpu, oracle, monkeypatch, tail):
    """Populations this small are stepped by the tail kernel (one agent per workgroup) from the first compaction on; with it
    switched off (OKENV_TAIL_MAX_AGENTS=0) the lists go through the cooperative kernel, with a limit of 40 agents through the
    cooperative kernel first and the tail kernel once the list is that short -- three ways to the same bits."""
    monkeypatch.setenv("OKENV_TAIL_MAX_AGENTS", tail)
    t, dev, orc, ga = make_ga(gpu, oracle, "Monza", 96, 32)
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    for generation in range(2):
        dev.reset_all(*start)
        ga.reset_all(*start)
        dev.step(1)
        orc.step(1)
        want = oracle_ga_loop(orc, ga, 1500)
        steps, live, listed = device_ga_loop(dev, 25, 1500)
        assert (steps, live) == want
        assert_same_state(dev.snapshot(), orc.snapshot(), "generation %d" % generation)
        assert np.array_equal(dev.ga_select_mate(5, generation), ga.select_mate(5, generation))
    dev.close()


def test_one_launch_for_the_whole_tail(gpu, oracle):
    """what the drivers do: once the list is short enough for one agent per workgroup, ask for all the steps that are left in one
    call (okenv_episode_tail_limit) -- the launch ends with the last crash; same step count, same state"""
    t, dev, orc, ga = make_ga(gpu, oracle, "Spa", 200, 32)
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    dev.reset_all(*start)
    ga.reset_all(*start)
    dev.step(1)
    orc.step(1)
    want = oracle_ga_loop(orc, ga, 2500)
    dev.episode_begin()
    tail = dev.episode_tail_limit()
    assert tail >= 256   # (Spa's image leaves room for one workgroup per CU)
    taken, listed, calls = 0, dev.N, 0
    while taken < 2500:
        n = 2500 - taken if listed <= tail else min(40, 2500 - taken)
        dev.rollout_policy(n)
        taken += n
        calls += 1
        alive, listed = dev.episode_compact()
        if alive == 0:
            break
    steps, live = dev.episode_end()
    assert (steps, live) == want and calls == 1
    assert_same_state(dev.snapshot(), orc.snapshot(), "one launch for the tail")
    dev.close()


def test_ga_episode_with_step_cap_and_agents_crashed_before_it_begins(gpu, oracle):
    """the caller's own cap ends the loop with agents alive: every step taken counts and nothing is put back; agents that are
    crashed when the episode begins are asked for an action once, like everybody else in the reference's loop"""
    t, dev, orc, ga = make_ga(gpu, oracle, "Monza", 64, 16)
    rng = np.random.default_rng(5)
    idx = rng.integers(0, t.P, 64)
    rot = t.heading[idx] + rng.uniform(-40, 40, 64).astype(np.float32)
    for e in (dev, orc):
        e.reset_agents(np.arange(64), t.x[idx], t.y[idx], rot)
    crashed = np.zeros(64, dtype=np.uint8)
    crashed[::5] = 1
    dev.set(gpu.capi.F_CRASHED, crashed)
    orc.set(oracle.F_CRASHED, crashed)
    dev.step(1)
    orc.step(1)
    want_steps, want_live = oracle_ga_loop(orc, ga, 60)
    steps, live, _ = device_ga_loop(dev, 25, 60)
    assert ga.alive_count() > 0 and want_steps == 60   # the cap, not the crashes, ended it
    assert (steps, live) == (want_steps, want_live)
    assert_same_state(dev.snapshot(), orc.snapshot(), "capped episode")
    dev.close()


def test_ga_episode_generic_kernel(gpu, oracle):
    """the global-memory grid form (okStepKernel) runs episodes too"""
    t, dev, orc, ga = make_ga(gpu, oracle, "Austin", 48, 15, flags=gpu.capi.FLAG_FORCE_GLOBAL_GRID)
    assert dev.info()["grid_in_lds"] == 0
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    dev.reset_all(*start)
    ga.reset_all(*start)
    dev.step(1)
    orc.step(1)
    want = oracle_ga_loop(orc, ga, 1200)
    steps, live, _ = device_ga_loop(dev, 40, 1200)
    assert (steps, live) == want
    assert_same_state(dev.snapshot(), orc.snapshot(), "generic kernel")
    dev.close()


def test_outside_changes_end_an_episode_and_misuse_is_refused(gpu, oracle):
    t, dev, orc, ga = make_ga(gpu, oracle, "Monza", 32, 16)
    with pytest.raises(gpu.capi.OkenvError):
        dev.episode_compact()          # no episode
    dev.episode_begin()
    dev.rollout_policy(5)
    dev.step(1)                        # a step without the policy: the episode is over, without corrections
    with pytest.raises(gpu.capi.OkenvError):
        dev.episode_end()
    dev.set_auto_reset(True)
    with pytest.raises(gpu.capi.OkenvError):
        dev.episode_begin()            # crashed agents must stay crashed
    dev.close()


def make_q(gpu, oracle, track_name, N, R):
    t = gpu.Track(track_name)
    fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32) if R == 5 else gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    dev.q_create()
    return t, dev, orc, oracle.OracleQ(orc)


@pytest.mark.parametrize("track_name,N,R,spl", [("Austin", 48, 5, 40), ("Silverstone", 64, 16, 9), ("Monza", 20, 16, 500), ("Spa", 200, 16, 25)])
def test_q_episode_equals_reference_loop(gpu, oracle, track_name, N, R, spl):
    t, dev, orc, oq = make_q(gpu, oracle, track_name, N, R)
    seed, eps, total = 31, np.float32(0.9), 0
    rng = np.random.default_rng(N)
    for episode in range(4):
        reset_idx = 3 if episode == 0 else int(rng.integers(0, t.P))
        dev.q_begin_episode(reset_idx)
        oq.begin_episode(reset_idx)
        # the reference's loop on the oracle, one step at a time: crashed agents keep drawing actions and learning -200
        want_steps, want_live = 0, 0
        while want_steps < 1200:
            want_live += oracle.lib().oracle_env_alive_count(orc.h)
            oq.rollout(1, float(eps), seed, 0, total + want_steps)
            want_steps += 1
            if oracle.lib().oracle_env_alive_count(orc.h) == 0:
                break
        dev.episode_begin()
        taken = 0
        while taken < 1200:
            n = min(spl, 1200 - taken)
            dev.rollout_q(n, float(eps), seed, 0, total + taken)
            taken += n
            alive, listed = dev.episode_compact()
            if alive == 0:
                break
        steps, live = dev.episode_end()
        assert (steps, live) == (want_steps, want_live), (episode, steps, want_steps, live, want_live)
        total += steps
        assert_same_state(dev.snapshot(), orc.snapshot(), "episode %d" % episode)
        assert np.array_equal(bits(dev.q_table()), bits(oq.table())), episode
        for got, want in zip(dev.q_state(), oq.state()):
            assert np.array_equal(got, want), episode
        eps = eps - np.float32(0.05) if eps > np.float32(0.05) else np.float32(0.0)
    assert (dev.q_table() > np.float32(-1e30)).any()
    dev.close()


@pytest.mark.parametrize("track_name,N,R,pre", [("Austin", 48, 5, 60), ("Silverstone", 96, 16, 35), ("Spa", 64, 16, 90)])
def test_q_episode_with_agents_crashed_in_earlier_steps_outside_an_episode(gpu, oracle, track_name, N, R, pre):
    """Agents that crash in plain okenv_rollout_q steps BEFORE okenv_episode_begin keep the state before the crash as their
    current state (q_racer_sim.cpp:177) while every later step's next state is discretizeState() of their stale rays (:173) -- in
    general another table row.  The episode settles their -200 updates at its end and must take max Q from that row."""
    t, dev, orc, oq = make_q(gpu, oracle, track_name, N, R)
    seed, eps = 77, np.float32(0.9)
    dev.q_begin_episode(3)
    oq.begin_episode(3)
    for s in range(pre):  # plain per-step calls, no episode: the step kernel itself makes the crashed agents' updates here
        dev.rollout_q(1, float(eps), seed, 0, s)
        oq.rollout(1, float(eps), seed, 0, s)
    crashed = orc.snapshot()["crashed"].astype(bool)
    assert 0 < crashed.sum() < N
    st, _, _ = oq.state()
    d = orc.snapshot()["dist"].reshape(N, R)
    from_stale = np.zeros(N, dtype=np.int64)
    rays = [int(np.argmin(np.abs(gpu.default_ray_fan(R) - a))) for a in (-70, -30, 0, 30, 70)] if R != 5 else list(range(5))
    for i, r in enumerate(rays):
        from_stale += np.where(d[:, r] < 5, 0, np.where(d[:, r] < 10, 1, 2)) * 3 ** i
    assert (from_stale[crashed] != np.asarray(st)[crashed]).any()  # the case the assumption "next state == stored state" gets wrong
    want_steps = 0
    while want_steps < 900:
        oq.rollout(1, float(eps), seed, 0, pre + want_steps)
        want_steps += 1
        if oracle.lib().oracle_env_alive_count(orc.h) == 0:
            break
    dev.episode_begin()
    taken = 0
    while taken < 900:
        dev.rollout_q(40, float(eps), seed, 0, pre + taken)
        taken += 40
        alive, _ = dev.episode_compact()
        if alive == 0:
            break
    steps, _ = dev.episode_end()
    assert steps == want_steps
    assert_same_state(dev.snapshot(), orc.snapshot(), "after the episode")
    assert np.array_equal(bits(dev.q_table()), bits(oq.table()))
    for got, want in zip(dev.q_state(), oq.state()):
        assert np.array_equal(got, want)
    dev.close()


def test_q_episode_short_list_gets_wider_lane_groups(gpu, oracle, monkeypatch):
    monkeypatch.setenv("OKENV_TAIL_MAX_AGENTS", "0")   # (the cooperative kernel's treatment of short lists; the tail kernel is the default)
    _q_short_list(gpu, oracle, monkeypatch)


def test_q_episode_short_list_on_the_tail_kernel(gpu, oracle, monkeypatch):
    _q_short_list(gpu, oracle, monkeypatch)


def _q_short_list(gpu, oracle, monkeypatch):
    """a population created with narrow lane groups (as the 16384 agents of BASELINE config 5 are: four agents to a wave) whose
    list has become short is launched with wider groups and no phase 1; nothing but the time may change"""
    monkeypatch.setenv("OKENV_LANES_PER_AGENT", "16")
    t, dev, orc, oq = make_q(gpu, oracle, "Silverstone", 300, 16)
    monkeypatch.delenv("OKENV_LANES_PER_AGENT")
    assert dev.info()["lanes_per_agent"] == 16
    seed, eps, total = 5, np.float32(0.9), 0
    for episode in range(2):
        reset_idx = 3 if episode == 0 else 700
        dev.q_begin_episode(reset_idx)
        oq.begin_episode(reset_idx)
        want_steps, want_live = 0, 0
        while want_steps < 1000:
            want_live += oracle.lib().oracle_env_alive_count(orc.h)
            oq.rollout(1, float(eps), seed, 0, total + want_steps)
            want_steps += 1
            if oracle.lib().oracle_env_alive_count(orc.h) == 0:
                break
        dev.episode_begin()
        taken = 0
        while taken < 1000:
            dev.rollout_q(30, float(eps), seed, 0, total + taken)   # the first launch is the full population's, narrow groups
            taken += 30
            alive, listed = dev.episode_compact()
            if alive == 0:
                break
        steps, live = dev.episode_end()
        assert (steps, live) == (want_steps, want_live)
        total += steps
        assert_same_state(dev.snapshot(), orc.snapshot(), "episode %d" % episode)
        assert np.array_equal(bits(dev.q_table()), bits(oq.table()))
        eps = eps - np.float32(0.05)
    dev.close()


def test_q_episode_arguments_must_stay_consistent(gpu, oracle):
    t, dev, orc, oq = make_q(gpu, oracle, "Austin", 16, 5)
    dev.q_begin_episode(3)
    dev.episode_begin()
    dev.rollout_q(10, 0.5, 7, 0, 100)
    with pytest.raises(gpu.capi.OkenvError):
        dev.rollout_q(10, 0.5, 7, 0, 100)      # step_base must advance with the steps taken
    with pytest.raises(gpu.capi.OkenvError):
        dev.rollout_q(10, 0.4, 7, 0, 110)      # epsilon is the episode's
    with pytest.raises(gpu.capi.OkenvError):
        dev.rollout_policy(1)                  # no policy, and the episode is Q-learning's
    dev.rollout_q(10, 0.5, 7, 0, 110)
    dev.episode_end()
    dev.close()
