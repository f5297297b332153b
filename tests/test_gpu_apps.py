"""The two applications the north star names, as C++ programs over the C ABI (openkitchen_amd/csrc/apps/, replacing the
reference's EvolutionaryRacer/genetic_learner_sim.cpp and RLRacers/Q_Learning/q_racer_sim.cpp): built with g++, run on the
GPU, and their dumped results -- per-generation scores and parents, the best agent's weights; per-episode step counts, the
final Q tables and agent states -- compared bit for bit with a replay of the REFERENCE's loops on the CPU oracle, one
Environment::step per iteration (the apps advance many steps per kernel launch; that must not be visible in any result)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def apps(gpu):
    from openkitchen_amd import buildlib
    return dict(zip(buildlib.APPS, buildlib.build_apps()))


def run(exe, *args):
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    return r.stdout


def test_genetic_learner_sim_matches_oracle_replay(gpu, oracle, apps, tmp_path):
    N, R, H, G, seed, spl = 72, 15, 30, 3, 2024, 50
    t = gpu.Track("Monza")
    dump = str(tmp_path / "ga.bin")
    out = run(apps["genetic_learner_sim"], t.path, "--agents", N, "--rays", R, "--hidden", H, "--generations", G, "--seed", seed,
              "--max-steps", 1500, "--steps-per-launch", spl, "--dump", dump)
    assert out.count("EPISODE") == G
    raw = np.fromfile(dump, dtype=np.uint8)
    per = (R + 2) * 32 + 32 * 8
    rec = 4 + 4 * N + 20 + 8
    assert raw.size == G * rec + 4 * per
    # replay on the oracle: genetic_learner_sim.cpp:47-96
    fan = gpu.default_ray_fan(R)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    orc.set(oracle.F_MODE, np.ones(N, dtype=np.uint8))
    ga = oracle.OracleGA(orc, H, seed, 0)
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    best_all_time, best_w = 0.0, None
    for g in range(G):
        blob = raw[g * rec:(g + 1) * rec]
        steps = int(blob[:4].view(np.int32)[0])
        scores = blob[4:4 + 4 * N].view(np.float32)
        parents = blob[4 + 4 * N:4 + 4 * N + 20].view(np.int32)
        colony_best, colony_mean = blob[4 + 4 * N + 20:].view(np.float32)  # one island: the colony is the island
        assert colony_best == scores.max() and colony_mean == np.float32(scores.astype(np.float64).sum() / N)
        ga.reset_all(*start)
        orc.step(1)
        it = 1
        while ga.alive_count() > 0 and it < 1500:  # the reference's loop, one iteration at a time: the app's launches of `spl`
            ga.rollout_policy(1)                    # steps must not show in its results
            it += 1
        assert it == steps, g
        want = ga.scores()
        assert np.array_equal(scores.view(np.uint32), want.view(np.uint32)), g
        if want.max() > best_all_time:
            best_all_time, best_w = float(want.max()), ga.weights()[int(np.argmax(want))].copy()
        assert np.array_equal(parents, ga.select_mate(seed, g)), g
    got_w = raw[G * rec:].view(np.float32)
    assert np.array_equal(got_w.view(np.uint32), best_w.view(np.uint32))
    assert scores.max() >= 3  # somebody got past the start line


@pytest.mark.parametrize("share", [0, 1])
def test_q_racer_sim_matches_oracle_replay(gpu, oracle, apps, tmp_path, share):
    N, R, E, seed, spl = 48, 5, 4, 77, 40
    t = gpu.Track("Austin")
    dump = str(tmp_path / "q.bin")
    out = run(apps["q_racer_sim"], t.path, "--agents", N, "--rays", R, "--episodes", E, "--seed", seed, "--max-steps", 1200,
              "--steps-per-launch", spl, "--share", share, "--dump", dump)
    assert out.count("EPISODE") == E
    raw = np.fromfile(dump, dtype=np.uint8)
    assert raw.size == E * 12 + 4 * N * 243 * 3 + 12 * N
    fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32)
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    oq = oracle.OracleQ(orc)
    eps, steps_total = np.float32(0.9), 0
    for e in range(E):
        rec = raw[e * 12:(e + 1) * 12]
        steps, reset_idx = (int(v) for v in rec[:8].view(np.int32))
        assert rec[8:].view(np.float32)[0] == eps
        assert 0 <= reset_idx < t.P and (e > 0 or reset_idx == 3)
        oq.begin_episode(reset_idx)
        done = 0
        while done < 1200:  # q_racer_sim.cpp:156-190 one iteration at a time: it leaves with the step in which the last agent
            oq.rollout(1, float(eps), seed, 0, steps_total + done)  # crashes, and so must the app whatever --steps-per-launch is
            done += 1
            if oracle.lib().oracle_env_alive_count(orc.h) == 0:
                break
        assert done == steps, e
        steps_total += done
        eps = eps - np.float32(0.05) if eps > np.float32(0.05) else np.float32(0.0)
        if share:  # shareCumulativeKnowledge, agent-order mean of the valid entries (q_racer_sim.cpp:24-75)
            tab = oq.table().reshape(N, -1)
            invalid = np.float32(np.finfo(np.float32).min)
            mean = np.full(tab.shape[1], invalid, dtype=np.float32)
            for k in range(tab.shape[1]):
                v = tab[:, k][tab[:, k] != invalid]
                if v.size:
                    tot = np.float32(0)
                    for x in v:
                        tot = np.float32(tot + x)
                    mean[k] = np.float32(tot / np.float32(v.size))
            oracle.lib().oracle_q_set_table(orc.h, np.ascontiguousarray(np.tile(mean, (N, 1)).reshape(-1)))
    off = E * 12
    table = raw[off:off + 4 * N * 243 * 3].view(np.float32).reshape(N, 243, 3)
    assert np.array_equal(table.view(np.uint32), oq.table().view(np.uint32))
    st = raw[off + 4 * N * 243 * 3:].view(np.int32).reshape(3, N)
    for got, want in zip(st, oq.state()):
        assert np.array_equal(got, want)
    assert (table > np.float32(-1e30)).any()
