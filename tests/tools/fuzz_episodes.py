"""Randomised differential test of EPISODES (okenv_episode_begin / _compact / _end: work follows the live agents, the reference
loop's own length comes back) against the reference's loop replayed on the CPU oracle one Environment::step at a time (run on
the GPU box).  Random track, population, fan, hidden width / Q-learning, launch lengths, step caps, agents crashed before the
episode, tail-kernel limit, lane-group width, phase-1 range.  usage: python tests/tools/fuzz_episodes.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, ".")
sys.path.insert(0, "tests")
import _oracle as O  # noqa: E402
import openkitchen_amd as ok  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
O.build_oracle(with_ref=False)
O.lib().oracle_set_threads(min(16, os.cpu_count() or 1))
KEYS = ["pos_x", "pos_y", "rot", "speed", "acc", "thr", "steer", "crashed", "timed_out", "disp_ctr", "disp_x", "disp_y", "disp_to",
        "hit_x", "hit_y", "rel_x", "rel_y", "dist"]
tracks = {n: ok.Track(n) for n in ("Austin", "Silverstone", "Monza", "Spa")}


def same(d, o, what):
    for k in KEYS:
        a, b = np.ascontiguousarray(d[k]), np.ascontiguousarray(o[k])
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        if not np.array_equal(a, b):
            print("MISMATCH in %s: %s" % (k, what), flush=True)
            sys.exit(1)


t0, cases, steps_total = time.time(), 0, 0
while time.time() - t0 < budget:
    name = str(rng.choice(list(tracks)))
    t = tracks[name]
    u = rng.random()
    kind = "q" if u < 0.35 else ("ctrl" if u < 0.55 else "ga")
    N = int(rng.choice([2, 9, 40, 100, 260, 420]))
    R = int(rng.choice([5, 8, 15, 16, 32] if kind == "q" else ([5, 9, 16, 33, 64] if kind == "ctrl" else [5, 8, 15, 16, 31, 32, 64])))
    fan = ok.default_ray_fan(R)
    spl = int(rng.choice([1, 3, 17, 50, 100, 333]))
    cap = int(rng.choice([40, 300, 1200]))
    env_set = {}
    for key, choices in (("OKENV_TAIL_MAX_AGENTS", [None, None, "0", "5", "64"]), ("OKENV_LANES_PER_AGENT", [None, None, None, "64"]),
                         ("OKENV_PHASE1_RANGE", [None, None, "0", "20"])):
        v = choices[int(rng.integers(0, len(choices)))]
        if v is None or (key == "OKENV_LANES_PER_AGENT" and int(v) < R):
            os.environ.pop(key, None)
        else:
            os.environ[key] = v
            env_set[key] = v
    what = "%s %s N=%d R=%d spl=%d cap=%d %s" % (kind, name, N, R, spl, cap, env_set)
    dev = ok.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    orc = O.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    seed = int(rng.integers(1, 1 << 30))
    pre_crashed = (rng.random(N) < 0.1).astype(np.uint8) if rng.random() < 0.3 else None
    if kind == "ga":
        hidden = int(rng.choice([1, 7, 30, 32]))
        for e in (dev, orc):
            e.set(ok.capi.F_MODE if e is dev else O.F_MODE, np.ones(N, dtype=np.uint8))
        dev.policy_mlp_create(hidden, seed, 0)
        ga = O.OracleGA(orc, hidden, seed, 0)
        for generation in range(2):
            idx = int(rng.integers(0, t.P))
            start = (float(t.x[idx]), float(t.y[idx]), float(t.heading[idx]))
            dev.reset_all(*start); ga.reset_all(*start)
            if pre_crashed is not None:
                dev.set(ok.capi.F_CRASHED, pre_crashed); orc.set(O.F_CRASHED, pre_crashed)
            dev.step(1); orc.step(1)
            it, live = 0, 0
            while it < cap:
                live += ga.alive_count(); ga.rollout_policy(1); it += 1
                if ga.alive_count() == 0:
                    break
            dev.episode_begin()
            taken = 0
            while taken < cap:
                n = min(spl, cap - taken)
                dev.rollout_policy(n); taken += n
                alive, listed = dev.episode_compact()
                if alive == 0:
                    break
            steps, dlive = dev.episode_end()
            if (steps, dlive) != (it, live):
                print("MISMATCH steps/live %s vs oracle %s: %s" % ((steps, dlive), (it, live), what), flush=True); sys.exit(1)
            same(dev.snapshot(), orc.snapshot(), what)
            if not np.array_equal(dev.ga_select_mate(seed, generation), ga.select_mate(seed, generation)):
                print("MISMATCH parents: " + what, flush=True); sys.exit(1)
            steps_total += it
    elif kind == "ctrl":
        # the CMA-ES racers' loop (main_eigen.cpp:135-160): controller, Environment::step, fitness bookkeeping, fused (okenv_rollout_controller)
        G = dev.info()["lanes_per_agent"]
        hidden = int(rng.choice([h for h in (2, 6, 16, 32, 64) if h <= 4 * G]))
        rk = int(rng.integers(0, 2))
        n_params = dev.controller_create(hidden)
        for generation in range(2):
            idx = rng.integers(0, t.P, N)
            for e in (dev, orc):
                e.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
            if pre_crashed is not None:
                dev.set(ok.capi.F_CRASHED, pre_crashed); orc.set(O.F_CRASHED, pre_crashed)
            dev.step(1); orc.step(1)
            params = rng.normal(0, float(rng.choice([0.3, 1.0])), (N, n_params)).astype(np.float32)
            dev.controller_set_params(params)
            for e in (dev, orc):
                e.tracker_create(rk); e.tracker_begin()
            it, live = 0, 0
            while it < cap:
                live += O.lib().oracle_env_alive_count(orc.h)
                O.lib().oracle_env_controller_act(orc.h, params, hidden, 100.0, 5.0); orc.step(1); orc.tracker_update(); it += 1
                if O.lib().oracle_env_alive_count(orc.h) == 0:
                    break
            if rk == 1:
                dev.episode_begin()
                taken = 0
                while taken < cap:
                    n = min(spl, cap - taken)
                    dev.rollout_controller(n, 100.0, 5.0); taken += n
                    alive, listed = dev.episode_compact()
                    if alive == 0:
                        break
                steps, dlive = dev.episode_end()
                if (steps, dlive) != (it, live):
                    print("MISMATCH steps/live %s vs oracle %s: %s" % ((steps, dlive), (it, live), what), flush=True); sys.exit(1)
            else:  # the +1-per-step reward counts for crashed agents too: no episode, launches of any length up to the loop's own
                taken = 0
                while taken < it:
                    n = min(spl, it - taken)
                    dev.rollout_controller(n, 100.0, 5.0); taken += n
            same(dev.snapshot(), orc.snapshot(), what + " hidden=%d reward=%d" % (hidden, rk))
            d, o = dev.tracker_snapshot(), orc.tracker_snapshot()
            for k in o:
                if not np.array_equal(np.ascontiguousarray(d[k]).view(np.uint32), np.ascontiguousarray(o[k]).view(np.uint32)):
                    print("MISMATCH tracker %s: %s hidden=%d reward=%d" % (k, what, hidden, rk), flush=True); sys.exit(1)
            steps_total += it
    else:
        dev.q_create(); oq = O.OracleQ(orc)
        eps, total = np.float32(rng.choice([0.9, 0.5, 0.0])), int(rng.integers(0, 5000))
        for episode in range(2):
            reset_idx = int(rng.integers(0, t.P))
            dev.q_begin_episode(reset_idx); oq.begin_episode(reset_idx)
            it, live = 0, 0
            while it < cap:
                live += O.lib().oracle_env_alive_count(orc.h)
                oq.rollout(1, float(eps), seed, 0, total + it); it += 1
                if O.lib().oracle_env_alive_count(orc.h) == 0:
                    break
            dev.episode_begin()
            taken = 0
            while taken < cap:
                n = min(spl, cap - taken)
                dev.rollout_q(n, float(eps), seed, 0, total + taken); taken += n
                alive, listed = dev.episode_compact()
                if alive == 0:
                    break
            steps, dlive = dev.episode_end()
            if (steps, dlive) != (it, live):
                print("MISMATCH steps/live %s vs oracle %s: %s" % ((steps, dlive), (it, live), what), flush=True); sys.exit(1)
            same(dev.snapshot(), orc.snapshot(), what)
            if not np.array_equal(dev.q_table().view(np.uint32), oq.table().view(np.uint32)):
                print("MISMATCH q table: " + what, flush=True); sys.exit(1)
            for got, want in zip(dev.q_state(), oq.state()):
                if not np.array_equal(got, want):
                    print("MISMATCH q state: " + what, flush=True); sys.exit(1)
            total += steps
            steps_total += it
    dev.close()
    cases += 1
    if cases % 10 == 0:
        print("%d cases, %d reference-loop steps, %.0f s" % (cases, steps_total, time.time() - t0), flush=True)
print("fuzz_episodes OK: %d cases, %d reference-loop steps without a differing bit" % (cases, steps_total), flush=True)
