"""Randomised differential test of the fused caller kernels (EvolutionaryRacer MLP policy + select/mate, tabular
Q-learning) against the CPU oracle (run on the GPU box).  usage: python tests/tools/fuzz_policies.py [seconds] [seed]"""
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
KEYS = ["pos_x", "pos_y", "rot", "speed", "acc", "thr", "steer", "crashed", "timed_out", "disp_ctr", "hit_x", "rel_y", "dist"]
tracks = {n: ok.Track(n) for n in ("Austin", "Silverstone", "Monza", "Spa")}


def same(d, o, what):
    for k in KEYS:
        a, b = np.ascontiguousarray(d[k]), np.ascontiguousarray(o[k])
        if a.dtype == np.float32:
            a, b = a.view(np.uint32), b.view(np.uint32)
        if not np.array_equal(a, b):
            print("MISMATCH in %s: %s" % (k, what), flush=True)
            sys.exit(1)


t0, cases = time.time(), 0
while time.time() - t0 < budget:
    name = rng.choice(list(tracks))
    t = tracks[name]
    N = int(rng.choice([1, 6, 40, 96, 200]))
    R = int(rng.integers(5, 65))
    fan = ok.default_ray_fan(R)
    g = rng.choice([None, None, "64"])
    if g is None:
        os.environ.pop("OKENV_LANES_PER_AGENT", None)
    else:
        os.environ["OKENV_LANES_PER_AGENT"] = g
    dev = ok.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    orc = O.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    seed = int(rng.integers(0, 2 ** 31))
    what = "track %s N %d R %d seed %d lanes %s" % (name, N, R, seed, g)
    if rng.random() < 0.5:
        hidden = int(rng.integers(1, 33))
        what = "GA hidden %d, " % hidden + what
        dev.policy_mlp_create(hidden, seed, 3)
        ga = O.OracleGA(orc, hidden, seed, 3)
        start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
        for env in (dev, orc):
            env.set(O.F_MODE, np.ones(N, dtype=np.uint8))
        for gen in range(2):
            dev.reset_all(*start)
            ga.reset_all(*start)
            dev.step(1)
            orc.step(1)
            for chunk in range(4):
                n = int(rng.integers(20, 200))
                dev.rollout_policy(n)
                ga.rollout_policy(n)
            same(dev.snapshot(), orc.snapshot(), what)
            if not np.array_equal(dev.ga_scores(), ga.scores()):
                print("MISMATCH in scores: " + what)
                sys.exit(1)
            pd, po = dev.ga_select_mate(seed, gen, 3), ga.select_mate(seed, gen, 3)
            if not np.array_equal(pd[: min(5, N)], po[: min(5, N)]) or not np.array_equal(dev.policy_weights().view(np.uint32), ga.weights().view(np.uint32)):
                print("MISMATCH in mating: " + what)
                sys.exit(1)
    else:
        what = "Q, " + what
        dev.q_create()
        oq = O.OracleQ(orc)
        eps, base = float(rng.uniform(0, 1)), 0
        for ep in range(2):
            ridx = int(rng.integers(0, t.P))
            dev.q_begin_episode(ridx)
            oq.begin_episode(ridx)
            for chunk in range(3):
                n = int(rng.integers(10, 120))
                dev.rollout_q(n, eps, seed, 9, base)
                oq.rollout(n, eps, seed, 9, base)
                base += n
            same(dev.snapshot(), orc.snapshot(), what)
            if not np.array_equal(dev.q_table().view(np.uint32), oq.table().view(np.uint32)):
                print("MISMATCH in Q table: " + what)
                sys.exit(1)
    cases += 1
    dev.close()
    if cases % 20 == 0:
        print("%d cases, %.0f s" % (cases, time.time() - t0), flush=True)
print("policy fuzz ok: %d cases in %.0f s" % (cases, time.time() - t0))
