"""EvolutionaryRacer on the device (BASELINE config 3 shape) against the oracle's
restatement: policy weights, fused MLP policy + step rollouts, scores, parent choice and mating, bit for bit."""
import numpy as np
import pytest

from test_gpu_parity import FIELDS_EXACT, assert_same_state, bits

pytestmark = pytest.mark.gpu


def make(gpu, oracle, track_name, N, R, hidden=30, seed=4321):
    t = gpu.Track(track_name)
    fan = gpu.default_ray_fan(R)
    dev = gpu.BatchedEnvironment(t.segments, N, fan, centerline=(t.x, t.y, t.heading))
    orc = oracle.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    mode = np.ones(N, dtype=np.uint8)  # GeneticAgent: MovementMode::ACCELERATION (GeneticAgent.hpp:28,34)
    dev.set(gpu.capi.F_MODE, mode)
    orc.set(oracle.F_MODE, mode)
    dev.policy_mlp_create(hidden, seed, 0)
    ga = oracle.OracleGA(orc, hidden, seed, 0)
    return t, dev, orc, ga


@pytest.mark.parametrize("track_name,N,R", [("Monza", 96, 32), ("Austin", 40, 15), ("Silverstone", 24, 64)])
def test_generation_loop_bit_exact(gpu, oracle, track_name, N, R):
    t, dev, orc, ga = make(gpu, oracle, track_name, N, R)
    assert np.array_equal(bits(dev.policy_weights()), bits(ga.weights()))
    w0 = dev.policy_weights()
    real = np.abs(w0).sum(axis=0) > 0
    assert real.sum() == (R + 2) * 30 + 30 * 6  # Network.hpp:92-95: (R+2) x 30 and 30 x 6, no biases
    assert w0.min() >= -1.0 and w0.max() < 1.0
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))  # genetic_learner_sim.cpp:34-36
    for generation in range(3):
        dev.reset_all(*start)
        ga.reset_all(*start)
        dev.step(1)  # initial observation (genetic_learner_sim.cpp:75)
        orc.step(1)
        steps = 0
        while steps < 1500:
            dev.rollout_policy(125)
            ga.rollout_policy(125)
            steps += 125
            assert_same_state(dev.snapshot(), orc.snapshot(), "gen %d step %d" % (generation, steps))
            alive = dev.alive_count()
            assert alive == ga.alive_count()
            if alive == 0:
                break
        sd, so = dev.ga_scores(), ga.scores()
        assert np.array_equal(sd, so)
        pd, po = dev.ga_select_mate(99, generation), ga.select_mate(99, generation)
        assert np.array_equal(pd, po)
        assert np.array_equal(bits(dev.policy_weights()), bits(ga.weights()))
        w = dev.policy_weights()
        assert np.array_equal(w[0], w0[pd[0]])  # offspring 0 is a clone of the best (Mating.hpp:129)
        changed = (w[1] != w0[pd[0]])[real].mean()
        assert 0.03 < changed < 0.2  # offspring 1: the best with ~10 % of its weights mutated (Mating.hpp:130,54)
        w0 = w


def test_policy_actions_are_the_reference_decode(gpu, oracle):
    """Thresholded outputs give throttle in {-0.3, 0, +0.3} and steering sums of {+1, +4, -1, -4}
    (GeneticAgent.hpp:20-24,47-53)."""
    t, dev, orc, ga = make(gpu, oracle, "Monza", 64, 32)
    dev.reset_all(float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    dev.step(1)
    dev.rollout_policy(3)
    s = dev.snapshot()
    assert set(np.unique(s["thr"])) <= {np.float32(-0.3), np.float32(0.0), np.float32(0.3)}
    assert set(np.unique(s["steer"])) <= {np.float32(v) for v in (-5, -4, -3, -1, 0, 1, 3, 4, 5)}
    assert len(np.unique(s["steer"])) > 2


def test_policy_needs_supported_fan(gpu):
    t = gpu.Track("Austin")
    env = gpu.BatchedEnvironment(t.segments, 4, gpu.default_ray_fan(3))
    with pytest.raises(gpu.capi.OkenvError):
        env.policy_mlp_create(30, 1, 0)
    with pytest.raises(gpu.capi.OkenvError):
        env.rollout_policy(1)
