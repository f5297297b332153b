"""The C++ island runner (openkitchen_amd/csrc/apps/genetic_learner_sim.cpp --gpus K; reference loop:
EvolutionaryRacer/genetic_learner_sim.cpp:47-96, the gathered vector: MiscUtils.hpp:64-71): one host
thread + one okenv handle + one stream per island, per generation ncclAllGather of the fitness vector from device memory.

On a one-GPU box: K = 1 over a real one-rank RCCL communicator (no special case in the app) must equal the Python driver
(openkitchen_amd/evolution.py) number for number; two islands are rehearsed on the one device with the gather staged through
host memory (RCCL refuses two ranks on one GPU) -- threads, global agent ids, seeds, the [K][N] layout -- and must equal the two
islands run one after the other by the Python driver.  The GA checkpoint (Network.hpp:29-51, MiscUtils.hpp:52-59) makes a
round trip through the reference's text format."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def exe(gpu):
    from openkitchen_amd import buildlib
    return dict(zip(buildlib.APPS, buildlib.build_apps()))["genetic_learner_sim"]


def run(exe, *args):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([exe] + [str(a) for a in args], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return r.stdout


def read_dump(path, N, R, G):
    raw = np.fromfile(path, dtype=np.uint8)
    per = (R + 2) * 32 + 32 * 8
    rec = 4 + 4 * N + 20 + 8
    assert raw.size == G * rec + 4 * per, (raw.size, G * rec + 4 * per)
    gens = []
    for g in range(G):
        b = raw[g * rec:(g + 1) * rec]
        gens.append({"steps": int(b[:4].view(np.int32)[0]), "scores": b[4:4 + 4 * N].view(np.float32).copy(),
                     "parents": b[4 + 4 * N:4 + 4 * N + 20].view(np.int32).copy(),
                     "colony_best": float(b[4 + 4 * N + 20:].view(np.float32)[0]), "colony_mean": float(b[4 + 4 * N + 20:].view(np.float32)[1])})
    return gens, raw[G * rec:].view(np.float32).copy()


def python_islands(ok, track, N, R, G, seed, K, max_steps=4000, spl=100):
    """The Python driver's islands 0..K-1 (what bench.py --config c4 builds on rank g), one after the other on device 0."""
    import torch
    from openkitchen_amd.evolution import EvolutionaryRacer
    out = []
    for g in range(K):
        env = ok.BatchedEnvironment.from_track(track, N, R, device=0)
        ga = EvolutionaryRacer(env, track, hidden=30, seed=seed + g, agent_base=g * N, max_steps=max_steps, steps_per_launch=spl,
                               device=torch.device("cuda", 0))
        gens = []
        for _ in range(G):
            rec = ga.run_generation()
            gens.append((rec, ga._fitness.cpu().numpy().copy()))
        out.append(gens)
        env.close()
    return out


def test_one_island_over_a_one_rank_rccl_communicator_equals_the_python_driver(gpu, exe, tmp_path):
    """BASELINE config 4's island at full size (8192 x 32 rays, Spa): ncclCommInitAll over one device, ncclAllGather per generation."""
    ok = gpu
    N, R, G, seed = 8192, 32, 2, 1234
    t = ok.Track("Spa")
    dump = str(tmp_path / "island.bin")
    out = run(exe, t.path, "--agents", N, "--rays", R, "--generations", G, "--seed", seed, "--steps-per-launch", 100, "--gpus", 1, "--dump", dump)
    assert out.count("EPISODE") == G
    gens, _ = read_dump(dump, N, R, G)
    want = python_islands(ok, t, N, R, G, seed, 1)[0]
    for g in range(G):
        rec, fit = want[g]
        assert gens[g]["steps"] == rec["steps"], g
        assert np.array_equal(gens[g]["scores"].view(np.uint32), fit.view(np.uint32)), g
        assert list(gens[g]["parents"]) == rec["parents"], g
        assert gens[g]["colony_best"] == rec["colony_best"] == rec["island_best"]
        assert gens[g]["colony_mean"] == rec["colony_mean"]  # integer-valued scores, power-of-two count: any summation order is exact


def test_two_islands_rehearsed_on_one_device_equal_the_python_islands(gpu, exe, tmp_path):
    ok = gpu
    N, R, G, seed, K = 512, 32, 3, 99, 2
    t = ok.Track("Spa")
    dump = str(tmp_path / "colony.bin")
    out = run(exe, t.path, "--agents", N, "--rays", R, "--generations", G, "--seed", seed, "--max-steps", 1500, "--gpus", K, "--devices", "0,0",
              "--gather", "host", "--dump", dump)
    assert out.count("EPISODE") == G * K and "all 2 islands" in out
    want = python_islands(ok, t, N, R, G, seed, K, max_steps=1500)
    dumps = [read_dump("%s.island%d" % (dump, g), N, R, G)[0] for g in range(K)]
    for gen in range(G):
        colony = np.stack([want[g][gen][1] for g in range(K)])  # [K][N], the all-gather's layout
        for g in range(K):
            rec, fit = want[g][gen]
            d = dumps[g][gen]
            assert d["steps"] == rec["steps"] and list(d["parents"]) == rec["parents"], (gen, g)
            assert np.array_equal(d["scores"].view(np.uint32), fit.view(np.uint32)), (gen, g)  # island g's row of the matrix
            assert d["colony_best"] == float(colony.max()) and d["colony_mean"] == float(np.float32(colony.astype(np.float64).mean()))
    assert not np.array_equal(dumps[0][0]["scores"], dumps[1][0]["scores"])  # different seeds and agent ids: different islands


def test_two_ranks_on_one_device_are_refused_by_rccl_loudly(gpu, exe):
    t = gpu.Track("Spa")
    r = subprocess.run([exe, t.path, "--agents", "64", "--generations", "1", "--gpus", "2", "--devices", "0,0"], capture_output=True, text=True,
                       timeout=300, env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode != 0 and "ncclCommInitAll" in r.stderr and "--gather host" in r.stderr


def test_checkpoint_round_trip_through_the_reference_text_format(gpu, exe, tmp_path):
    ok = gpu
    N, R, H, seed = 256, 15, 30, 5
    t = ok.Track("Monza")
    save = tmp_path / "ckpt"
    save.mkdir()
    dump = str(tmp_path / "a.bin")
    run(exe, t.path, "--agents", N, "--rays", R, "--generations", 3, "--seed", seed, "--max-steps", 1500, "--dump", dump, "--save-dir", save)
    _, best = read_dump(dump, N, R, 3)

    def parse(path):
        lines = open(path).read().splitlines()
        rows, cols = (int(v) for v in lines[0].split())
        m = np.array([[np.float32(float(v)) for v in line.split(" ")] for line in lines[1:]], dtype=np.float32)
        assert m.shape == (rows, cols)
        return m

    w1, w2 = parse(save / "agent_weights_1.txt"), parse(save / "agent_weights_2.txt")
    assert w1.shape == (R + 2, H) and w2.shape == (H, 6)
    b1 = best[:(R + 2) * 32].reshape(R + 2, 32)[:, :H]
    b2 = best[(R + 2) * 32:].reshape(32, 8)[:H, :6]
    six = lambda m: np.array([[np.float32(float("%g" % float(v))) for v in row] for row in m], dtype=np.float32)  # noqa: E731
    assert np.array_equal(w1.view(np.uint32), six(b1).view(np.uint32)) and np.array_equal(w2.view(np.uint32), six(b2).view(np.uint32))
    # kInitFromCheckpoint: every agent starts from the file -> one generation, everybody drives the same
    dump2 = str(tmp_path / "b.bin")
    run(exe, t.path, "--agents", N, "--rays", R, "--generations", 1, "--seed", seed, "--max-steps", 1500, "--dump", dump2, "--init-from", save)
    gens2, best2 = read_dump(dump2, N, R, 1)
    assert (gens2[0]["scores"] == gens2[0]["scores"][0]).all()
    # the same network given to the Python driver's population scores the same
    import torch
    from openkitchen_amd.evolution import EvolutionaryRacer
    env = ok.BatchedEnvironment.from_track(t, N, R, device=0)
    ga = EvolutionaryRacer(env, t, hidden=H, seed=seed, agent_base=0, max_steps=1500, steps_per_launch=50, device=torch.device("cuda", 0))
    block = np.zeros((R + 2) * 32 + 32 * 8, dtype=np.float32)
    block[:(R + 2) * 32].reshape(R + 2, 32)[:, :H] = w1
    block[(R + 2) * 32:].reshape(32, 8)[:H, :6] = w2
    env.set_policy_weights(np.tile(block, (N, 1)))
    rec = ga.run_generation()
    assert rec["steps"] == gens2[0]["steps"] and rec["island_best"] == float(gens2[0]["scores"][0])
    assert np.array_equal(best2.view(np.uint32), block.view(np.uint32))  # the all-time best of that run IS the loaded network
    env.close()
