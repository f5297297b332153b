"""Two ranks of the benchmark on ONE GPU box (gloo as the process-group backend, both ranks on device 0): the closest
rehearsal of the multi-GPU path a single-GPU machine allows.  What runs is the product itself -- BatchedEnvironment per
rank, global agent ids, EvolutionaryRacer.run_generation with the per-generation fitness all-gather, the barriers and the
max-over-ranks timing of bench.py.  RCCL over xGMI needs the 8-GPU node and is the driver's to run."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def torchrun_bench(*args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--single-device"] + list(args), capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]  # rank 0 prints ONE JSON line
    return json.loads(lines[0])


def plain_bench(*args):
    """`python bench.py --gpus 2 ...` WITHOUT torchrun: the script starts its two ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--single-device"] + list(args),
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    assert "starting the ranks" in r.stderr
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[-2000:]  # nothing but the line (gloo's chatter went to stderr)
    return json.loads(lines[0])


def test_plain_start_honours_gpus(gpu):
    j = plain_bench("--steps", "20", "--warmup", "5", "--repeats", "3", "--no-cpu-baseline", "--headline-only")
    assert j["n_gpus"] == 2 and j["config"]["global_agents"] == 8192 and j["config"]["agent_base_per_rank"] == [0, 4096]


def test_wrong_world_size_fails_loudly(gpu):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "4", "--dist-backend", "gloo",
                        "--single-device", "--steps", "5", "--warmup", "1"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and "--gpus 4 but WORLD_SIZE=2" in r.stderr
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_headline_config_on_two_ranks(gpu):
    j = torchrun_bench("--steps", "20", "--warmup", "5", "--repeats", "5", "--no-cpu-baseline", "--headline-only")
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 20 and j["repeats"] == 5
    assert j["config"]["global_agents"] == 2 * j["config"]["agents_per_gpu"] == 8192
    assert j["config"]["agent_base_per_rank"] == [0, 4096]
    assert j["value"] > 1e7
    # counters come from the committed profile only while it is the current kernel's (tests/test_bench_profile_hash.py)
    assert (j["roofline"]["traffic"] is not None) != bool(j["counter_profile_note"])


def test_island_populations_with_fitness_all_gather_on_two_ranks(gpu):
    """C4 on two ranks: every island's fitness vector is written into a DEVICE tensor (okenv_ga_scores -> ga_scores_into) and
    all-gathered from there (under gloo the collective itself is staged through a pinned host copy, sharding.all_gather_fitness);
    the colony statistics must equal those of the same two islands run one after the other in this process."""
    import torch
    from openkitchen_amd.evolution import EvolutionaryRacer

    j = plain_bench("--config", "c4", "--generations", "2")
    assert j["n_gpus"] == 2 and "all-gather" in j["config"]["workload"]
    ok = gpu
    track = ok.Track("Spa")
    N, R, seed = 8192, 32, 1234
    per_island = []
    for rank in range(2):  # what bench_evolution builds on rank 0 and rank 1
        env = ok.BatchedEnvironment.from_track(track, N, R, device=0)
        ga = EvolutionaryRacer(env, track, hidden=30, seed=seed + rank, agent_base=rank * N, max_steps=4000, steps_per_launch=100,
                               device=torch.device("cuda", 0))
        scores = []
        for _ in range(3):  # the warm-up generation + two timed ones
            rec = ga.run_generation()
            scores.append((rec, ga._fitness.clone()))
        per_island.append(scores)
        env.close()
    for g in range(2):
        rec0, f0 = per_island[0][g + 1]
        rec1, f1 = per_island[1][g + 1]
        colony = torch.stack([f0, f1])  # [world, N], the all-gather's layout
        got = j["generations"][g]
        assert got["steps"] == rec0["steps"] and got["island_best"] == rec0["island_best"] and got["island_mean"] == rec0["island_mean"]
        assert got["colony_best"] == float(colony.max()) and got["colony_mean"] == float(colony.mean())
        assert got["colony_best"] >= got["island_best"] > 0
