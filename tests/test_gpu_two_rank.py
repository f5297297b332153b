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


def test_headline_config_on_two_ranks(gpu):
    j = torchrun_bench("--steps", "20", "--warmup", "5", "--repeats", "5", "--no-cpu-baseline", "--headline-only")
    assert j["n_gpus"] == 2 and j["scaling"] == "weak" and j["steps"] == 20 and j["repeats"] == 5
    assert j["config"]["global_agents"] == 2 * j["config"]["agents_per_gpu"] == 8192
    assert j["config"]["agent_base_per_rank"] == [0, 4096]
    assert j["value"] > 1e7 and j["roofline"]["traffic"] is not None


def test_island_populations_with_fitness_all_gather_on_two_ranks(gpu):
    j = torchrun_bench("--config", "c4", "--generations", "1")
    assert j["n_gpus"] == 2 and "all-gather" in j["config"]["workload"]
    g = j["generations"][0]
    assert g["steps"] > 100 and g["colony_best"] >= g["island_best"] > 0
