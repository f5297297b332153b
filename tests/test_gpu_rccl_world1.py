"""The RCCL lines of the multi-GPU path executed on a ONE-rank communicator (the vector gathered is
EvolutionaryRacer/MiscUtils.hpp:64-71's scores).  A communicator of one rank is legal RCCL: `init_process_group("nccl",
device_id=...)`, `barrier(device_ids=...)`, the float64 MAX all-reduce of the region times on a device tensor and
`all_gather_into_tensor` of the fitness vector straight from device memory all run -- the same four lines the 8-GPU C4 run
lives on -- instead of being skipped at world size 1.  What this cannot show is xGMI: that needs the 8-GPU node."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def bench(*args):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + list(args), capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    # stdout is the line and nothing else: RCCL's version banner (it prints one on stdout when a communicator is created) and
    # every other library's chatter went to stderr
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[-2000:]
    return json.loads(lines[0])


def test_c4_island_with_fitness_all_gather_on_a_one_rank_rccl_communicator(gpu):
    plain = bench("--config", "c4", "--generations", "2")
    forced = bench("--config", "c4", "--generations", "2", "--force-dist")
    assert plain["dist"] is None and plain["n_gpus"] == 1
    d = forced["dist"]
    assert d["backend"] == "nccl" and d["world_size"] == 1 and d["forced_at_world_1"] and forced["n_gpus"] == 1
    assert "all_gather_into_tensor" in d["collectives"] and "all-gather" in forced["config"]["workload"]
    assert len(forced["all_gather_us"]) == 2 and all(0 < t < 1e6 for t in forced["all_gather_us"])
    # the collective changes no result: same loop lengths, same scores; with one island the colony IS the island
    for gp, gf in zip(plain["generations"], forced["generations"]):
        for k in ("steps", "live_agent_steps", "island_best", "island_mean", "colony_best", "colony_mean"):
            assert gp[k] == gf[k], k
        assert gf["colony_best"] == gf["island_best"] and gf["colony_mean"] == gf["island_mean"]
    # ... and costs next to nothing (two generations of ~10 ms each: allow the box's scheduling noise, not a slowdown)
    assert forced["value"] > 0.75 * plain["value"], (forced["value"], plain["value"])


def test_headline_config_on_a_one_rank_rccl_communicator(gpu):
    j = bench("--force-dist", "--steps", "20", "--warmup", "5", "--repeats", "10", "--no-cpu-baseline", "--headline-only")
    assert j["dist"]["backend"] == "nccl" and j["dist"]["world_size"] == 1 and j["n_gpus"] == 1
    assert j["config"]["global_agents"] == 4096 and j["value"] > 1e8


@pytest.fixture
def rccl_group():
    import socket

    import torch
    import torch.distributed as dist
    assert not dist.is_initialized()
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    try:
        yield dist
    finally:
        dist.destroy_process_group()


def test_sharding_collectives_run_on_rccl_at_world_one(gpu, rccl_group):
    """The library functions themselves, on device tensors, with a live one-rank RCCL group: no world-size short-circuit."""
    import numpy as np
    import torch
    from openkitchen_amd import sharding
    from openkitchen_amd.evolution import EvolutionaryRacer
    dist = rccl_group
    assert dist.get_backend() == "nccl" and sharding.group_active() and sharding.world() == (0, 1)
    sharding.barrier(device_ids=[0])
    assert sharding.max_over_ranks(1.25, device="cuda") == 1.25
    assert sharding.max_over_ranks_list([3.0, 1.0, 2.0], device="cuda") == [3.0, 1.0, 2.0]
    f = torch.arange(8192, dtype=torch.float32, device="cuda").flip(0)
    g = sharding.all_gather_fitness(f[::2])  # a non-contiguous device tensor is accepted
    assert g.is_cuda and g.shape == (1, 4096) and torch.equal(g[0], f[::2])
    assert g.data_ptr() != f.data_ptr()  # gathered into a buffer of its own by the collective, not an alias
    # the product's generation on top of it: scores leave a device tensor, colony statistics equal the island's
    ok = gpu
    track = ok.Track("Spa")
    env = ok.BatchedEnvironment.from_track(track, 512, 32, device=0)
    ga = EvolutionaryRacer(env, track, hidden=30, seed=7, agent_base=0, max_steps=600, steps_per_launch=100, device=torch.device("cuda", 0))
    rec = ga.run_generation()
    assert rec["colony_best"] == rec["island_best"] == float(ga._fitness.max()) and rec["colony_mean"] == rec["island_mean"]
    assert rec["all_gather_s"] > 0
    env.close()
    # shareCumulativeKnowledge through the all-reduce equals the local form
    from openkitchen_amd.qlearning import QLearningRacers
    tr = ok.Track("Silverstone")
    tabs = []
    for shared_via_group in (True, False):
        env = ok.BatchedEnvironment.from_track(tr, 256, 16, device=0)
        ql = QLearningRacers(env, tr, seed=3, agent_base=0, steps_per_launch=100)
        ql.run_episode()
        if shared_via_group:
            sharding.share_q_knowledge(env)
        else:
            env.q_share_knowledge()
        tabs.append(env.q_table().copy())
        env.close()
    assert np.array_equal(tabs[0].view(np.uint32), tabs[1].view(np.uint32))
