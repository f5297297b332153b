"""N>1 path on CPU: two (and four) processes over gloo.  The shard engines are oracle-backed stand-ins (tests only); what is under
test is the product's sharding logic (openkitchen_amd/sharding.py): global agent ids per rank, the fitness all-gather's
layout, max-over-ranks timing -- and that a sharded population reproduces the unsharded one."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import _oracle as O
    from openkitchen_amd import sharding

    class OracleShard:
        """Test-only engine with the BatchedEnvironment methods ShardedPopulation uses."""
        def __init__(self, n, agent_base):
            self.t = O.Track("Austin")
            fan = O.default_ray_fan(8)
            self.env = O.OracleEnv(self.t.segments, n, 8, fan, (self.t.x, self.t.y, self.t.heading))
        def init_bench_state(self, agent_base, mode): self.env.init_bench_state(agent_base, mode)
        def rollout_random(self, n, seed, agent_base, step_base): self.env.rollout_random(n, seed, agent_base, step_base)
        def sync(self): pass
        def nearest_track_idx(self):
            s = self.env.snapshot()
            out = np.zeros(self.env.N, dtype=np.int32)
            O.lib().oracle_nearest_track_idx(self.t.x, self.t.y, self.t.P, s["pos_x"], s["pos_y"], self.env.N, out)
            return out

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    pop = sharding.ShardedPopulation(lambda n, base: OracleShard(n, base), agents_per_rank=12)
    assert (pop.agent_base, pop.n) == (rank * 12, 12)
    elapsed = pop.timed(lambda: pop.rollout(60, seed=99, steps_per_launch=25))
    assert elapsed > 0
    fit = pop.fitness()
    assert fit.shape == (world, 12)
    slow = sharding.max_over_ranks(1.0 + rank)
    assert slow == float(world)
    assert sharding.split_population(10, 3) == [(0, 4), (4, 3), (7, 3)]

    # shareCumulativeKnowledge across ranks: sums and counts are all-reduced, every rank assigns the same means
    class QStub:
        def q_table_sums(self):
            sums = np.full(729, 1.0 + rank, dtype=np.float32); counts = np.full(729, 2.0, dtype=np.float32)
            sums[rank] = np.finfo(np.float32).min; counts[rank] = 0      # an entry this rank has never visited
            sums[700:] = np.finfo(np.float32).min; counts[700:] = 0      # entries nobody has visited
            return sums, counts
        def q_assign_mean(self, sums, counts): self.got = (sums.copy(), counts.copy())
        def q_share_knowledge(self): raise AssertionError("single-process path taken with a process group")
    q = QStub()
    sharding.share_q_knowledge(q)
    sums, counts = q.got
    total = world * (world + 1) / 2.0
    want_s = np.full(729, total, dtype=np.float32); want_c = np.full(729, 2.0 * world, dtype=np.float32)
    for r in range(world):             # entry r: every rank but r contributed
        want_s[r], want_c[r] = total - (1.0 + r), 2.0 * (world - 1)
    want_s[700:], want_c[700:] = np.finfo(np.float32).min, 0.0
    assert np.array_equal(sums, want_s) and np.array_equal(counts, want_c)
    if rank == 0:
        np.save(%(out)r, fit.numpy())
    dist.barrier()
    dist.destroy_process_group()
''')


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_ranks_population_matches_unsharded(oracle, tmp_path, world):
    out = str(tmp_path / "fitness.npy")
    script = tmp_path / "worker.py"
    script.write_text(WORKER % {"root": ROOT, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % world, "--master-addr", "127.0.0.1",
                    "--master-port", str(free_port()), str(script)], check=True, env=env, timeout=600, cwd=ROOT)
    fit = np.load(out)
    assert fit.shape == (world, 12)
    # the same population, unsharded, through the oracle
    n = 12 * world
    t = oracle.Track("Austin")
    env1 = oracle.OracleEnv(t.segments, n, 8, oracle.default_ray_fan(8), (t.x, t.y, t.heading))
    env1.init_bench_state(0, 0)
    env1.rollout_random(60, 99, 0, 0)
    s = env1.snapshot()
    want = np.zeros(n, dtype=np.int32)
    oracle.lib().oracle_nearest_track_idx(t.x, t.y, t.P, s["pos_x"], s["pos_y"], n, want)
    assert np.array_equal(fit.reshape(-1).astype(np.int32), want)


GA_WORKER = textwrap.dedent('''
    import json, os, sys
    import numpy as np
    import torch
    import torch.distributed as dist
    sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests"))
    import _oracle as O
    from openkitchen_amd.evolution import EvolutionaryRacer

    class OracleIsland:
        """Test-only engine with the BatchedEnvironment methods EvolutionaryRacer uses (the product's engine needs a GPU)."""
        def __init__(self, track, n, rays):
            self.N = n
            self.env = O.OracleEnv(track.segments, n, rays, O.default_ray_fan(rays), (track.x, track.y, track.heading))
            self.ga = None
        def set(self, field, arr): self.env.set(field, arr)
        def policy_mlp_create(self, hidden, seed, agent_base): self.ga = O.OracleGA(self.env, hidden, seed, agent_base)
        def reset_all(self, x, y, rot): self.ga.reset_all(x, y, rot)
        def step(self, n): self.env.step(n)
        def alive_count(self): return self.ga.alive_count()
        # episodes (include/okenv.h): the reference's loop, which leaves with the step in which the last agent crashes
        def episode_begin(self): self._taken, self._T, self._live = 0, 0, 0
        def episode_tail_limit(self): return 8   # (short lists: all remaining steps in one call)
        def rollout_policy(self, n):
            for _ in range(n):
                alive = self.ga.alive_count()
                if alive > 0 or self._taken == 0:   # (at least one iteration is always taken)
                    self._live += alive
                    self.ga.rollout_policy(1)
                    self._T = self._taken + 1
                self._taken += 1
        def episode_compact(self): return self.ga.alive_count(), self.ga.alive_count()
        def episode_end(self): return self._T, self._live
        def ga_scores(self): return self.ga.scores()
        def ga_select_mate(self, seed, generation, agent_base): return self.ga.select_mate(seed, generation, agent_base)
        def sync(self): pass

    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    track = O.Track("Monza")
    N = 20
    island = OracleIsland(track, N, 8)
    ga = EvolutionaryRacer(island, track, hidden=30, seed=500 + rank, agent_base=rank * N, max_steps=260, steps_per_launch=65, device=None)
    recs = [ga.run_generation() for _ in range(2)]
    gathered = [None] * world
    dist.all_gather_object(gathered, recs)
    if rank == 0:
        json.dump(gathered, open(%(out)r, "w"))
    dist.barrier()
    dist.destroy_process_group()
''')


def test_two_rank_generation_loop_all_gathers_fitness(oracle, tmp_path):
    """The product's EvolutionaryRacer.run_generation (openkitchen_amd/evolution.py) on two gloo ranks, one island population
    per rank: island statistics equal a single-process replay of each island, and colony_best / colony_mean -- computed from
    the all-gathered fitness -- equal the statistics of the two islands' scores taken together, on both ranks."""
    import json
    out = str(tmp_path / "ga.json")
    script = tmp_path / "ga_worker.py"
    script.write_text(GA_WORKER % {"root": ROOT, "out": out})
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                    "--master-port", str(free_port()), str(script)], check=True, env=env, timeout=900, cwd=ROOT)
    ranks = json.load(open(out))
    assert len(ranks) == 2 and all(len(r) == 2 for r in ranks)
    # single-process replay of both islands (genetic_learner_sim.cpp:47-96 on the oracle)
    t = oracle.Track("Monza")
    N = 20
    start = (float(t.x[3]), float(t.y[3]), float(t.heading[0]))
    scores = {}
    for rank in range(2):
        env1 = oracle.OracleEnv(t.segments, N, 8, oracle.default_ray_fan(8), (t.x, t.y, t.heading))
        env1.set(oracle.F_MODE, np.ones(N, dtype=np.uint8))
        ga = oracle.OracleGA(env1, 30, 500 + rank, rank * N)
        for g in range(2):
            ga.reset_all(*start)
            env1.step(1)
            steps = 1
            while steps < 260:  # the reference's loop, one iteration at a time
                ga.rollout_policy(1)
                steps += 1
                if ga.alive_count() == 0:
                    break
            scores[(rank, g)] = ga.scores()
            rec = ranks[rank][g]
            assert rec["steps"] == steps
            assert rec["island_best"] == float(scores[(rank, g)].max())
            assert rec["island_mean"] == float(np.float32(torch_mean(scores[(rank, g)])))
            assert rec["parents"] == [int(v) for v in ga.select_mate(500 + rank, g, rank * N)]
    for g in range(2):
        both = np.stack([scores[(0, g)], scores[(1, g)]])
        for rank in range(2):
            assert ranks[rank][g]["colony_best"] == float(both.max())
            assert ranks[rank][g]["colony_mean"] == float(np.float32(torch_mean(both)))


def torch_mean(a):
    import torch
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).mean().item()
