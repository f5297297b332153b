"""One-GPU-per-shard data parallelism for agent populations.

Agents never interact inside Environment::step -- rays test track segments only -- so a population shards by agent
range with no data-path collective: rank g owns global agents [g*n, (g+1)*n), the track is replicated, and the Philox
streams are keyed by the GLOBAL agent id so a sharded run reproduces the unsharded one bit for bit.  The only exchange
is per generation: an all-gather of the fp32 fitness vector (EvolutionaryRacer; 32 KB per rank at 8192 agents,
latency-bound on the xGMI mesh), done with torch.distributed (backend "nccl" = RCCL on ROCm, "gloo" in CPU tests).
"""
import time

import torch
import torch.distributed as dist


_PINNED = {}  # (numel, dtype) -> pinned staging buffer of the gloo rehearsal path


def world():
    """(rank, world_size) of the default process group, (0, 1) when not initialised."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def shard_range(agents_per_rank, rank=None):
    """Global agent ids owned by `rank` under weak scaling (every rank owns `agents_per_rank` agents)."""
    if rank is None:
        rank, _ = world()
    return rank * agents_per_rank, (rank + 1) * agents_per_rank


def split_population(total_agents, world_size):
    """Strong-scaling split of a fixed population: contiguous, sizes differ by at most one."""
    base, extra = divmod(total_agents, world_size)
    sizes = [base + (1 if r < extra else 0) for r in range(world_size)]
    starts = [sum(sizes[:r]) for r in range(world_size)]
    return list(zip(starts, sizes))


def group_active():
    """True when a default process group exists -- at ANY world size: a one-rank RCCL communicator is a legal group and its
    collectives run (bench.py --force-dist), so that world size 1 exercises the same lines as world size 8."""
    return dist.is_available() and dist.is_initialized()


def barrier(device_ids=None):
    if group_active():
        if device_ids is not None and dist.get_backend() == "nccl":
            dist.barrier(device_ids=device_ids)
        else:
            dist.barrier()


def max_over_ranks(value, device="cpu"):
    """MAX all-reduce of a python float (the benchmark's elapsed time)."""
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    if group_active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def max_over_ranks_list(values, device="cpu"):
    """Element-wise MAX all-reduce of a list of python floats (the benchmark's per-repetition elapsed times)."""
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    if group_active():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.tolist()]


def all_gather_fitness(local_fitness):
    """All-gather of the per-rank fitness vector (1-D tensor, same length on every rank).  Returns [world, n]."""
    if not group_active():
        return local_fitness.unsqueeze(0).clone()
    w, n = dist.get_world_size(), local_fitness.numel()
    if local_fitness.is_cuda and dist.get_backend() != "nccl":
        # gloo rehearsal of the GPU path (no RCCL between two ranks on one card): the device tensor is what the product hands
        # over, only the collective itself goes through a pinned host copy and the gathered matrix returns to the device
        key = (n, local_fitness.dtype)
        host = _PINNED.get(key)
        if host is None:
            host = _PINNED[key] = torch.empty(n, dtype=local_fitness.dtype, pin_memory=True)
        host.copy_(local_fitness.contiguous().view(-1))
        flat = torch.empty(w * n, dtype=local_fitness.dtype)
        dist.all_gather_into_tensor(flat, host)
        return flat.to(local_fitness.device).view(w, n)
    # RCCL (any world size, 1 included) reads the device tensor; gloo gathers host tensors
    flat = torch.empty(w * n, dtype=local_fitness.dtype, device=local_fitness.device)
    dist.all_gather_into_tensor(flat, local_fitness.contiguous().view(-1))
    return flat.view(w, n)


class ShardedPopulation:
    """A population of `agents_per_rank * world` agents, this rank's shard living in `engine`.

    `engine_factory(num_agents, agent_base)` builds the shard's environment: openkitchen_amd.BatchedEnvironment on a
    GPU; the CPU tests inject an oracle-backed stand-in to exercise the sharding and the collective without a GPU.
    The engine must offer init_bench_state(agent_base, mode), rollout_random(n, seed, agent_base, step_base),
    nearest_track_idx() and sync().
    """

    def __init__(self, engine_factory, agents_per_rank, mode=0):
        self.rank, self.world = world()
        self.n = int(agents_per_rank)
        self.agent_base, _ = shard_range(self.n, self.rank)
        self.engine = engine_factory(self.n, self.agent_base)
        self.engine.init_bench_state(self.agent_base, mode)
        self.steps_done = 0

    def rollout(self, n_steps, seed, steps_per_launch=None):
        spl = n_steps if not steps_per_launch else steps_per_launch
        done = 0
        while done < n_steps:
            c = min(spl, n_steps - done)
            self.engine.rollout_random(c, seed, self.agent_base, self.steps_done + done)
            done += c
        self.steps_done += n_steps

    def fitness(self, device="cpu"):
        """EvolutionaryRacer's score: nearest centre-line index of every agent (MiscUtils.hpp:64-71), all-gathered."""
        idx = self.engine.nearest_track_idx()
        local = torch.as_tensor(idx, dtype=torch.float32, device=device)
        return all_gather_fitness(local)

    def timed(self, fn):
        """Runs fn() between barriers and returns the MAX elapsed time over ranks."""
        self.engine.sync()
        barrier()
        t0 = time.perf_counter()
        fn()
        self.engine.sync()
        elapsed = time.perf_counter() - t0
        barrier()
        return max_over_ranks(elapsed)


def share_q_knowledge(env):
    """shareCumulativeKnowledge (reference RLRacers/Q_Learning/q_racer_sim.cpp:24-75) across every rank's population: the
    per-entry sums and counts of the valid Q values are all-reduced (2 x 729 floats = 5.8 KB) and every agent on every
    rank receives the mean.  With one process this is okenv_q_share_knowledge."""
    if not group_active():
        env.q_share_knowledge()
        return
    import numpy as np

    sums, counts = env.q_table_sums()
    invalid = np.float32(np.finfo(np.float32).min)
    s = torch.as_tensor(np.where(counts > 0, sums, np.float32(0)), dtype=torch.float32)
    c = torch.as_tensor(counts, dtype=torch.float32)
    if dist.get_backend() == "nccl":
        s, c = s.cuda(), c.cuda()
    dist.all_reduce(s)
    dist.all_reduce(c)
    s, c = s.cpu().numpy(), c.cpu().numpy()
    env.q_assign_mean(np.where(c > 0, s, invalid).astype(np.float32), c.astype(np.float32))
