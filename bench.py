#!/usr/bin/env python3
"""bench.py -- headline benchmark of the batched Environment step on MI355X.

Metric (BASELINE.json): agent-steps/sec for config C2 = 4096 agents x 64 rays on Silverstone.csv with random
actions (SURVEY.md section 8d recipe: VELOCITY mode, throttle~U[0,100), steer~U[-5,5) from Philox4x32 keyed
(seed 1234, agent, step); crashed agents are re-placed on a Philox-chosen centre-line point at the start of the
next step).  A "step" is one Environment::step() over the whole population.  State is resident in HBM before
the timed region starts; nothing crosses PCIe inside it.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

Multi-GPU: one process per GPU; every rank owns its own shard of `--agents` agents (weak scaling: the global
population is N x 4096, agent ids are global so the Philox streams do not repeat across ranks).  Agents never
interact, so there is NO data-path collective; torch.distributed (RCCL) is used for the barriers around the
timed region and for the max-over-ranks of the elapsed time.

The single JSON line printed by rank 0 carries, besides the contract's keys:
  roofline     -- HBM roofline of the step kernel from ALGORITHMIC bytes (354 B per agent-step for C2,
                  SURVEY.md section 8d) divided by the kernel's average duration measured with HIP events on
                  the stream the kernel runs on.  The path is VALU/LDS-bound, not HBM-bound (BASELINE.md
                  section 5), so this fraction is small by construction; `traffic` is filled from the PMC
                  profile under profiles/ when available.
  cpu_baseline -- the CPU oracle (a port of the reference algorithm: brute-force sweep, `-O2
                  -ffp-contract=off`) timed on this host, single thread, on a bounded sample of the same
                  workload; `cpu_baseline_allcores` the same over all host cores; `cpu_c1` BASELINE config 1.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s


def algorithmic_bytes_per_agent_step(N, R, S):
    """read 44 B + write 36 B of agent state, 4*R B of observation (distances), and the
    shared segment array amortised over the population (16*S/N)."""
    return 44.0 + 36.0 + 4.0 * R + 16.0 * S / N


# everything the profiled kernel's behaviour depends on: the kernels, the walk, the grid builder, the launch geometry and defaults
# (cell edge, phase-1 range, block / grid sizes, tail limits: okenv_capi.hip) and the shared math (sincos, Philox).  (include/okenv.h is
# declarations and comments: what it declares is defined, and would change, in okenv_capi.hip)
KERNEL_SOURCES = ["openkitchen_amd/csrc/okenv_kernels.h", "openkitchen_amd/csrc/ok_raycast.h", "openkitchen_amd/csrc/ok_grid.h",
                  "openkitchen_amd/csrc/okenv_capi.hip", "include/okenv_math.h"]


def kernel_source_hash(root=ROOT):
    """sha256 over the step kernel's sources: profiles/hbm_traffic.json records the hash of the sources its counters were
    collected on, so that counters of an older kernel are never reported as the current kernel's."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(root, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def load_counter_profile(N, R, track_name, path=None, root=ROOT):
    """(profile dict or None, note or None): the committed PMC figures for this workload, only if they were collected on the
    kernel sources as they are now."""
    path = path or os.path.join(root, "profiles", "hbm_traffic.json")
    try:
        tj = json.load(open(path))
    except Exception as e:  # noqa: BLE001
        return None, "no counter profile (%s)" % type(e).__name__
    if (tj.get("agents"), tj.get("rays"), tj.get("track")) != (N, R, track_name):
        return None, "counter profile is for another workload"
    if tj.get("kernel_source_sha256") != kernel_source_hash(root):
        return None, "stale_profile: profiles/hbm_traffic.json was collected on other kernel sources (its kernel_source_sha256 differs); rerun tools/profile_run.sh on the GPU box and copy hbm_traffic.json"
    return tj, None


C3_BYTES_STATE = 44.0 + 36.0  # agent state read + written per agent-step


def bench_secondary_configs(args, ok, torch, local_rank, log):
    """BASELINE configs 3, 4 (one island) and 5 inside the default run (rank 0, N=1): one warm + two timed EvolutionaryRacer
    generations on Monza and on Spa (8192 x 32 rays, fused MLP policy) and one warm + two timed Q-learning episodes on Silverstone
    (16384 x 16 rays).
    `value` counts agents x steps of the reference's loop (every agent is in the loop until the last one has crashed);
    `live_value` counts only the agent-steps of agents that entered the step alive -- the ones that move, cast rays and read
    their policy's weights."""
    from openkitchen_amd.evolution import EvolutionaryRacer
    from openkitchen_amd.qlearning import QLearningRacers
    out = {}
    # ---- C3 (Monza) and one island of C4 (Spa: BASELINE's 8 x (8192 x 32) is eight of these plus a 32 KB fitness all-gather per generation) ----
    import torch.distributed as dist
    for key, track_name in (("c3", "Monza"), ("c4_island", "Spa")):
        N, R = 8192, 32
        track = ok.Track(track_name)
        # the C4 island gathers its fitness vector over a ONE-rank RCCL communicator (a legal group; the 8-GPU config has eight
        # ranks in it): the collective's lines run in the driver's default run too.  A failure to set it up must never cost the line.
        own_group, dist_note = False, None
        if key == "c4_island" and not dist.is_initialized():
            try:
                import socket
                s = socket.socket()
                s.bind(("127.0.0.1", 0))
                port = s.getsockname()[1]
                s.close()
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1,
                                        device_id=torch.device("cuda", local_rank))
                own_group = True
            except Exception as e:  # noqa: BLE001
                dist_note = "no one-rank RCCL group (%s: %s): the fitness all-gather was skipped" % (type(e).__name__, e)
        env = ok.BatchedEnvironment.from_track(track, N, R, device=local_rank)
        ga = EvolutionaryRacer(env, track, hidden=30, seed=args.seed, agent_base=0, max_steps=4000, steps_per_launch=args.steps_per_launch,
                               device=torch.device("cuda", local_rank))
        ga.run_generation()
        env.sync()
        env.set_timing(True)
        t0 = time.perf_counter()
        recs = [ga.run_generation() for _ in range(2)]
        env.sync()
        dt = time.perf_counter() - t0
        kernel_ms, launches = env.get_timing()
        env.set_timing(False)
        steps = sum(r["steps"] for r in recs)
        live = sum(r["live_agent_steps"] for r in recs)
        b_alg = C3_BYTES_STATE + 4.0 * R + 16.0 * track.S / N
        b_w = 4.0 * ((R + 2) * 30 + 30 * 6)  # the agent's 1200 policy weights, read once per live agent-step when not cached
        dist_rec = None
        if key == "c4_island":
            dist_rec = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                        "all_gather_us": [1e6 * r["all_gather_s"] for r in recs]} if dist.is_initialized() else {"note": dist_note}
        if own_group:
            dist.destroy_process_group()
        out[key] = {
            "dist": dist_rec,
            # value = agent-steps of agents that entered their step ALIVE (they move, cast rays, read their weights) per second;
            # nominal_value = agents x steps of the reference's loop, which keeps every agent in it until the last one has crashed
            "value": live / dt, "nominal_value": N * steps / dt, "live_value": live / dt, "unit": "agent-steps/s", "ms_per_step": dt / steps * 1e3,
            # who is still alive when a generation ends at the step cap, and where: agents outside the raycast grid's box have
            # tunnelled through both boundaries and can no longer crash
            "alive_at_end": [r["alive_at_end"] for r in recs], "off_grid_alive": [r["off_grid_alive"] for r in recs],
            "off_grid_agents": [r["off_grid_agents"] for r in recs],
            "generation_ms": [1e3 * (r["rollout_s"] + r["select_mate_s"]) for r in recs], "steps": [r["steps"] for r in recs],
            "live_fraction": live / float(N * steps), "kernel_ms_step_launches": kernel_ms, "launches": int(launches),
            "workload": "%s: EvolutionaryRacer, %d agents x %d rays, %s.csv, fused 34-30-6 MLP policy + Environment::step, rollout until all "
                        "crashed, score, select top-5, mate; 2 generations after 1 warm-up%s"
                        % ("C3" if key == "c3" else "C4, ONE island on one GPU", N, R, track_name,
                           "" if key == "c3" else " (the 8-GPU config adds the per-generation fitness all-gather: bench.py --config c4 --gpus 8)"),
            "roofline": {"bound": "hbm", "algorithmic_bytes_per_agent_step": b_alg + b_w, "of_which_policy_weights": b_w,
                         "achieved": (b_alg + b_w) * live / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (b_alg + b_w) * live / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else None,
                         "note": "bytes of LIVE agent-steps (crashed agents are not stepped) over the step launches' HIP-event time"}}
        env.close()
    # ---- C5 ----
    N, R = 16384, 16
    track = ok.Track("Silverstone")
    env = ok.BatchedEnvironment.from_track(track, N, R, device=local_rank)
    ql = QLearningRacers(env, track, seed=args.seed, agent_base=0, steps_per_launch=args.steps_per_launch)
    ql.run_episode()
    env.sync()
    env.set_timing(True)
    t0 = time.perf_counter()
    recs = [ql.run_episode() for _ in range(2)]
    env.sync()
    dt = time.perf_counter() - t0
    kernel_ms, launches = env.get_timing()
    env.set_timing(False)
    steps = sum(r["steps"] for r in recs)
    live = sum(r["live_agent_steps"] for r in recs)
    b_alg = C3_BYTES_STATE + 4.0 * R + 16.0 * track.S / N + 2 * 12.0 + 4.0  # + two table rows read, one entry written
    out["c5"] = {
        "value": live / dt, "nominal_value": N * steps / dt, "live_value": live / dt, "unit": "agent-steps/s", "ms_per_step": dt / steps * 1e3,
        "episode_ms": [1e3 * r["wall_s"] for r in recs], "steps": [r["steps"] for r in recs], "live_fraction": live / float(N * steps),
        "kernel_ms_step_launches": kernel_ms, "launches": int(launches),
        "workload": "C5: tabular Q-learning, %d agents x %d rays, Silverstone.csv, epsilon-greedy + reward + Q update fused into the step "
                    "kernel, 243x3 table per agent; 2 episodes after 1 warm-up" % (N, R),
        "roofline": {"bound": "hbm", "algorithmic_bytes_per_agent_step": b_alg,
                     "achieved": b_alg * live / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": b_alg * live / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else None,
                     "note": "bytes of LIVE agent-steps over the step launches' HIP-event time"}}
    env.close()
    log("secondary configs: c3 live %.3e (nominal %.3e) agent-steps/s, c5 live %.3e (nominal %.3e)" %
        (out["c3"]["value"], out["c3"]["nominal_value"], out["c5"]["value"], out["c5"]["nominal_value"]))
    return out


def cpu_baseline(track_name, R, seed, log):
    """Times the oracle on this host.  Only rank 0 at N=1 calls this; ~10-30 s of CPU work in total."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as O
    O.build_oracle(with_ref=False)
    tr = O.Track(track_name)
    fan = O.default_ray_fan(R)
    out = {}

    def run(N, steps, threads, track=tr, rays=fan):
        env = O.OracleEnv(track.segments, N, rays.size, rays, (track.x, track.y, track.heading))
        env.init_bench_state(0, 0)
        env.rollout_random(5, seed, 0, 0, threads=threads)  # touch memory, leave the start line
        t0 = time.perf_counter()
        env.rollout_random(steps, seed, 0, 5, threads=threads)
        dt = time.perf_counter() - t0
        return N * steps / dt, dt

    v, dt = run(128, 60, 1)
    log("cpu oracle, 1 thread, 128 agents x %d rays x 60 steps on %s: %.3e agent-steps/s (%.1f s)" % (R, track_name, v, dt))
    out["cpu_baseline"] = {"value": v, "unit": "agent-steps/s", "cores": 1, "kind": "port",
                           "sample": "128 agents x %d rays x 60 steps of the same recipe on %s, brute-force sweep over all %d segments"
                                     % (R, track_name, tr.S)}
    cores = os.cpu_count() or 1
    threads = min(cores, 64)
    v, dt = run(8 * threads, 60, threads)
    log("cpu oracle, %d threads: %.3e agent-steps/s (%.1f s)" % (threads, v, dt))
    out["cpu_baseline_allcores"] = {"value": v, "unit": "agent-steps/s", "cores": threads, "kind": "port",
                                    "sample": "%d agents x %d rays x 60 steps, agents partitioned over %d threads" % (8 * threads, R, threads)}
    tr1 = O.Track("Austin")
    fan1 = O.default_ray_fan(16)
    v, dt = run(64, 400, 1, tr1, fan1)
    log("cpu oracle, BASELINE config 1 (64 x 16, Austin, 1 thread): %.3e agent-steps/s (%.1f s)" % (v, dt))
    out["cpu_c1"] = {"value": v, "unit": "agent-steps/s", "cores": 1, "kind": "port",
                     "sample": "BASELINE config 1: 64 agents x 16 rays, Austin, 400 steps"}
    return out


def bench_callers(args, torch, local_rank, log):
    """Secondary figures for the callers of SURVEY.md section 8f on the tensor binding: one kernel launch per Environment
    step with a PyTorch policy in between (PPO-shaped loop) and one CMA-ES generation.  Rank 0, N=1 only."""
    from openkitchen_amd.cmaes import CmaEsRacers
    from openkitchen_amd.rollout import PPO_ACTIONS
    from openkitchen_amd.torch_env import VectorEnvironment
    out = {}
    N, steps = args.agents, 300
    fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32)
    venv = VectorEnvironment(args.track, N, ray_angles_deg=fan, device=local_rank, auto_reset=True, randomize_lane=True,
                             randomize_heading=True, seed=args.seed, reward="step")
    torch.manual_seed(args.seed)
    actor = torch.nn.Sequential(torch.nn.Linear(5, 128), torch.nn.ReLU(), torch.nn.Linear(128, 3), torch.nn.Softmax(dim=1)).cuda()
    table = torch.tensor(PPO_ACTIONS, dtype=torch.float32, device=venv.device)
    venv.reset()

    def loop(n):
        with torch.no_grad():
            for _ in range(n):
                probs = torch.clamp(actor(venv.observation()), 1e-8, 1.0 - 1e-8)
                venv.step(table[torch.multinomial(probs, 1).squeeze(1)])

    loop(20)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["vector_env_torch_policy"] = {
        "value": N * steps / dt, "unit": "agent-steps/s", "us_per_step": dt / steps * 1e6,
        "workload": "%d agents x 5 rays, %s, PPO actor 5-128-3 in PyTorch + multinomial between steps, device auto-reset "
                    "(lane + heading), +1 reward bookkeeping; one okenv_step launch per step" % (N, args.track),
        "episodes_finished": int((venv.episode_return > 0).sum())}
    # the same iteration captured once into a HIP graph and replayed (VectorEnvironment.capture)
    graph = venv.capture(lambda: loop(1))
    for _ in range(20):
        graph.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        graph.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out["vector_env_torch_policy_graph"] = {"value": N * steps / dt, "unit": "agent-steps/s", "us_per_step": dt / steps * 1e6,
                                            "workload": "same iteration, captured into one HIP graph and replayed"}
    venv.close()
    for key, fused, rollout in (("cmaes_generation", True, True), ("cmaes_generation_three_calls_graph", True, False),
                                ("cmaes_generation_torch_controller", False, False)):
        racers = CmaEsRacers(args.track, N, device=local_rank, seed=args.seed, max_steps=400, fused=fused, rollout=rollout)
        racers.run_generation()  # warm-up (eigh, allocator, graph capture)
        gen_ms = []
        for _ in range(7):  # the GPU box stalls a process for ~75 ms about ten times a second (its monitor): median and min
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            best, gsteps = racers.run_generation()
            torch.cuda.synchronize()
            gen_ms.append((time.perf_counter() - t0) * 1e3)
        dt = float(np.median(gen_ms)) * 1e-3
        # the loop alone on a freshly reset population, without sampling / eigh / tell (best of 4 x 100 iterations): the iteration
        # graph replayed, or -- the fused rollout -- one launch of 100 iterations
        racers.venv.reset(epoch=racers.generation)
        loop_us = []
        for _ in range(4):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            if rollout:
                racers.venv.env.rollout_controller(100, 100.0, 5.0)
            else:
                for _ in range(100):
                    racers._graph.replay()
            torch.cuda.synchronize()
            loop_us.append((time.perf_counter() - t1) / 100 * 1e6)
        loop_dt = min(loop_us) * 1e-6 * gsteps
        out[key] = {
            "value": N * gsteps / dt, "unit": "agent-steps/s", "generation_ms": dt * 1e3, "generation_ms_min": min(gen_ms),
            "generations_timed": len(gen_ms), "steps": gsteps, "best_fitness": best, "loop_us_per_step": loop_dt / gsteps * 1e6,
            "workload": "population %d (reference: 20), 250 parameters, controller 5-16-8-2 %s; generation_ms includes sampling and the host "
                        "eigendecomposition of the 250 x 250 covariance"
                        % (N, "fused with Environment::step and the index-progress fitness into the step kernel, the loop run as an episode "
                              "(okenv_rollout_controller)" if rollout else
                           ("as a libokenv kernel (okenv_controller_act), iteration (controller, Environment::step, index-progress fitness) "
                            "replayed as one HIP graph" if fused else "in PyTorch, iteration replayed as one HIP graph"))}
        racers.venv.close()
    # the C++ drop-in classes (include/Environment/): microseconds per Environment::step() at the population sizes the
    # reference's applications use (Template 1, PPO / REINFORCE 15, EvolutionaryRacer 50 agents); five-ray fan
    try:
        import subprocess
        exe = os.path.join(ROOT, "tools", "_build", "facade_bench")
        os.makedirs(os.path.dirname(exe), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), "-o", exe, os.path.join(ROOT, "tools", "facade_bench.cpp"),
                        "-L", os.path.join(ROOT, "openkitchen_amd"), "-lokenv", "-Wl,-rpath," + os.path.join(ROOT, "openkitchen_amd")], check=True)
        def run_facade(resident):
            env = dict(os.environ)
            if resident is not None:
                env["OKENV_RESIDENT"] = resident
            r = subprocess.run([exe, ok_track_path(args.track), "3000", "1", "15", "50"], check=True, capture_output=True, text=True, timeout=120,
                               env=env)
            return {"agents_%s" % line.split()[0]: float(line.split()[1]) for line in r.stdout.strip().splitlines()}
        out["facade_step_us"] = run_facade(None)  # as shipped: steps in quick succession are served by a resident kernel
        out["facade_step_us"]["launch_per_step"] = run_facade("0")  # what an application with long pauses between its steps sees
        out["facade_step_us"]["workload"] = ("Environment::step() of the C++ drop-in classes in a tight loop, five-ray agents on %s, resetAgent "
                                            "on crash; records handed over through mapped host memory (okenv_step_packed) to a step kernel "
                                            "that stays resident while steps keep coming within 100 us of each other; launch_per_step: "
                                            "OKENV_RESIDENT=0, one kernel launch per step" % args.track)
    except Exception as e:  # noqa: BLE001
        out["facade_step_us"] = {"error": "%s: %s" % (type(e).__name__, e)}
    log("callers: %s" % out)
    return out


def ok_track_path(name):
    import openkitchen_amd as ok
    return ok.track_path(name)


def bench_evolution(args, ok, torch, dist, rank, world, local_rank, log):
    """BASELINE configs 3 and 4: full EvolutionaryRacer generations on the device, one island population per GPU."""
    from openkitchen_amd import sharding
    from openkitchen_amd.evolution import EvolutionaryRacer

    track_name = "Monza" if args.config == "c3" else "Spa"
    N, R = 8192, 32
    track = ok.Track(track_name)
    env = ok.BatchedEnvironment.from_track(track, N, R, device=local_rank)
    ga = EvolutionaryRacer(env, track, hidden=30, seed=args.seed + rank, agent_base=rank * N, max_steps=4000,
                           steps_per_launch=args.steps_per_launch, device=torch.device("cuda", local_rank))  # scores stay on the device whatever the backend
    warm = ga.run_generation()  # warm-up generation (untimed)
    env.sync()
    torch.cuda.synchronize()
    sharding.barrier(device_ids=[local_rank])
    t0 = time.perf_counter()
    recs = [ga.run_generation() for _ in range(args.generations)]
    env.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sharding.barrier(device_ids=[local_rank])
    elapsed_max = sharding.max_over_ranks(elapsed, device=args.tensor_dev)
    steps = sum(r["steps"] for r in recs)
    steps_t = torch.tensor([float(steps), float(sum(r["live_agent_steps"] for r in recs))], dtype=torch.float64, device=args.tensor_dev)
    if args.dist_on:
        dist.all_reduce(steps_t)
    if rank == 0:
        total_agent_steps = N * float(steps_t[0].item())
        emit(args, {
            "metric": "agent-steps/sec", "value": total_agent_steps / elapsed_max, "unit": "agent-steps/s", "n_gpus": world,
            # agents x steps of the reference's loop (everybody is in it until the last agent has crashed); live_value counts the
            # agent-steps of agents that entered the step alive -- what is actually stepped
            "live_value": float(steps_t[1].item()) / elapsed_max, "live_fraction": float(steps_t[1].item()) / total_agent_steps,
            "steps": steps, "warmup": recs[0]["steps"], "ms_per_step": elapsed_max / max(steps, 1) * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%s: EvolutionaryRacer, %d agents x %d rays per GPU, %s.csv, fused 34-30-6 MLP policy + step, "
                                   "rollout until all crashed (<= 4000 steps), score, select top-5, mate%s"
                                   % (args.config.upper(), N, R, track_name, ", %s fitness all-gather per generation" % ("RCCL" if args.dist_backend == "nccl" else "gloo") if args.dist_on else ""),
                       "generations": args.generations, "parallelism": "dp%d island populations" % world},
            "generation_wall_s": elapsed_max / args.generations,
            "dist": dist_record(args, dist, world),
            "all_gather_us": [1e6 * r["all_gather_s"] for r in recs],
            # (for profiles taken over the whole process: rank 0's live agent-steps including the warm-up generation)
            "rank0_live_agent_steps_with_warmup": int(warm["live_agent_steps"] + sum(r["live_agent_steps"] for r in recs)),
            "rank0_steps_with_warmup": int(warm["steps"] + steps),
            "generations": [{k: r[k] for k in ("generation", "steps", "live_agent_steps", "rollout_s", "select_mate_s", "all_gather_s", "island_best", "island_mean", "colony_best", "colony_mean")} for r in recs],
        })
    env.close()
    if args.dist_on:
        sharding.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def bench_qlearning(args, ok, torch, dist, rank, world, local_rank, log):
    """BASELINE config 5: 16384 parallel Q-learning agents x 16 rays, discretised observation, device-side Q-table update."""
    from openkitchen_amd import sharding
    from openkitchen_amd.qlearning import QLearningRacers

    N, R = 16384, 16
    track = ok.Track(args.track)
    env = ok.BatchedEnvironment.from_track(track, N, R, device=local_rank)
    ql = QLearningRacers(env, track, seed=args.seed + rank, agent_base=rank * N, steps_per_launch=args.steps_per_launch)
    warm = ql.run_episode()  # warm-up episode
    env.sync()
    torch.cuda.synchronize()
    sharding.barrier(device_ids=[local_rank])
    t0 = time.perf_counter()
    steps, recs = 0, []
    while steps < args.steps:
        r = ql.run_episode()
        recs.append(r)
        steps += r["steps"]
    env.sync()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sharding.barrier(device_ids=[local_rank])
    elapsed_max = sharding.max_over_ranks(elapsed, device=args.tensor_dev)
    steps_t = torch.tensor([float(steps), float(sum(r["live_agent_steps"] for r in recs))], dtype=torch.float64, device=args.tensor_dev)
    if args.dist_on:
        dist.all_reduce(steps_t)
    if rank == 0:
        table = env.q_table()
        emit(args, {
            "metric": "agent-steps/sec", "value": N * float(steps_t[0].item()) / elapsed_max, "unit": "agent-steps/s", "n_gpus": world,
            "live_value": float(steps_t[1].item()) / elapsed_max, "live_fraction": float(steps_t[1].item()) / (N * float(steps_t[0].item())),
            "steps": steps, "warmup": 0, "ms_per_step": elapsed_max / max(steps, 1) * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C5: tabular Q-learning, %d agents x %d rays per GPU, %s.csv, epsilon-greedy + reward + Q update fused into "
                                   "the step kernel, 243x3 table per agent (%.1f MB)" % (N, R, args.track, N * 243 * 3 * 4 / 1e6),
                       "episodes": len(recs), "parallelism": "dp%d" % world},
            "dist": dist_record(args, dist, world),
            "episodes": recs, "learned_entries_fraction": float((table > -1e30).mean()),
            "rank0_live_agent_steps_with_warmup": int(warm["live_agent_steps"] + sum(r["live_agent_steps"] for r in recs)),
            "rank0_steps_with_warmup": int(warm["steps"] + steps),
        })
    env.close()
    if args.dist_on:
        sharding.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


def emit(args, result):
    """The run's one line, on the descriptor that was stdout when the script started."""
    os.write(args.json_fd, (json.dumps(result) + "\n").encode())


def dist_record(args, dist, world):
    """What the line says about the process group its collectives ran on (None: no group, every collective was skipped)."""
    if not (args.dist_on and dist.is_initialized()):
        return None
    return {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "forced_at_world_1": bool(args.force_dist and world == 1),
            "collectives": "barrier(device_ids), all_reduce(MAX) of the region times, all_reduce(SUM) of the step counts"
                           + (", all_gather_into_tensor of the fitness vector per generation" if args.config == "c4" or args.config == "c3" else "")}


def launch_ranks(n, argv):
    """Starts `n` ranks of this script (one per GPU) through torch.distributed.run and returns their exit code.  Runs in a
    process that has not imported torch, let alone touched the GPU; the ranks are fresh children, nothing is exec'ed.  Rank 0's
    stdout (the ONE JSON line) is relayed as it comes, anything else found there goes to stderr; stderr is shared."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % n, "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    print("[bench] --gpus %d without WORLD_SIZE: starting the ranks: %s" % (n, " ".join(cmd)), file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, cwd=ROOT)
    try:
        for line in child.stdout:
            # stdout carries the ONE JSON line; whatever else the ranks' libraries print there (gloo announces its connections
            # on stdout) goes to stderr
            out = sys.stdout if line.lstrip().startswith("{") else sys.stderr
            out.write(line)
            out.flush()
        return child.wait()
    except BaseException:
        child.terminate()  # the exact process we started; torchrun forwards the signal to its ranks
        child.wait()
        raise


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--agents", type=int, default=4096, help="agents per GPU (C2: 4096)")
    ap.add_argument("--rays", type=int, default=64)
    ap.add_argument("--track", default="Silverstone")
    ap.add_argument("--seed", type=int, default=1234)
    ap.add_argument("--steps-per-launch", type=int, default=100,
                    help="Environment steps advanced by one kernel launch (the action source is on the device)")
    ap.add_argument("--grid-cell", type=float, default=0.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", default="c2", choices=["c2", "c3", "c4", "c5"],
                    help="c2: headline (random actions); c3/c4: EvolutionaryRacer generations (population 8192 x 32 rays per GPU, "
                         "Monza / Spa, fused MLP policy, score, select, mate; c4 adds the per-generation RCCL fitness all-gather)")
    ap.add_argument("--repeats", type=int, default=30,
                    help="the K-step timed region is run this many times back to back (state continues); `value` is from the median")
    ap.add_argument("--generations", type=int, default=5)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl (= RCCL, default).  gloo + --single-device rehearses the multi-process path on a one-GPU box")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses GPU 0")
    ap.add_argument("--force-dist", action="store_true",
                    help="create the process group at world size 1 too (RANK 0 of 1, 127.0.0.1): the barriers, the max-over-ranks "
                         "all-reduce and C4's fitness all-gather then run on a real one-rank RCCL communicator instead of being skipped")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary one-launch-per-step and host-boundary loops (used under rocprofv3 so that the "
                         "kernel trace holds only the timed region's launches)")
    args = ap.parse_args()

    # ---- --gpus N is a promise: the line printed at the end says n_gpus = N or the run fails ----
    # Under torchrun (the driver's way of starting N > 1) WORLD_SIZE is set and must equal --gpus.  Started plainly with
    # --gpus N > 1, this process becomes a launcher: it starts the N ranks itself, BEFORE anything here has touched the GPU
    # (torch is not even imported yet), relays rank 0's JSON line and exits with the ranks' exit code.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print("[bench] rank %d: --gpus %d but WORLD_SIZE=%d: refusing to run (the result would carry the wrong n_gpus); start it as "
              "`python bench.py --gpus %d ...` or under torchrun with --nproc-per-node %d"
              % (rank, args.gpus, world, args.gpus, args.gpus), file=sys.stderr, flush=True)
        raise SystemExit(2)

    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on stdout when a communicator is created,
    # gloo announces its connections there.  The descriptor is put aside for the line and everything else that writes to fd 1
    # from here on (Python or native) lands on stderr.
    sys.stdout.flush()
    args.json_fd = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault("RCCL_LOG_LEVEL", "0")

    import torch
    import torch.distributed as dist

    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:  # --force-dist started plainly: this process is rank 0 of 1, no launcher, nothing re-exec'ed
            import socket
            s = socket.socket()
            s.bind(("127.0.0.1", 0))
            os.environ.setdefault("MASTER_PORT", str(s.getsockname()[1]))
            s.close()
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            os.environ.setdefault("LOCAL_RANK", "0")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the Environment step has no CPU fallback")
    if args.single_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    tensor_dev = "cuda" if args.dist_backend == "nccl" else "cpu"  # where the tiny timing tensors live
    if dist_on:  # the first thing that touches the GPU after set_device: the RCCL communicator
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend="gloo")
    args.tensor_dev = tensor_dev
    args.dist_on = dist_on

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    import openkitchen_amd as ok

    ok.build()
    if args.config in ("c3", "c4"):
        return bench_evolution(args, ok, torch, dist, rank, world, local_rank, log)
    if args.config == "c5":
        return bench_qlearning(args, ok, torch, dist, rank, world, local_rank, log)
    track = ok.Track(args.track)
    N, R = args.agents, args.rays
    env = ok.BatchedEnvironment.from_track(track, N, R, device=local_rank, grid_cell=args.grid_cell)
    info = env.info()
    log("env: %s" % info)
    agent_base = rank * N
    env.init_bench_state(agent_base, ok.capi.MODE_VELOCITY)
    spl = max(1, min(args.steps_per_launch, args.steps))

    def run_steps(n, step0):
        done = 0
        while done < n:
            c = min(spl, n - done)
            env.rollout_random(c, args.seed, agent_base, step0 + done)
            done += c

    from openkitchen_amd import sharding

    def fence():
        env.sync()
        torch.cuda.synchronize()
        sharding.barrier(device_ids=[local_rank])

    # ---- warm-up (untimed) ----
    run_steps(args.warmup, 0)
    fence()
    # ---- timed region: exactly K steps between two fences, repeated `repeats` times back to back (the state simply
    # continues).  One region at the driver's --steps 20 is a single 0.3 ms launch, so a single sample says little: the
    # headline is the MEDIAN region, min / max / spread are reported beside it.  Every region's time is the max over ranks.
    repeats = max(1, args.repeats)
    env.set_timing(True)
    region_s = []
    for rep in range(repeats):
        fence()
        t0 = time.perf_counter()
        run_steps(args.steps, args.warmup + rep * args.steps)
        env.sync()
        torch.cuda.synchronize()
        region_s.append(time.perf_counter() - t0)
    sharding.barrier(device_ids=[local_rank])
    kernel_ms, launches = env.get_timing()
    env.set_timing(False)
    region_s = sharding.max_over_ranks_list(region_s, device=args.tensor_dev)
    elapsed_max = float(np.median(region_s))
    steps_done = args.warmup + repeats * args.steps

    # ---- secondary figure: one launch per step (what a host-side policy between steps would see) ----
    one_steps = 0 if args.headline_only else min(args.steps, 500)
    env.sync()
    t1 = time.perf_counter()
    for s in range(one_steps):
        env.rollout_random(1, args.seed, agent_base, steps_done + s)
    env.sync()
    one_elapsed = time.perf_counter() - t1

    # ---- tertiary figure: the host boundary (actions uploaded, distances downloaded every step: PCIe inclusive) ----
    pcie_steps = 0 if args.headline_only else min(args.steps, 200)
    thr_h = np.full(N, 30.0, dtype=np.float32)
    steer_h = np.zeros(N, dtype=np.float32)
    dist_h = np.zeros((N, R), dtype=np.float32)
    env.sync()
    t2 = time.perf_counter()
    for s in range(pcie_steps):
        env.set_actions(thr_h, steer_h)
        env.step(1)
        env.get(ok.capi.F_DIST, dist_h)
    pcie_elapsed = time.perf_counter() - t2

    state = env.snapshot()
    crashed_frac = float(state["crashed"].mean())
    work = None
    try:  # what the broad phase leaves of the reference's R x S tests per agent-step, at the population's poses right now
        split = info.get("front_back_bytes", 0) > 0
        ws = env.work_stats_split() if split else env.work_stats()
        if ws["rays"] > 0:
            work = {k: ws[k] / float(ws["rays"]) for k in ("tests", "cells", "points")}
            if split:
                work.update({k: ws[k] / float(ws["rays"]) for k in ("certified", "ambiguous", "back_walked")})
                wc = env.work_stats()
                work["combined_image"] = {k: wc[k] / float(wc["rays"]) for k in ("tests", "cells", "points")}
    except Exception as e:  # noqa: BLE001
        log("work stats unavailable: %s" % e)
    callers = None
    if rank == 0 and world == 1 and not args.headline_only:
        try:  # secondary figures must never cost the headline line
            callers = bench_callers(args, torch, local_rank, log)
        except Exception as e:  # noqa: BLE001
            callers = {"error": "%s: %s" % (type(e).__name__, e)}
            log("callers benchmark failed: %s" % callers["error"])

    if rank == 0:
        total_agents = N * world
        value = total_agents * args.steps / elapsed_max
        b_alg = algorithmic_bytes_per_agent_step(N, R, track.S)
        avg_launch_s = (kernel_ms * 1e-3) / max(launches, 1)
        steps_per_launch_avg = repeats * args.steps / max(launches, 1)
        bytes_per_launch = b_alg * N * steps_per_launch_avg
        achieved_gbs = bytes_per_launch / avg_launch_s / 1e9 if avg_launch_s > 0 else 0.0
        kernel_us_per_step = kernel_ms * 1e3 / (repeats * args.steps)
        traffic, valu = None, None
        # measured HBM bytes and issue statistics from the committed PMC profile of this kernel, kept PER AGENT-STEP so that they
        # apply to any --steps / --steps-per-launch; the workload has to match and so do the kernel's sources (a profile of an
        # older kernel is reported as stale, never as current)
        tj, profile_note = load_counter_profile(N, R, args.track)
        try:
            if tj is not None:
                traffic = tj["hbm_bytes_per_agent_step"] * N * steps_per_launch_avg
                w = tj["per_wave_step"]
                simds = 4 * info["compute_units"]  # four SIMDs per CU; the CU count is what HIP reports for this device
                waves_per_simd = N * info["lanes_per_agent"] / 64.0 / simds
                cyc_per_step = kernel_us_per_step * 1e-6 * tj["clock_ghz"] * 1e9
                valu = {"bound": "the waves' own instruction streams: VALU issue share x lane utilisation is what roofline.frac reports; the rest "
                                 "of a wave's time is waiting on LDS round trips and issue stalls of its dependent chain",
                        "valu_insts_per_wave_step": w["valu"], "salu_insts_per_wave_step": w["salu"], "lds_insts_per_wave_step": w["lds"],
                        "cycles_per_valu_inst": 2, "simds": simds, "clock_ghz": tj["clock_ghz"],
                        # share of a SIMD's issue cycles its waves' VALU instructions take at 2 cycles each
                        # (/opt/skills/guides/MI355X_MICROARCH.md: v_fma_f32 wave64 = 2 cycles on the SIMD-32)
                        "simd_valu_issue_frac": w["valu"] * waves_per_simd * 2.0 / cyc_per_step if cyc_per_step > 0 else None,
                        "wave_time_active_frac": w["active_frac"], "wave_time_wait_frac": w["wait_frac"],
                        "wave_time_issue_stall_frac": w["issue_stall_frac"], "valu_lane_utilisation": w.get("lane_util"),
                        "source": tj["source"]}
        except Exception as e:  # noqa: BLE001
            traffic, valu, profile_note = None, None, "counter profile unreadable (%s)" % type(e).__name__
        secondary = None
        if world == 1 and not args.headline_only:
            try:  # secondary figures must never cost the headline line
                secondary = bench_secondary_configs(args, ok, torch, local_rank, log)
            except Exception as e:  # noqa: BLE001
                secondary = {"error": "%s: %s" % (type(e).__name__, e)}
                log("secondary configs failed: %s" % secondary["error"])
        hbm = {"achieved": achieved_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved_gbs / HBM_PEAK_GBS}
        roofline = {
            # what binds: BASELINE.json words the target as a share of the HBM roofline, and that share is reported (hbm_*), but the
            # path is not HBM-bound and cannot be (354 algorithmic bytes against ~2000 VALU instructions per agent-step: SURVEY section
            # 8d) -- it is bound by the VALU work of the grid walk: `frac` is the share of the machine's VALU lane-cycles that did
            # anything = VALU issue share of the SIMDs (2 cycles per wave64 instruction) x lane utilisation, from the committed
            # counter profile of THIS kernel (null -> bound falls back to "hbm" when that profile is stale)
            "bound": "valu" if valu and valu.get("simd_valu_issue_frac") and valu.get("valu_lane_utilisation") else "hbm",
            "hbm_achieved": hbm["achieved"], "hbm_peak": hbm["peak"], "hbm_unit": "GB/s", "hbm_frac": hbm["frac"],
            "traffic": traffic,
            "traffic_unit": "HBM bytes per launch = measured bytes per agent-step (PMC FETCH_SIZE x2 + WRITE_SIZE, "
                            "profiles/hbm_traffic.json) x agents x steps per launch",
            "algorithmic_bytes_per_launch": bytes_per_launch,
            "kernel": "okStepCoopKernel" if info["grid_in_lds"] and R <= 64 else "okStepKernel",
            "algorithmic_bytes_per_agent_step": b_alg,
            "avg_launch_ms": avg_launch_s * 1e3, "launches": int(launches),
            "kernel_only_agent_steps_per_sec": N * repeats * args.steps / (kernel_ms * 1e-3) if kernel_ms > 0 else None,
            "kernel_us_per_step": kernel_us_per_step,
            "note": "algorithmic HBM traffic is ~0.35 KB per agent-step (BASELINE.md section 5); achieved / peak / unit / frac describe "
                    "the binding resource named by `bound`, the HBM roofline BASELINE.json asks for is hbm_achieved / hbm_peak / hbm_frac"}
        if roofline["bound"] == "valu":
            peak_tl = valu["simds"] * 32.0 * valu["clock_ghz"] * 1e9 / 1e12  # wave64 VALU instruction = 2 cycles of a SIMD: 32 lanes per cycle
            fr = valu["simd_valu_issue_frac"] * valu["valu_lane_utilisation"]
            roofline.update({"achieved": fr * peak_tl, "peak": peak_tl, "unit": "Tlane-inst/s", "frac": fr,
                             "valu_issue_frac": valu["simd_valu_issue_frac"], "valu_lane_utilisation": valu["valu_lane_utilisation"]})
        else:
            roofline.update({"achieved": hbm["achieved"], "peak": hbm["peak"], "unit": hbm["unit"], "frac": hbm["frac"]})
        result = {
            "metric": "agent-steps/sec",
            "value": value,
            "unit": "agent-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed_max / args.steps * 1e3,
            "repeats": repeats,
            "ms_per_step_min": min(region_s) / args.steps * 1e3,
            "ms_per_step_max": max(region_s) / args.steps * 1e3,
            "spread": (max(region_s) - min(region_s)) / elapsed_max if elapsed_max > 0 else None,
            "spread_p10_p90": float(np.percentile(region_s, 90) - np.percentile(region_s, 10)) / elapsed_max if elapsed_max > 0 else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: %d agents x %d rays per GPU, %s.csv (S=%d segments), Philox random actions + auto reset, "
                                   "HIP fused kinematics+raycast+collision" % (N, R, args.track, track.S),
                       "agents_per_gpu": N, "rays": R, "track": args.track, "segments": track.S,
                       "global_agents": total_agents, "agent_base_per_rank": [r * N for r in range(world)],
                       "steps_per_launch": spl, "parallelism": "dp%d (agent shards, no collective)" % world,
                       "grid_cell": info["grid_cell"], "lds_bytes": info["front_back_bytes"] or info["lds_bytes"]},
            "rays_per_sec": value * R,
            # what the reference's sweep would have to evaluate for the same result: every ray against all S segments
            "brute_force_equivalent_ray_segment_tests_per_sec": value * R * track.S,
            "valu_roofline": valu,
            "counter_profile_note": profile_note,
            # S_tested = exact ray-segment tests per ray (one whole-ray walk per ray at the final poses; the
            # cooperative kernel's speculative intervals add to it), valu_fraction = agent-steps/s x R x S_tested x 15 FLOP over
            # the 157.3 TFLOP/s fp32 vector peak -- the share of the VALU peak the REFERENCE's arithmetic per surviving test takes
            "broad_phase": None if work is None else {
                "s_tested_per_ray": work["tests"], "cells_per_ray": work["cells"], "points_per_ray": work["points"],
                "segments": track.S, "tested_fraction_of_sweep": work["tests"] / track.S,
                # front / back split of the segment set: share of the rays whose origin is certified to lie between
                # the inner boundaries, whose front walk was ambiguous, that walked the back image too; and the same poses' figures on
                # the combined image (what every ray met before the split)
                "front_back": None if "certified" not in work else {
                    "rays_of_certified_origins": work["certified"], "ambiguous_front_walks": work["ambiguous"],
                    "rays_walking_the_back_image": work["back_walked"], "lds_bytes_front_and_back": info["front_back_bytes"],
                    "back_segments": info["back_segments"], "combined_image_per_ray": work["combined_image"]},
                "valu_fraction": value / world * R * work["tests"] * 15.0 / 157.3e12},
            "configs": secondary,
            "value_one_launch_per_step": (N * one_steps / one_elapsed) if one_steps else None,
            "value_host_boundary_pcie_inclusive": (N * pcie_steps / pcie_elapsed) if pcie_steps else None,
            "crashed_fraction_at_end": crashed_frac,
            "dist": dist_record(args, dist, world),
            "callers": callers,
            "roofline": roofline,
        }
        result["cpu_baseline"] = None
        if world == 1 and not args.no_cpu_baseline:
            try:
                result.update(cpu_baseline(args.track, R, args.seed, log))
            except Exception as e:  # noqa: BLE001  (e.g. no C compiler for the oracle on this host)
                log("cpu baseline failed: %s: %s" % (type(e).__name__, e))
        emit(args, result)
    env.close()
    if args.dist_on:
        sharding.barrier(device_ids=[local_rank])
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
