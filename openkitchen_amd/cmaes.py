"""CMA-ES population racing on the device environment.

Mirror of the reference's CovarianceMatrixAdaptationEvolution app for N candidates at once:
  * `CmaEsSolver`      -- CmaEsSolverEigen.cpp:26-132 (host, float64; the rank-mu/rank-one update of arXiv:1604.00772)
  * `BatchedController`-- Controller.cpp:3-23, one parameter vector per candidate, evaluated for all candidates at once
  * `CmaEsRacers`      -- the generation loop of main_eigen.cpp:113-171: sample, resetAgent, one observation step, then
                          {updateAction; env.step(); fitness += |index progress|} until every candidate has crashed.
The environment step, the candidates' controllers and the fitness bookkeeping run on the device.  By default the whole inner loop
is ONE kernel per launch (okenv_rollout_controller: controller, Environment::step and bookkeeping fused into the step kernel, as an
episode -- later launches cover the candidates that can still change, the loop's own length comes back); `rollout=False` makes the
three calls per iteration (okenv_controller_act, okenv_step, okenv_tracker_update), replayed as one HIP graph; `fused=False`
evaluates the controllers in plain PyTorch instead (`BatchedController`, a dozen small kernels per iteration).  All three give the
same fitness bit for bit except the PyTorch controllers (their tanh and summation order are torch's).  The reference seeds std::mt19937 from std::random_device
(CmaEsSolverEigen.h:49), so its sample streams are not reproducible; here a seeded numpy Generator draws z.
"""
import contextlib

import numpy as np
import torch

from . import _capi as capi
from .torch_env import VectorEnvironment

try:  # the solver's 250 x 250 linear algebra gains nothing from dozens of BLAS threads, and their spin-waiting after each call
    # can use up a container's CPU quota: the whole process is then paused until the next scheduler period (measured on the GPU
    # box: a 21 ms generation took 100 ms)
    from threadpoolctl import threadpool_limits as _threadpool_limits
except ImportError:  # pragma: no cover
    _threadpool_limits = None


def _few_blas_threads(n=4):
    return _threadpool_limits(limits=n) if _threadpool_limits is not None else contextlib.nullcontext()


class CmaEsSolver:
    def __init__(self, num_params, population_size, seed=0, sigma=0.5, device=None):
        """device: where the candidates are drawn and transformed (a torch device; None = numpy on the host).  The
        250 x 250 state (mean, C, evolution paths) always lives on the host in float64."""
        n, lam = int(num_params), int(population_size)
        self.num_params, self.population_size, self.num_parents = n, lam, lam // 2
        self.sigma = float(sigma)
        self.mean = np.zeros(n)
        self.C = np.eye(n)
        self.p_sigma = np.zeros(n)
        self.p_c = np.zeros(n)
        mu = self.num_parents
        w = np.log(mu + 0.5) - np.log(np.arange(1, mu + 1))
        self.weights = w / w.sum()
        self.mu_eff = 1.0 / np.sum(self.weights ** 2)
        self.c_sigma = (self.mu_eff + 2.0) / (n + self.mu_eff + 5.0)
        self.d_sigma = 1.0 + 2.0 * max(0.0, np.sqrt((self.mu_eff - 1.0) / (n + 1.0)) - 1.0) + self.c_sigma
        self.c_c = (4.0 + self.mu_eff / n) / (n + 4.0 + 2.0 * self.mu_eff / n)
        self.c_1 = 2.0 / ((n + 1.3) ** 2 + self.mu_eff)
        self.c_mu = min(1.0 - self.c_1, 2.0 * (self.mu_eff - 2.0 + 1.0 / self.mu_eff) / ((n + 2.0) ** 2 + self.mu_eff))
        self.chi_n = np.sqrt(n) * (1.0 - 1.0 / (4.0 * n) + 1.0 / (21.0 * n * n))
        self.B = np.eye(n)
        self.D = np.ones(n)
        self.rng = np.random.default_rng(seed)
        self.device = device
        if device is not None:
            self.generator = torch.Generator(device=device).manual_seed(int(seed))

    def sample(self):
        """Candidates x_i = mean + sigma * B (D * z_i), z_i ~ N(0, I); returns float32 [population, num_params]."""
        with _few_blas_threads(1):  # (250 x 250: 2.8 ms on one thread, 3.6 ms on four or more -- measured on the GPU box's host)
            evals, self.B = np.linalg.eigh(self.C)
        self.D = np.sqrt(evals)
        if self.device is None:
            z = self.rng.standard_normal((self.population_size, self.num_params))
            y = (z * self.D) @ self.B.T
            return (self.mean + self.sigma * y).astype(np.float32)
        # population x n normals and the n x n rotation on the device (float64): with thousands of candidates this is
        # the bulk of the solver's arithmetic
        f64 = dict(dtype=torch.float64, device=self.device)
        z = torch.randn(self.population_size, self.num_params, generator=self.generator, **f64)
        y = (z * torch.as_tensor(self.D, **f64)) @ torch.as_tensor(self.B, **f64).T
        return (torch.as_tensor(self.mean, **f64) + self.sigma * y).to(torch.float32)

    def tell(self, solutions, fitness):
        """Higher fitness is better (the reference sorts descending, CmaEsSolverEigen.cpp:86-90)."""
        order = np.argsort(-np.asarray(fitness, dtype=np.float64), kind="stable")[: self.num_parents]
        if torch.is_tensor(solutions):
            # the two products over the parents (the weighted mean and the rank-mu matrix: with thousands of candidates the bulk of
            # the update's arithmetic) are formed where the candidates live, in float64; the mean and an n x n matrix travel to the host
            f64 = dict(dtype=torch.float64, device=solutions.device)
            parents = solutions[torch.as_tensor(order, device=solutions.device)].to(torch.float64)
            w = torch.as_tensor(self.weights, **f64)
            mean = w @ parents
            y = (parents - torch.as_tensor(self.mean, **f64)) / self.sigma
            rank_mu = (y * w[:, None]).T @ y
            mean, rank_mu = mean.cpu().numpy(), rank_mu.cpu().numpy()
        else:
            parents = np.asarray(solutions, dtype=np.float64)[order]
            with _few_blas_threads():
                mean = self.weights @ parents
                y = (parents - self.mean) / self.sigma
                rank_mu = (y * self.weights[:, None]).T @ y
        with _few_blas_threads():
            self._update(mean, rank_mu)

    def _update(self, mean, rank_mu):
        old_mean = self.mean
        self.mean = mean
        y_w = (self.mean - old_mean) / self.sigma
        # C^(-1/2) y_w = B diag(1 / D) B^T y_w as three matrix-vector products (the n x n matrix itself is never needed)
        inv_sqrt_c_y = self.B @ ((self.B.T @ y_w) / self.D)
        self.p_sigma = (1.0 - self.c_sigma) * self.p_sigma + np.sqrt(self.c_sigma * (2.0 - self.c_sigma) * self.mu_eff) * inv_sqrt_c_y
        self.p_c = (1.0 - self.c_c) * self.p_c + np.sqrt(self.c_c * (2.0 - self.c_c) * self.mu_eff) * y_w
        self.C = (1.0 - self.c_1 - self.c_mu) * self.C + self.c_1 * np.outer(self.p_c, self.p_c) + self.c_mu * rank_mu
        self.sigma *= np.exp((self.c_sigma / self.d_sigma) * (np.linalg.norm(self.p_sigma) / self.chi_n - 1.0))

    def get_best_solution(self):
        return self.mean.astype(np.float32)


class BatchedController:
    """tanh(fc3(tanh(fc2(tanh(fc1(x)))))) with fc1: in->h, fc2: h->h/2, fc3: h/2->out, one weight set per candidate.
    The flat parameter order is torch's `parameters()` order of the reference module: fc1.weight [h, in] row-major,
    fc1.bias, fc2.weight, fc2.bias, fc3.weight, fc3.bias (Controller.cpp:36-53).  The parameters live in one
    preallocated [population, num_params] tensor (`set_params` copies into it), so a captured graph of `forward`
    stays valid across generations.  Layers are evaluated as broadcast multiply + sum: for 16x5 matrices that is
    one small elementwise kernel, six times cheaper than a batched GEMM call."""

    def __init__(self, input_size, hidden_size, output_size, device, population=None):
        self.sizes = [(hidden_size, input_size), (hidden_size // 2, hidden_size), (output_size, hidden_size // 2)]
        self.device = device
        self.flat = None if population is None else torch.zeros(population, self.count_params(), device=device)
        self.layers = None if population is None else self._views()

    def count_params(self):
        return sum(o * i + o for o, i in self.sizes)

    def _views(self):
        layers, off = [], 0
        for o, i in self.sizes:
            w = self.flat[:, off:off + o * i].view(-1, o, i)
            off += o * i
            b = self.flat[:, off:off + o]
            off += o
            layers.append((w, b))
        return layers

    def set_params(self, flat):
        if not torch.is_tensor(flat):
            flat = torch.from_numpy(np.ascontiguousarray(flat, dtype=np.float32))
        assert flat.dim() == 2 and flat.shape[1] == self.count_params()
        if self.flat is None or self.flat.shape != flat.shape:
            self.flat = torch.empty(flat.shape, dtype=torch.float32, device=self.device)
            self.layers = self._views()
        self.flat.copy_(flat)  # host array or device tensor, straight into the preallocated buffer

    def forward(self, x):
        for w, b in self.layers:
            x = torch.tanh((w * x.unsqueeze(1)).sum(dim=2) + b)
        return x


class CmaEsRacers:
    RAYS = (-70.0, -30.0, 0.0, 30.0, 70.0)  # CmaEsAgent's fan (main_eigen.cpp:26-31)
    HIDDEN, OUTPUTS = 16, 2                  # main_eigen.cpp:18-19

    def __init__(self, track, population_size=20, device=0, seed=0, reset_randomly=False, max_steps=None, fused=True, rollout=True,
                 steps_per_launch=400):
        self.venv = VectorEnvironment(track, population_size, ray_angles_deg=np.array(self.RAYS, dtype=np.float32),
                                      device=device, movement_mode=capi.MODE_VELOCITY, auto_reset=False,
                                      pick_random_point=reset_randomly, seed=seed, reward="progress")
        self.fused = bool(fused)
        self.rollout = bool(rollout) and self.fused
        # (long launches: a wave whose candidates are all done leaves by itself, and unlike the MLP / Q-learning rollouts there is no
        # one-agent-per-workgroup kernel to hand short lists to -- 400 steps per launch: 7.9 ms per generation, 100: 8.8 ms)
        self.steps_per_launch = int(steps_per_launch)
        self.controller = BatchedController(len(self.RAYS), self.HIDDEN, self.OUTPUTS, self.venv.device, population_size)
        if self.fused:
            assert self.venv.env.controller_create(self.HIDDEN) == self.controller.count_params()
        self.solver = CmaEsSolver(self.controller.count_params(), population_size, seed=seed, device=self.venv.device)
        self.max_steps = max_steps
        self.generation = 0
        self._graph = None

    def set_params(self, population):
        """Controller::set_params for every candidate (main_eigen.cpp:120-125)."""
        if self.fused:
            if torch.is_tensor(population):
                population = population.to(device=self.venv.device, dtype=torch.float32).contiguous()
            self.venv.env.controller_set_params(population)
        else:
            self.controller.set_params(population)

    def update_action(self):
        """CmaEsAgent::updateAction (main_eigen.cpp:58-68): full throttle, steering = 5 * first output."""
        if self.fused:
            self.venv.env.controller_act(100.0, 5.0)
            return
        out = self.controller.forward(self.venv.observation())
        self.venv.set_action(100.0, out[:, 0] * 5.0)

    def _iteration(self):
        self.update_action()
        self.venv.step()

    def _rollout_episode(self):
        """main_eigen.cpp:135-160 until every candidate has crashed (or max_steps): returns the loop's own length."""
        env = self.venv.env
        env.episode_begin()
        budget = self.max_steps if self.max_steps is not None else 1 << 30
        taken = 0
        while taken < budget:
            n = min(self.steps_per_launch, budget - taken)
            env.rollout_controller(n, 100.0, 5.0)
            taken += n
            alive, _ = env.episode_compact()
            if alive == 0:
                break
        steps, self.live_agent_steps = env.episode_end()
        return steps

    def run_generation(self, check_every=16, use_graph=True):
        """One pass of the while(true) body, main_eigen.cpp:113-182; returns (best fitness, steps taken).  The loop
        iteration (controller forward, env.step(), fitness bookkeeping: a dozen small kernels) is captured once into a
        HIP graph and replayed; `use_graph=False` launches it eagerly."""
        venv = self.venv
        population = self.solver.sample()
        self.set_params(population)
        if self.rollout:
            # resetAgent for every candidate, the initial-observation step, prev_track_idx_ (main_eigen.cpp:120-133), then the
            # while (!all_done) loop as an episode of fused launches
            venv.reset(epoch=self.generation)
            steps = self._rollout_episode()
            fitness = venv.fitness.cpu().numpy().astype(np.float64)
            self.solver.tell(population, fitness)
            self.generation += 1
            return float(fitness.max()), steps
        if use_graph and self._graph is None:
            self._graph = venv.capture(self._iteration, warmup=2)
        # resetAgent for every candidate, the initial-observation step, prev_track_idx_ (main_eigen.cpp:120-133)
        venv.reset(epoch=self.generation)
        steps = 0
        while True:
            if use_graph:
                self._graph.replay()
            else:
                self._iteration()
            steps += 1
            # crashed agents neither move nor score, so looking at the flags every few steps changes nothing
            if steps % check_every == 0 and venv.env.alive_count() == 0:
                break
            if self.max_steps is not None and steps >= self.max_steps:
                break
        fitness = venv.fitness.cpu().numpy().astype(np.float64)
        self.solver.tell(population, fitness)
        self.generation += 1
        return float(fitness.max()), steps
