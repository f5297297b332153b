"""Host-side pieces of the CMA-ES / PPO callers (openkitchen_amd/cmaes.py, rollout.py): the solver's constants and
update against the formulas of CovarianceMatrixAdaptationEvolution/CmaEsSolverEigen.cpp:26-132, and the discounted
returns against a loop restatement of RLRacers/PPO/ExperienceBuffer.hpp:47-71.  No GPU involved."""
import importlib.util
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_module(name):
    """cmaes.py / rollout.py import the GPU binding at module level; pull in only the classes under test."""
    import openkitchen_amd  # noqa: F401  (package import never touches the GPU)
    return importlib.import_module("openkitchen_amd." + name)


def test_solver_constants_follow_the_reference_formulas():
    cm = load_module("cmaes")
    n, lam = 250, 20
    s = cm.CmaEsSolver(n, lam, seed=1)
    mu = lam // 2
    w = np.array([np.log(mu + 0.5) - np.log(i + 1) for i in range(mu)])
    w /= w.sum()
    assert np.allclose(s.weights, w) and abs(s.weights.sum() - 1) < 1e-12 and (np.diff(s.weights) < 0).all()
    mu_eff = 1.0 / (w ** 2).sum()
    assert np.isclose(s.mu_eff, mu_eff)
    assert np.isclose(s.c_sigma, (mu_eff + 2) / (n + mu_eff + 5))
    assert np.isclose(s.c_c, (4 + mu_eff / n) / (n + 4 + 2 * mu_eff / n))
    assert np.isclose(s.c_1, 2 / ((n + 1.3) ** 2 + mu_eff))
    assert np.isclose(s.chi_n, np.sqrt(n) * (1 - 1 / (4 * n) + 1 / (21 * n * n)))
    assert s.sigma == 0.5 and s.num_parents == 10
    x = s.sample()
    assert x.shape == (lam, n) and x.dtype == np.float32
    # first generation: C = I, so candidates are mean + sigma * z
    assert 0.4 < x.std() < 0.6


def test_solver_update_step_by_step():
    cm = load_module("cmaes")
    n, lam = 6, 8
    s = cm.CmaEsSolver(n, lam, seed=4)
    x = s.sample().astype(np.float64)
    fit = -np.sum((x - 1.0) ** 2, axis=1)
    old_mean, sigma0, B, D = s.mean.copy(), s.sigma, s.B.copy(), s.D.copy()
    s.tell(x, fit)
    order = sorted(range(lam), key=lambda i: -fit[i])[: lam // 2]
    mean = sum(s.weights[k] * x[i] for k, i in enumerate(order))
    assert np.allclose(s.mean, mean)
    y_w = (mean - old_mean) / sigma0
    p_sigma = np.sqrt(s.c_sigma * (2 - s.c_sigma) * s.mu_eff) * (B @ np.diag(1 / D) @ B.T @ y_w)
    p_c = np.sqrt(s.c_c * (2 - s.c_c) * s.mu_eff) * y_w
    assert np.allclose(s.p_sigma, p_sigma) and np.allclose(s.p_c, p_c)
    rank_mu = sum(s.weights[k] * np.outer((x[i] - old_mean) / sigma0, (x[i] - old_mean) / sigma0) for k, i in enumerate(order))
    C = (1 - s.c_1 - s.c_mu) * np.eye(n) + s.c_1 * np.outer(p_c, p_c) + s.c_mu * rank_mu
    assert np.allclose(s.C, C)
    assert np.isclose(s.sigma, sigma0 * np.exp((s.c_sigma / s.d_sigma) * (np.linalg.norm(p_sigma) / s.chi_n - 1)))


def test_solver_maximises_a_concave_function():
    cm = load_module("cmaes")
    n = 10
    target = np.linspace(-1, 1, n)
    s = cm.CmaEsSolver(n, 24, seed=7)
    for _ in range(150):
        x = s.sample()
        s.tell(x, -np.sum((x - target) ** 2, axis=1))
    assert np.abs(s.get_best_solution() - target).max() < 1e-2


def test_batched_controller_matches_a_torch_module():
    cm = load_module("cmaes")
    torch.manual_seed(0)
    ctrl = cm.BatchedController(5, 16, 2, torch.device("cpu"))
    assert ctrl.count_params() == 5 * 16 + 16 + 16 * 8 + 8 + 8 * 2 + 2 == 250
    mods = []
    for _ in range(3):
        m = torch.nn.Sequential(torch.nn.Linear(5, 16), torch.nn.Tanh(), torch.nn.Linear(16, 8), torch.nn.Tanh(),
                                torch.nn.Linear(8, 2), torch.nn.Tanh())
        mods.append(m)
    flat = torch.stack([torch.cat([p.detach().flatten() for p in m.parameters()]) for m in mods])  # get_flat_params
    ctrl.set_params(flat)
    x = torch.rand(3, 5)
    want = torch.stack([m(x[i]) for i, m in enumerate(mods)])
    assert torch.allclose(ctrl.forward(x), want, atol=1e-6)


def test_discounted_returns_single_agent_equals_the_reference_loop():
    ro = load_module("rollout")
    rewards = torch.ones(37, 1)
    got = ro.discounted_returns(rewards, 0.99, normalize=True)[:, 0]
    disc, run = [0.0] * 37, np.float32(0)
    for i in range(36, -1, -1):  # ExperienceBuffer.hpp:57-64
        run = np.float32(1.0) + np.float32(0.99) * run
        disc[i] = run
    t = torch.tensor(disc)
    want = (t - t.mean()) / (t.std() + torch.finfo(torch.float32).eps)
    assert torch.allclose(got, want, atol=1e-5)
    raw = ro.discounted_returns(torch.ones(5, 3), 0.5, normalize=False)
    assert torch.allclose(raw[:, 1], torch.tensor([1.9375, 1.875, 1.75, 1.5, 1.0]))


def test_solver_device_path_matches_host_update():
    """sample()/tell() with a torch device (here the CPU device) against the numpy path fed the same candidates."""
    cm = load_module("cmaes")
    a = cm.CmaEsSolver(12, 16, seed=3, device=torch.device("cpu"))
    b = cm.CmaEsSolver(12, 16, seed=3)
    for _ in range(5):
        xa = a.sample()
        b.sample()  # advances b's eigendecomposition the same way
        assert torch.is_tensor(xa) and xa.dtype == torch.float32 and xa.shape == (16, 12)
        fit = -((xa.double().numpy() - 0.5) ** 2).sum(axis=1)
        a.tell(xa, fit)
        b.tell(xa.numpy(), fit)
        assert np.allclose(a.mean, b.mean, atol=1e-12) and np.allclose(a.C, b.C, atol=1e-12) and np.isclose(a.sigma, b.sigma)
