"""PPO racers on the device environment: the reference's RLRacers/PPO app (ppo_sim.cpp + PPOAgent.hpp) for thousands of
agents, with the environment, resets and reward bookkeeping on the GPU and the learner in PyTorch-ROCm.

    python examples/ppo_racer.py [--agents 1024] [--episodes 20] [--track Silverstone]

Per episode (ppo_sim.cpp:49-89): resetAgent to random centre-line points, one observation step, then act / step until
every agent has crashed; then PPOAgent::updatePolicy (PPOAgent.hpp:106-160): discounted returns (gamma 0.99,
normalised), 5 epochs of clipped-surrogate actor updates and MSE critic updates, Adam 3e-4.  Differences from the
reference, both forced by scale: returns are discounted per agent along time (the reference discounts across its
interleaved 15-agent buffer) and minibatches are 4096 samples (the reference uses 64).
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from openkitchen_amd.rollout import collect_episode, discounted_returns  # noqa: E402
from openkitchen_amd.torch_env import VectorEnvironment  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=1024)
    ap.add_argument("--episodes", type=int, default=20)
    ap.add_argument("--track", default="Silverstone")
    ap.add_argument("--max-steps", type=int, default=3000)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    torch.manual_seed(args.seed)
    rays = np.array([-70, -30, 0, 30, 70], dtype=np.float32)          # PPOAgent.hpp:56-61
    venv = VectorEnvironment(args.track, args.agents, ray_angles_deg=rays, auto_reset=False, seed=args.seed, reward="step")
    actor = torch.nn.Sequential(torch.nn.Linear(5, 128), torch.nn.ReLU(), torch.nn.Linear(128, 3), torch.nn.Softmax(dim=1)).cuda()
    critic = torch.nn.Sequential(torch.nn.Linear(5, 128), torch.nn.ReLU(), torch.nn.Linear(128, 1)).cuda()
    opt_a = torch.optim.Adam(actor.parameters(), lr=3e-4)             # kLearningRate
    opt_c = torch.optim.Adam(critic.parameters(), lr=3e-4)
    clip, epochs, batch = 0.2, 5, 4096
    for episode in range(args.episodes):
        t0 = time.perf_counter()
        ep = collect_episode(venv, actor, max_steps=args.max_steps)
        alive = ep["alive"]
        lengths = alive.sum(dim=0).float()
        returns = discounted_returns(ep["rewards"] * alive)          # reward only while the agent is driving
        mask = alive.reshape(-1)
        states = ep["states"].reshape(-1, 5)[mask]
        actions = ep["actions"].reshape(-1, 1)[mask]
        old_logp = ep["log_probs"].reshape(-1, 1)[mask]
        ret = returns.reshape(-1, 1)[mask]
        t1 = time.perf_counter()
        for _ in range(epochs):
            perm = torch.randperm(states.shape[0], device=states.device)
            for i in range(0, states.shape[0], batch):
                j = perm[i:i + batch]
                values = critic(states[j])
                adv = ret[j] - values.detach()
                probs = torch.clamp(actor(states[j]), 1e-8, 1 - 1e-8)
                ratio = torch.exp(torch.log(probs.gather(1, actions[j])) - old_logp[j])
                actor_loss = -torch.min(ratio * adv, torch.clamp(ratio, 1 - clip, 1 + clip) * adv).mean()
                critic_loss = torch.nn.functional.mse_loss(values, ret[j])
                opt_a.zero_grad()
                actor_loss.backward()
                opt_a.step()
                opt_c.zero_grad()
                critic_loss.backward()
                opt_c.step()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print("episode %3d: %5d steps, mean episode length %7.1f (max %5d), %7d samples, rollout %.2f s, update %.2f s" % (
            episode, ep["states"].shape[0], float(lengths.mean()), int(lengths.max()), states.shape[0], t1 - t0, t2 - t1), flush=True)
    return float(lengths.mean())


if __name__ == "__main__":
    main()
