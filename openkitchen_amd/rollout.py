"""Episode collection for the policy-gradient callers (RLRacers/PPO/ppo_sim.cpp:46-92, RLRacers/Reinforce) on the
device environment.

The reference runs 15 agents until ALL have crashed, pushing (state, action, log-prob, +1 reward) per agent and step
into one shared buffer -- crashed agents keep contributing their frozen observation.  `collect_episode` is that loop
for N agents with everything kept on the GPU as [T, N, ...] tensors; `alive[t, i]` marks the entries an agent produced
before it crashed, so a learner can mask the frozen tail (or keep it, as the reference does).
"""
import torch

from . import _capi as capi

# PPOAgent::kActionMap (RLRacers/PPO/PPOAgent.hpp:41-43): action index -> (throttle, steering)
PPO_ACTIONS = ((60.0, 0.0), (30.0, 5.0), (30.0, -5.0))


def collect_episode(venv, policy, max_steps=None, check_every=8, actions=PPO_ACTIONS):
    """One pass of the episode loop (ppo_sim.cpp:49-89).

    policy(states [N, R]) -> action probabilities [N, len(actions)] (an Actor::forward, RLRacers/PPO/Actor.hpp:20-27).
    Returns a dict of stacked device tensors: states [T, N, R] (sensor_hits_.norm() / kSensorRange), actions [T, N] i64,
    log_probs [T, N], rewards [T, N] (= 1), alive [T, N] bool.
    """
    assert venv.reward_kind == capi.REWARD_STEP, 'create the VectorEnvironment with reward="step"'
    assert not venv.auto_reset or max_steps is not None, "with auto-reset on the episode never ends: pass max_steps"
    table = torch.tensor(actions, dtype=torch.float32, device=venv.device)
    # resetAgent for every agent + the initial-observation step (ppo_sim.cpp:53-60)
    venv.reset()
    states, acts, logps, rewards, alive = [], [], [], [], []
    steps = 0
    while True:
        state = venv.observation()
        with torch.no_grad():
            probs = torch.clamp(policy(state), 1e-8, 1.0 - 1e-8)        # kProbClamp (PPOAgent.hpp:27,83)
            action = torch.multinomial(probs, 1).squeeze(1)             # :86
            logp = torch.log(probs.gather(1, action.unsqueeze(1))).squeeze(1)
        alive.append(~venv.done.clone())
        states.append(state)
        acts.append(action)
        logps.append(logp)
        venv.step(table[action])
        rewards.append(venv.reward.clone())
        steps += 1
        if steps % check_every == 0 and venv.env.alive_count() == 0:
            break
        if max_steps is not None and steps >= max_steps:
            break
    return {"states": torch.stack(states), "actions": torch.stack(acts), "log_probs": torch.stack(logps),
            "rewards": torch.stack(rewards), "alive": torch.stack(alive)}


def discounted_returns(rewards, gamma=0.99, normalize=True):
    """Reward-to-go along the time axis of a [T, N] reward tensor, then (optionally) the whole-buffer normalisation of
    ExperienceBuffer::calculateDiscountedRewards (RLRacers/PPO/ExperienceBuffer.hpp:47-71).  The reference discounts
    across its single interleaved (step-major, agent-minor) buffer; per agent along time is what that code intends
    and what makes sense for thousands of agents."""
    out = torch.empty_like(rewards)
    running = torch.zeros_like(rewards[0])
    for t in range(rewards.shape[0] - 1, -1, -1):
        running = rewards[t] + gamma * running
        out[t] = running
    if normalize:
        out = (out - out.mean()) / (out.std() + torch.finfo(torch.float32).eps)
    return out
