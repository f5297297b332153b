import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from openkitchen_amd.torch_env import VectorEnvironment
from openkitchen_amd.rollout import PPO_ACTIONS
N = 4096
fan = np.array([-70, -30, 0, 30, 70], dtype=np.float32)
venv = VectorEnvironment("Silverstone", N, ray_angles_deg=fan, auto_reset=True, randomize_lane=True, randomize_heading=True, seed=1, reward="step")
print(venv.env.info())
actor = torch.nn.Sequential(torch.nn.Linear(5, 128), torch.nn.ReLU(), torch.nn.Linear(128, 3), torch.nn.Softmax(dim=1)).cuda()
table = torch.tensor(PPO_ACTIONS, dtype=torch.float32, device="cuda")
venv.reset()
def loop(n):
    with torch.no_grad():
        for _ in range(n):
            probs = torch.clamp(actor(venv.observation()), 1e-8, 1.0 - 1e-8)
            venv.step(table[torch.multinomial(probs, 1).squeeze(1)])
loop(20); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(200); torch.cuda.synchronize(); print("eager us/step", (time.perf_counter() - t0) / 200 * 1e6)
