"""examples/ppo_racer.py runs end to end on the device environment (rollouts on the GPU, PPO update in PyTorch)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_ppo_racer_example_runs(gpu):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "ppo_racer.py"), "--agents", "256", "--episodes", "3",
                          "--max-steps", "600"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300, cwd=ROOT)
    text = out.stdout.decode()
    assert out.returncode == 0, text[-2000:]
    lines = [ln for ln in text.splitlines() if ln.startswith("episode")]
    assert len(lines) == 3
    lengths = [float(ln.split("mean episode length")[1].split("(")[0]) for ln in lines]
    assert all(v > 0 for v in lengths)
