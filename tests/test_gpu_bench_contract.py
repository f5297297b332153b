"""The line the driver reads: `python bench.py --gpus 1 --steps K --warmup W` prints ONE JSON object with the contract's keys, the
roofline and cpu_baseline objects, and -- this round -- BASELINE configs 3 and 5 and the broad-phase figures."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_run_prints_the_contract_line(gpu):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--repeats", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 20 and j["warmup"] == 5 and j["scaling"] == "weak" and j["vs_baseline"] is None
    assert j["dtype"] == "f32" and j["data"] == "synthetic" and "workload" in j["config"] and "model" not in j["config"]
    assert j["value"] > 1e8 and abs(j["value"] - 4096 * 20 / (j["ms_per_step"] * 20e-3)) / j["value"] < 1e-6
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert rf["launches"] == 4 and rf["avg_launch_ms"] > 0    # HIP events on the kernel's stream, one launch per region
    assert (rf["traffic"] is not None) != bool(j["counter_profile_note"])
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 100 and "sample" in cb
    bp = j["broad_phase"]
    assert 1.0 < bp["s_tested_per_ray"] < 20.0 and bp["segments"] == 4712 and 0 < bp["valu_fraction"] < 1
    for name, n_agents in (("c3", 8192), ("c4_island", 8192), ("c5", 16384)):
        c = j["configs"][name]
        assert c["value"] > c["live_value"] > 1e7 and 0.02 < c["live_fraction"] < 0.6
        assert str(n_agents) in c["workload"] and c["roofline"]["frac"] < 1
