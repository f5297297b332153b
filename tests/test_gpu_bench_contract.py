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
    # the HBM roofline BASELINE.json words its target in is always there ...
    assert rf["hbm_unit"] == "GB/s" and rf["hbm_peak"] == 8000.0 and abs(rf["hbm_frac"] - rf["hbm_achieved"] / rf["hbm_peak"]) < 1e-12
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9
    # ... and `bound` names what binds: the VALU (issue share x lane utilisation) while the committed counters are this kernel's,
    # the HBM figures otherwise (no stale counters)
    if j["counter_profile_note"]:
        assert rf["bound"] == "hbm" and rf["traffic"] is None and rf["frac"] == rf["hbm_frac"] and j["valu_roofline"] is None
    else:
        assert rf["bound"] == "valu" and rf["unit"] == "Tlane-inst/s" and rf["traffic"] is not None
        assert abs(rf["frac"] - rf["valu_issue_frac"] * rf["valu_lane_utilisation"]) < 1e-12 and 0.05 < rf["frac"] < 1
        assert j["valu_roofline"]["simds"] == 4 * 256 and 70 < rf["peak"] < 90  # an MI355X in SPX mode: 256 CUs, the device says so
    assert rf["launches"] == 4 and rf["avg_launch_ms"] > 0    # HIP events on the kernel's stream, one launch per region
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 100 and "sample" in cb
    bp = j["broad_phase"]
    assert 1.0 < bp["s_tested_per_ray"] < 20.0 and bp["segments"] == 4712 and 0 < bp["valu_fraction"] < 1
    for name, n_agents in (("c3", 8192), ("c4_island", 8192), ("c5", 16384)):
        c = j["configs"][name]
        # value = the live rate (agents that entered their step alive); the reference loop's agents x steps is nominal_value
        assert c["nominal_value"] > c["value"] == c["live_value"] > 1e7 and 0.02 < c["live_fraction"] < 0.6
        assert abs(c["value"] / c["nominal_value"] - c["live_fraction"]) < 1e-9
        assert str(n_agents) in c["workload"] and c["roofline"]["frac"] < 1
    isl = j["configs"]["c4_island"]
    assert isl["dist"]["backend"] == "nccl" and isl["dist"]["world_size"] == 1 and len(isl["dist"]["all_gather_us"]) == 2
    for k in ("alive_at_end", "off_grid_alive", "off_grid_agents"):
        assert len(isl[k]) == 2 and len(j["configs"]["c3"][k]) == 2
    # a generation that runs into the 4000-step cap does so because of agents that left the track for good, and the line says so
    for steps, alive, off_alive in zip(isl["steps"], isl["alive_at_end"], isl["off_grid_alive"]):
        assert (steps == 4000) == (alive > 0) and off_alive <= alive
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines  # stdout is the line and nothing else
