"""bench.py's promise about --gpus N (CPU-only checks: nothing here needs a GPU).  A run either carries n_gpus == N or fails:
under torchrun a WORLD_SIZE different from --gpus is refused before anything is initialised; started plainly with --gpus N > 1
the script starts the N ranks itself and hands their failure on (here they fail because this container has no GPU)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    e.update(env)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True, timeout=600, cwd=ROOT, env=e)


def test_world_size_mismatch_is_refused():
    for world, gpus in (("1", "2"), ("2", "1"), ("4", "8")):
        r = run_bench(["--gpus", gpus], WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        assert r.returncode == 2, (world, gpus, r.returncode, r.stderr[-500:])
        assert "--gpus %s but WORLD_SIZE=%s" % (gpus, world) in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]  # no result line with a wrong n_gpus


def test_gpus_must_be_positive():
    r = run_bench(["--gpus", "0"])
    assert r.returncode != 0 and "--gpus must be >= 1" in r.stderr


def test_plain_start_with_two_gpus_spawns_two_ranks_and_relays_their_exit_code():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("covered with a GPU by tests/test_gpu_two_rank.py")
    r = run_bench(["--gpus", "2", "--steps", "5", "--warmup", "1", "--no-cpu-baseline"])
    assert "starting the ranks" in r.stderr and "--nproc-per-node=2" in r.stderr
    # both ranks stop at "bench.py needs a GPU" (there is no CPU fallback); the launcher must not turn that into success
    assert r.returncode != 0
    assert r.stderr.count("bench.py needs a GPU") >= 1
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
