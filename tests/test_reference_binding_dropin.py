"""Source-level drop-in of the reference's Python binding: oracle/Makefile's `refbind` target compiles the reference's
own Pybind/bindings.cpp, unchanged and from where it lies, against this project's include/Environment/ headers and links
it to libokenv.so (output oracle/_ref/open_kitchen_pybind*.so, git-ignored).  Where that module exists it must import
and expose the reference's API; on a GPU it must run the reference's example loop (Pybind/example.py:10-20)."""
import glob
import importlib.util
import os

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_module(ok):
    paths = glob.glob(os.path.join(ROOT, "oracle", "_ref", "open_kitchen_pybind*.so"))
    if not paths:
        pytest.skip("oracle/_ref/open_kitchen_pybind*.so not built (needs /root/reference at build time)")
    ok.capi.load()  # libokenv.so first, so that its HIP runtime is the one torch already loaded
    spec = importlib.util.spec_from_file_location("open_kitchen_pybind", paths[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_reference_bindings_compile_against_our_headers(ok, oracle):
    mod = load_module(ok)
    assert {"set_action", "step", "get_render_target", "get_render_target_info"} <= set(dir(mod.Environment))
    assert {"width", "height", "channels", "row_bytes"} <= set(dir(mod.RenderTargetInfo))


@pytest.mark.gpu
def test_reference_example_loop_runs(gpu):
    mod = load_module(gpu)
    env = mod.Environment(gpu.track_path("Austin"), draw_rays=False, hidden_window=True)
    for _ in range(50):
        env.step()
        env.set_action(30.0, 0.0)
    info = env.get_render_target_info()
    assert (info.width, info.height, info.channels) == (1600, 1400, 4) and info.row_bytes() == 6400
    img = np.frombuffer(bytes(env.get_render_target()), dtype=np.uint8)
    assert img.size == info.height * info.row_bytes()  # headless: the render target is blank


def run_app(name, track, seconds, cwd=None, may_finish=False):
    """Runs a reference application built by `make -C oracle refapps` for a while (most of them loop forever) and
    returns its output; skips when the binary was not built."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs /root/reference at build time)" % exe)
    try:
        done = subprocess.run([exe, track], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=seconds, cwd=cwd)
        if may_finish and done.returncode == 0:
            return done.stdout.decode()
        pytest.fail("the application ended on its own (rc %d):\n%s" % (done.returncode, done.stdout.decode()[-2000:]))
    except subprocess.TimeoutExpired as e:
        return (e.stdout or b"").decode()


@pytest.mark.gpu
def test_reference_template_app_runs_on_the_device_environment(gpu):
    """Template/main.cpp, unchanged: one agent, resetAgent + step in a loop, episodes end by crash or standstill."""
    out = run_app("template_main", gpu.track_path("Austin"), 10)
    episodes = out.count("EPISODE")
    assert episodes >= 3, out[-1500:]


@pytest.mark.gpu
def test_reference_cmaes_app_runs_on_the_device_environment(gpu):
    """CovarianceMatrixAdaptationEvolution/main_torch.cpp + its solver and controller, unchanged: 20 candidates per
    generation driven through env.step() / findNearestTrackIndexBruteForce of this project's Environment."""
    import re
    out = run_app("cma_main_torch", gpu.track_path("Austin"), 25)
    best = [float(x) for x in re.findall(r"Generation \d+ Best Fitness: ([0-9.eE+-]+)", out)]
    assert len(best) >= 2, out[-1500:]
    assert max(best) > 0


@pytest.mark.gpu
def test_reference_ddpg_app_runs_on_the_device_environment(gpu):
    """RLRacers/DDPG/ddpg_sim.cpp + DDPGAgent.hpp, unchanged (its libtorch learner trains on the CPU between steps)."""
    out = run_app("ddpg_sim", gpu.track_path("Austin"), 15)
    assert out.count("EPISODE") >= 2, out[-1500:]


@pytest.mark.gpu
def test_reference_data_collector_runs_on_the_device_environment(gpu, tmp_path):
    """FieldNavigators/collect_data/collect_data_random.cpp, unchanged: resetAgent with lane and heading randomisation,
    a potential-field driver, one measurement file per step.  It looks for SaoPaulo.csv in a folder and writes into
    ./SaoPaulo_random."""
    import shutil
    tracks = tmp_path / "tracks"
    tracks.mkdir()
    shutil.copy(gpu.track_path("Austin"), str(tracks / "SaoPaulo.csv"))
    out = run_app("collect_data_random", str(tracks), 60, cwd=str(tmp_path), may_finish=True)  # ends after 200 goals
    files = os.listdir(str(tmp_path / "SaoPaulo_random"))
    assert len(files) > 50, out[-1500:]
    throttle, steer = open(str(tmp_path / "SaoPaulo_random" / sorted(files)[0])).read().split()
    assert abs(float(steer)) <= 10.0  # kSteeringAngleClampDeg


@pytest.mark.gpu
@pytest.mark.parametrize("app", ["ppo_sim", "reinforce_sim"])
def test_reference_policy_gradient_apps_run_on_the_device_environment(gpu, app):
    """RLRacers/PPO/ppo_sim.cpp (+ PPOAgent.hpp, Actor/Critic, ExperienceBuffer) and RLRacers/Reinforce/reinforce_sim.cpp,
    unchanged: 15 agents, resetAgent to random points, +1 reward per step, a libtorch update per episode."""
    out = run_app(app, gpu.track_path("Austin"), 25)
    assert out.count("EPISODE") >= 2, out[-1500:]
