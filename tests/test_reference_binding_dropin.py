"""Source-level drop-in of the reference's Python binding: oracle/Makefile's `refbind` target compiles the reference's
own Pybind/bindings.cpp, unchanged and from where it lies, against this project's include/Environment/ headers and links
it to libokenv.so (output oracle/_ref/open_kitchen_pybind*.so, git-ignored).  Where that module exists it must import
and expose the reference's API; on a GPU it must run the reference's example loop (Pybind/example.py:10-20).

The reference's applications built the same way (`make refapps`) run on the GPU with a step trace switched on in the
facade (OKENV_TRACE_FILE): every Environment::step / checkCollision call they make -- real policies, real resets -- is then
replayed on the CPU oracle and must give the same bits."""
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


RECORD = np.dtype([("pos_x", "f4"), ("pos_y", "f4"), ("rot", "f4"), ("speed", "f4"), ("acc", "f4"), ("throttle", "f4"), ("steer", "f4"),
                   ("disp_x", "f4"), ("disp_y", "f4"), ("disp_ctr", "u4"), ("mode", "u1"), ("crashed", "u1"), ("timed_out", "u1"),
                   ("disp_to", "u1")])  # okenv_agent_record (include/okenv.h)


def read_trace(path):
    """Step trace written by the facade (OKENV_TRACE_FILE): per Environment::step / checkCollision the records handed in,
    the records handed back and sensor_hits_."""
    raw = np.fromfile(path, dtype=np.uint8) if os.path.exists(path) else np.zeros(0, dtype=np.uint8)
    out, off = [], 0
    while off + 20 <= raw.size:
        magic, n, r, flags = (int(v) for v in raw[off:off + 16].view(np.uint32))
        assert magic == 0x4F4B5452
        size = 20 + 4 * r + 2 * n * RECORD.itemsize + 8 * n * r
        if off + size > raw.size:
            break  # the application was stopped in the middle of a record
        p = off + 16
        offset = raw[p:p + 4].view(np.float32)[0]
        fan = raw[p + 4:p + 4 + 4 * r].view(np.float32).copy()
        p += 4 + 4 * r
        rin = raw[p:p + n * RECORD.itemsize].view(RECORD).copy()
        p += n * RECORD.itemsize
        rout = raw[p:p + n * RECORD.itemsize].view(RECORD).copy()
        p += n * RECORD.itemsize
        hits = raw[p:p + 8 * n * r].view(np.float32).reshape(n, r, 2).copy()
        out.append((flags, offset, fan, rin, rout, hits))
        off += size
    return out


def replay_on_oracle(oracle, track_name, trace, limit):
    """Every traced call again on the CPU oracle, from the state the application handed in: what came back must be the
    oracle's result bit for bit (pose, speed, flags, DisplacementStats, sensor_hits_).  World hit points of crashed agents
    live on from call to call, on the device and in the oracle alike, so the calls are replayed in order."""
    t = oracle.Track(track_name)
    flags0, _, fan, rin0, _, _ = trace[0]
    n, r = rin0.size, fan.size
    orc = oracle.OracleEnv(t.segments, n, r, fan, (t.x, t.y, t.heading))
    O = oracle
    crashes = timeouts = moved = 0
    for k, (flags, offset, fan_k, rin, rout, hits) in enumerate(trace[:limit]):
        assert np.array_equal(fan_k, fan) and rin.size == n
        O.lib().oracle_env_set_sensor_offset(orc.h, float(offset))
        for f, key in ((O.F_POS_X, "pos_x"), (O.F_POS_Y, "pos_y"), (O.F_ROT, "rot"), (O.F_SPEED, "speed"), (O.F_ACC, "acc"),
                       (O.F_THR, "throttle"), (O.F_STEER, "steer"), (O.F_MODE, "mode"), (O.F_CRASHED, "crashed"),
                       (O.F_TIMED_OUT, "timed_out")):
            orc.set(f, np.ascontiguousarray(rin[key]))
        if flags & 1:  # OKENV_PACKED_WITH_STATS
            for f, key in ((O.F_DISP_CTR, "disp_ctr"), (O.F_DISP_X, "disp_x"), (O.F_DISP_Y, "disp_y"), (O.F_DISP_TO, "disp_to")):
                orc.set(f, np.ascontiguousarray(rin[key]))
        if flags & 2:  # OKENV_PACKED_COLLIDE_ONLY
            orc.collide()
        else:
            orc.step(1)
        o = orc.snapshot()
        for key, okey in (("pos_x", "pos_x"), ("pos_y", "pos_y"), ("rot", "rot"), ("speed", "speed"), ("acc", "acc")):
            assert np.array_equal(rout[key].view(np.uint32), o[okey].view(np.uint32)), (k, key)
        assert np.array_equal(rout["crashed"], o["crashed"]) and np.array_equal(rout["timed_out"], o["timed_out"]), k
        if flags & 1:
            assert np.array_equal(rout["disp_ctr"], o["disp_ctr"]) and np.array_equal(rout["disp_to"], o["disp_to"]), k
            assert np.array_equal(rout["disp_x"].view(np.uint32), o["disp_x"].view(np.uint32)), k
        assert np.array_equal(hits[..., 0].view(np.uint32), o["rel_x"].reshape(n, r).view(np.uint32)), k
        assert np.array_equal(hits[..., 1].view(np.uint32), o["rel_y"].reshape(n, r).view(np.uint32)), k
        crashes += int(((rout["crashed"] == 1) & (rin["crashed"] == 0) & (rout["timed_out"] == 0)).sum())
        timeouts += int(((rout["timed_out"] == 1) & (rin["timed_out"] == 0)).sum())
        moved += int((rout["pos_x"] != rin["pos_x"]).sum())
    return {"calls": min(limit, len(trace)), "agents": n, "rays": r, "crashes": crashes, "timeouts": timeouts, "moved": moved}


def run_app(name, track, until, timeout=60, cwd=None, may_finish=False, trace=None):
    """Runs a reference application built by `make -C oracle refapps` until `until(output_text)` holds (most of them loop
    forever), then stops it; skips when the binary was not built.  `trace`: path for the facade's step trace."""
    import subprocess
    import time
    exe = os.path.join(ROOT, "oracle", "_ref", name)
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs /root/reference at build time)" % exe)
    env = dict(os.environ)
    # libtorch's intra-op pool would start a thread per host core for these tiny networks; their spin-waiting eats a container's
    # CPU quota (the whole process is then paused until the next scheduler period)
    env.setdefault("OMP_NUM_THREADS", "4")
    if trace:
        env.update(OKENV_TRACE_FILE=trace, OKENV_TRACE_STEPS="4000")
    log = (trace or os.path.join(cwd or "/tmp", name)) + ".log"
    with open(log, "wb") as fh:
        p = subprocess.Popen([exe, track], stdout=fh, stderr=subprocess.STDOUT, cwd=cwd, env=env)
        t0 = time.time()
        try:
            while time.time() - t0 < timeout:
                text = open(log, errors="ignore").read()
                if p.poll() is not None:
                    if may_finish and p.returncode == 0:
                        return text
                    pytest.fail("the application ended on its own (rc %s):\n%s" % (p.returncode, text[-2000:]))
                if until(text):
                    return text
                time.sleep(0.2)
            pytest.fail("condition not reached in %d s:\n%s" % (timeout, open(log, errors="ignore").read()[-2000:]))
        finally:
            if p.poll() is None:
                p.kill()
                p.wait()


@pytest.mark.gpu
def test_reference_template_app_runs_on_the_device_environment(gpu, oracle, tmp_path):
    """Template/main.cpp, unchanged: one agent, resetAgent + step in a loop, episodes end by crash or standstill.  Every
    Environment::step the application made is replayed on the oracle."""
    trace = str(tmp_path / "template.trace")
    out = run_app("template_main", gpu.track_path("Austin"), lambda text: text.count("EPISODE") >= 4, trace=trace)
    assert out.count("EPISODE") >= 4, out[-1500:]
    stats = replay_on_oracle(oracle, "Austin", read_trace(trace), 1500)
    assert stats["calls"] >= 200 and stats["agents"] == 1 and stats["moved"] > 100, stats
    assert stats["crashes"] + stats["timeouts"] >= 2, stats


@pytest.mark.gpu
def test_reference_cmaes_app_runs_on_the_device_environment(gpu, oracle, tmp_path):
    """CovarianceMatrixAdaptationEvolution/main_torch.cpp + its solver and controller, unchanged: 20 candidates per
    generation driven through env.step() / findNearestTrackIndexBruteForce of this project's Environment."""
    import re
    trace = str(tmp_path / "cma.trace")
    out = run_app("cma_main_torch", gpu.track_path("Austin"), lambda text: len(re.findall(r"Generation \d+ Best Fitness", text)) >= 2,
                  trace=trace)
    best = [float(x) for x in re.findall(r"Generation \d+ Best Fitness: ([0-9.eE+-]+)", out)]
    assert len(best) >= 2 and max(best) > 0, out[-1500:]
    stats = replay_on_oracle(oracle, "Austin", read_trace(trace), 400)
    assert stats["calls"] >= 100 and stats["agents"] >= 1 and stats["moved"] > 50, stats


@pytest.mark.gpu
def test_reference_ddpg_app_runs_on_the_device_environment(gpu, oracle, tmp_path):
    """RLRacers/DDPG/ddpg_sim.cpp + DDPGAgent.hpp, unchanged (its libtorch learner trains on the CPU between steps)."""
    trace = str(tmp_path / "ddpg.trace")
    out = run_app("ddpg_sim", gpu.track_path("Austin"), lambda text: text.count("EPISODE") >= 2, trace=trace)
    assert out.count("EPISODE") >= 2, out[-1500:]
    stats = replay_on_oracle(oracle, "Austin", read_trace(trace), 600)
    assert stats["calls"] >= 50 and stats["moved"] > 20, stats


@pytest.mark.gpu
def test_reference_data_collector_runs_on_the_device_environment(gpu, oracle, tmp_path):
    """FieldNavigators/collect_data/collect_data_random.cpp, unchanged: resetAgent with lane and heading randomisation,
    a potential-field driver, one measurement file per step.  It looks for SaoPaulo.csv in a folder and writes into
    ./SaoPaulo_random."""
    import shutil
    tracks = tmp_path / "tracks"
    tracks.mkdir()
    shutil.copy(gpu.track_path("Austin"), str(tracks / "SaoPaulo.csv"))
    trace = str(tmp_path / "collect.trace")
    folder = str(tmp_path / "SaoPaulo_random")
    enough = lambda text: (os.path.isdir(folder) and len(os.listdir(folder)) > 20 and os.path.exists(trace)  # noqa: E731
                           and os.path.getsize(trace) > 80000)
    out = run_app("collect_data_random", str(tracks), enough, cwd=str(tmp_path), may_finish=True, trace=trace)  # ends on its own after 200 goals
    files = os.listdir(folder)
    assert len(files) > 20, out[-1500:]
    texts = sorted(f for f in files if f.endswith(".txt"))
    images = [f for f in files if f.endswith(".png")]
    assert texts and len(images) >= len(texts) - 1  # bird's-eye mode: an action file and a frame (Environment::saveImage) per sample
    assert open(os.path.join(folder, images[0]), "rb").read(8) == b"\x89PNG\r\n\x1a\n"
    throttle, steer = open(os.path.join(folder, texts[0])).read().split()
    assert abs(float(steer)) <= 10.0  # kSteeringAngleClampDeg
    stats = replay_on_oracle(oracle, "Austin", read_trace(trace), 600)
    assert stats["calls"] >= 100 and stats["moved"] > 50, stats


@pytest.mark.gpu
@pytest.mark.parametrize("app", ["ppo_sim", "reinforce_sim"])
def test_reference_policy_gradient_apps_run_on_the_device_environment(gpu, oracle, tmp_path, app):
    """RLRacers/PPO/ppo_sim.cpp (+ PPOAgent.hpp, Actor/Critic, ExperienceBuffer) and RLRacers/Reinforce/reinforce_sim.cpp,
    unchanged: 15 agents (PPO) / one agent (REINFORCE), resetAgent to random points, +1 reward per step, a libtorch update per
    episode."""
    trace = str(tmp_path / (app + ".trace"))
    out = run_app(app, gpu.track_path("Austin"), lambda text: text.count("EPISODE") >= 2, trace=trace)
    assert out.count("EPISODE") >= 2, out[-1500:]
    stats = replay_on_oracle(oracle, "Austin", read_trace(trace), 600)
    assert stats["calls"] >= 50 and stats["agents"] == (15 if app == "ppo_sim" else 1) and stats["moved"] > 200, stats


@pytest.mark.gpu
def test_reference_q_racer_sim_runs_on_the_device_environment(gpu, oracle, tmp_path):
    """RLRacers/Q_Learning/q_racer_sim.cpp + QAgent.hpp, unchanged -- one of the two callers the north star names: 30
    tabular Q-learning agents x 5 rays through the legacy `Environment env(path); env.setAgent(&agent)` API, raylib's names
    served headless by include/Environment/.  Every Environment::step it makes is replayed on the oracle."""
    trace = str(tmp_path / "q_racer.trace")
    out = run_app("q_racer_sim", gpu.track_path("Austin"), lambda text: text.count("EPISODE") >= 4, trace=trace)
    assert out.count("EPISODE") >= 4 and "eps: 0.9" in out, out[-1500:]
    stats = replay_on_oracle(oracle, "Austin", read_trace(trace), 1500)
    assert stats["calls"] >= 100 and stats["agents"] == 30 and stats["rays"] == 5, stats
    assert stats["crashes"] >= 30 and stats["moved"] > 1000, stats  # every agent crashed at least once per finished episode
