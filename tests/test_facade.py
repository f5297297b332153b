"""The C++ drop-in classes (include/Environment/*.h over the C ABI) against the CPU oracle: a scripted caller in the
shape of the reference's Template/main.cpp runs on the GPU, the same actions run through the oracle, and every
float of every step is compared bit for bit."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def build_facade_test(ok):
    out_dir = os.path.join(HERE, "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "test_facade")
    lib_dir = os.path.dirname(ok.capi.lib_path())
    subprocess.run(["hipcc", "-std=c++17", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(HERE, "cpp", "test_facade.cpp"),
                    "-o", exe, "-L", lib_dir, "-lokenv", "-Wl,-rpath," + lib_dir], check=True)
    return exe


def test_facade_headers_compile_and_link(ok):
    """No GPU needed: the reference-shaped headers compile against a caller and link against libokenv.so."""
    assert os.path.exists(build_facade_test(ok))


@pytest.mark.gpu
def test_facade_matches_oracle(gpu, oracle, tmp_path):
    exe = build_facade_test(gpu)
    N, steps, R = 20, 440, 15
    out = str(tmp_path / "traj.bin")
    track = gpu.track_path("Austin")
    subprocess.run([exe, track, out, str(N), str(steps)], check=True, timeout=300)
    raw = np.fromfile(out, dtype=np.uint32)
    per_agent = 6 + 3 + 2 * R
    frames = steps + 2  # initial step, `steps` steps, the collision-only pass
    main = raw[: frames * N * per_agent].reshape(frames, N, per_agent)
    legacy = raw[frames * N * per_agent:].reshape(3, per_agent)

    t = oracle.Track("Austin")
    fan = np.arange(-70, 71, 10, dtype=np.float32)
    orc = oracle.OracleEnv(t.segments, N, R, fan)
    idx = (np.arange(N) * 37 + 3) % t.P
    orc.reset_agents(np.arange(N), t.x[idx], t.y[idx], t.heading[idx])
    mode = (np.arange(N) % 2).astype(np.uint8)
    orc.set(oracle.F_MODE, mode)
    parked = (np.arange(N) % 5 == 4)

    def check(frame, label):
        s = orc.snapshot()
        f = main[frame]
        for col, key in enumerate(["pos_x", "pos_y", "rot", "speed", "acc"]):
            assert np.array_equal(f[:, col], s[key].view(np.uint32)), (label, key)
        assert np.array_equal(f[:, 6], s["crashed"].astype(np.uint32)), (label, "crashed")
        assert np.array_equal(f[:, 7], s["timed_out"].astype(np.uint32)), (label, "timed_out")
        assert np.array_equal(f[:, 8], s["disp_ctr"]), (label, "disp_ctr")
        hits = f[:, 9:].reshape(N, R, 2)
        assert np.array_equal(hits[..., 0], s["rel_x"].view(np.uint32)), (label, "rel_x")
        assert np.array_equal(hits[..., 1], s["rel_y"].view(np.uint32)), (label, "rel_y")
        return s

    orc.step(1)
    check(0, "initial")
    ids = np.arange(N)
    saw_crash = saw_timeout = False
    for s in range(steps):
        thr = ((ids * 7 + s * 3) % 60).astype(np.float32) * np.where(mode == 1, np.float32(0.25), np.float32(1.0))
        thr[parked] = 0.0
        steer = (((ids * 5 + s) % 7) - 3).astype(np.float32)
        orc.set(oracle.F_THR, thr)
        orc.set(oracle.F_STEER, steer)
        orc.step(1)
        snap = check(s + 1, "step %d" % s)
        saw_crash |= bool(snap["crashed"].any())
        saw_timeout |= bool(snap["timed_out"].any())
        if s == steps // 2:
            who = np.where((snap["crashed"] == 1) & (ids % 2 == 0))[0]
            if who.size:
                orc.reset_agents(who, np.full(who.size, t.x[3]), np.full(who.size, t.y[3]), np.full(who.size, t.heading[3]))
    assert saw_crash and saw_timeout
    orc.collide()
    check(steps + 1, "collision-only pass")

    # legacy constructor scenario: 3 agents, 20 steps
    orc2 = oracle.OracleEnv(t.segments, 3, R, fan)
    orc2.reset_agents([0, 1, 2], [t.x[3]] * 3, [t.y[3]] * 3, [t.heading[0]] * 3)
    i3 = np.arange(3)
    for s in range(20):
        orc2.set(oracle.F_THR, ((i3 * 7 + s * 3) % 60).astype(np.float32))
        orc2.set(oracle.F_STEER, (((i3 * 5 + s) % 7) - 3).astype(np.float32))
        orc2.step(1)
    s2 = orc2.snapshot()
    assert np.array_equal(legacy[:, 0], s2["pos_x"].view(np.uint32)) and np.array_equal(legacy[:, 2], s2["rot"].view(np.uint32))
    assert np.array_equal(legacy[:, 9:].reshape(3, R, 2)[..., 0], s2["rel_x"].view(np.uint32))
