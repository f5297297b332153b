"""INTEGRATION.md section 2 as a real program: tests/cpp/capi_population.c is plain C (gcc, -std=c11 -pedantic: the
header must be a C header), links libokenv.so and drives the population API; on the GPU its observations and done
flags are compared with the oracle fed the same scripted actions."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def build(ok):
    out_dir = os.path.join(HERE, "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "capi_population")
    lib_dir = os.path.dirname(ok.capi.lib_path())
    subprocess.run(["gcc", "-std=c11", "-pedantic", "-Wall", "-Werror", "-O2", "-I", os.path.join(ROOT, "include"),
                    os.path.join(HERE, "cpp", "capi_population.c"), "-o", exe, "-L", lib_dir, "-lokenv", "-Wl,-rpath," + lib_dir,
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return exe


def test_c_program_compiles_as_c11(ok):
    assert os.path.exists(build(ok))


@pytest.mark.gpu
def test_c_program_matches_oracle(gpu, oracle, tmp_path):
    exe = build(gpu)
    N, steps, R = 48, 200, 16
    out = str(tmp_path / "obs.bin")
    subprocess.run([exe, gpu.track_path("Monza"), str(N), str(steps), out], check=True, timeout=120)
    raw = np.fromfile(out, dtype=np.uint8)
    frame_bytes = N * R * 4 + N
    frames = raw.reshape(-1, frame_bytes)
    t = oracle.Track("Monza")
    fan = (np.float32(-70.0) + np.float32(140.0) * np.arange(R, dtype=np.float32) / np.float32(R - 1)).astype(np.float32)
    orc = oracle.OracleEnv(t.segments, N, R, fan)
    k = (np.arange(N) * 37 + 3) % t.P
    orc.reset_agents(np.arange(N), t.x[k], t.y[k], t.heading[k])
    i = np.arange(N)
    f = 0
    for s in range(steps):
        orc.set(oracle.F_THR, (20.0 + ((i * 7 + s) % 60)).astype(np.float32))
        orc.set(oracle.F_STEER, (((i + 3 * s) % 11) - 5.0).astype(np.float32))
        orc.step(1)
        if s % 10 == 9 or s == steps - 1:
            snap = orc.snapshot()
            want = np.concatenate([snap["dist"].view(np.uint8).reshape(-1),
                                   (snap["crashed"] | (snap["timed_out"] << 1)).astype(np.uint8)])
            assert np.array_equal(frames[f], want), "frame %d (step %d)" % (f, s)
            f += 1
    assert f == len(frames) and snap["crashed"].any()
