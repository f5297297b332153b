"""The GA checkpoint in the reference's text format (EvolutionaryRacer/Network.hpp:29-51 writeMatrixToFile, :53-80
readMatrixFromFile; MiscUtils.hpp:52-59 agent_weights_{1,2}.txt): openkitchen_amd/csrc/apps/ga_checkpoint.h, host-only.
The format, independently restated here: first line "rows cols", then one line per row, entries separated by single blanks,
each printed as an ostream prints a float (printf's %g: six significant digits); reading parses every entry as a double and
narrows it to float."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def exe():
    out = os.path.join(ROOT, "tests", "cpp", "_build")
    os.makedirs(out, exist_ok=True)
    path = os.path.join(out, "checkpoint_check")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "openkitchen_amd", "csrc", "apps"),
                    "-o", path, os.path.join(ROOT, "tests", "cpp", "checkpoint_check.cpp")], check=True)
    return path


def reference_text(m):
    """What writeMatrixToFile produces for matrix m, restated: operator<<(float) is %g."""
    lines = ["%d %d" % m.shape]
    for row in m:
        lines.append(" ".join("%g" % float(v) for v in row))
    return "\n".join(lines) + "\n"


def test_written_text_is_the_reference_format(exe, tmp_path):
    rng = np.random.default_rng(5)
    m = rng.uniform(-1, 1, size=(17, 30)).astype(np.float32)
    m[0, :6] = [0.0, 1.0, -1.0, 1e-7, 123456.789, -0.5]  # %g's switches: plain zero, integers, exponent form, rounding to 6 digits
    m[1, :3] = [np.float32(3.0e-5), np.float32(1.0e10), np.float32(0.1)]
    f = str(tmp_path / "agent_weights_1.txt")
    subprocess.run([exe, "write", f, "17", "30"], input=m.tobytes(), check=True)
    text = open(f).read()
    assert text == reference_text(m)
    assert text.splitlines()[0] == "17 30" and len(text.splitlines()) == 18 and not text.endswith(" \n")


def test_reading_a_reference_written_file(exe, tmp_path):
    """A file as the reference writes it (restated writer above) comes back as float(double(text)), row-major."""
    rng = np.random.default_rng(6)
    m = rng.uniform(-1, 1, size=(30, 6)).astype(np.float32)
    f = str(tmp_path / "agent_weights_2.txt.safe")
    open(f, "w").write(reference_text(m))
    r = subprocess.run([exe, "read", f], capture_output=True, check=True)
    rows, cols = np.frombuffer(r.stdout[:8], dtype=np.int32)
    got = np.frombuffer(r.stdout[8:], dtype=np.float32).reshape(rows, cols)
    want = np.array([[np.float32(float("%g" % float(v))) for v in row] for row in m], dtype=np.float32)
    assert (rows, cols) == (30, 6) and np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert np.abs(got - m).max() < 1e-6  # six significant digits: the format is lossy, as in the reference


def test_write_read_round_trip_is_idempotent_after_the_first_pass(exe, tmp_path):
    rng = np.random.default_rng(7)
    m = rng.uniform(-1, 1, size=(34, 30)).astype(np.float32)
    f1, f2 = str(tmp_path / "a.txt"), str(tmp_path / "b.txt")
    subprocess.run([exe, "write", f1, "34", "30"], input=m.tobytes(), check=True)
    once = subprocess.run([exe, "read", f1], capture_output=True, check=True).stdout[8:]
    subprocess.run([exe, "write", f2, "34", "30"], input=once, check=True)
    assert open(f1).read() == open(f2).read()


def test_missing_or_malformed_files_are_refused(exe, tmp_path):
    assert subprocess.run([exe, "read", str(tmp_path / "nope.txt")], capture_output=True).returncode == 2
    bad = tmp_path / "bad.txt"
    bad.write_text("3 2\n1 2\n3 4\n5\n")  # one entry short
    assert subprocess.run([exe, "read", str(bad)], capture_output=True).returncode == 2
    bad.write_text("0 5\n")
    assert subprocess.run([exe, "read", str(bad)], capture_output=True).returncode == 2


@pytest.mark.parametrize("R,H", [(15, 30), (32, 30), (5, 7), (64, 32)])
def test_padded_block_round_trip(exe, R, H):
    """weights_1_ ((R+2) x H) and weights_2_ (H x 6) <-> the device's padded per-agent block: padding zero, real entries kept."""
    rng = np.random.default_rng(R * 100 + H)
    w = rng.uniform(-1, 1, size=(R + 2) * H + H * 6).astype(np.float32)
    r = subprocess.run([exe, "pad", str(R), str(H)], input=w.tobytes(), capture_output=True)
    assert r.returncode == 0
    assert np.array_equal(np.frombuffer(r.stdout, dtype=np.float32).view(np.uint32), w.view(np.uint32))
