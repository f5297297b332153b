"""The front / back split of the segment set on the GPU (openkitchen_amd/csrc/ok_grid.h: okClassifyFrontBack; the cooperative,
tail and direct forms of the step kernel): the outer boundary polylines sit in an image of their own and are walked only by rays
that need them.  Every other GPU test runs with the split as well (it is the default) and compares with the oracle; here: the
same library on the combined image (OKENV_FRONT_BACK=0) must give the same bits, the statistics say where it pays, agents put
where it must NOT be used -- in the 3 px strip between the inner and the outer boundary, outside the track -- equal the oracle."""
import os
import subprocess
import sys

import numpy as np
import pytest

from test_gpu_parity import assert_same_state, make_pair

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_split_exists_on_the_config_tracks_and_not_on_other_segment_sets(gpu):
    for name in ("Austin", "Silverstone", "Monza", "Spa"):
        t = gpu.Track(name)
        env = gpu.BatchedEnvironment.from_track(t, 64, 16)
        info = env.info()
        assert info["front_back_bytes"] > info["lds_bytes"] and info["front_back_bytes"] < 160 * 1024 - 16 * 1024
        assert info["back_segments"] > 0.45 * t.S
        env.close()
    rng = np.random.default_rng(0)
    star = np.concatenate([rng.uniform(100, 900, (64, 2)), rng.uniform(100, 900, (64, 2))], axis=1).astype(np.float32)
    env = gpu.BatchedEnvironment(star, 16, gpu.default_ray_fan(8))
    assert env.info()["front_back_bytes"] == 0
    with pytest.raises(RuntimeError, match="no front / back split"):
        env.work_stats_split()
    env.close()


CHILD = r'''
import sys, numpy as np
sys.path.insert(0, %(root)r)
import openkitchen_amd as ok
t = ok.Track("Monza")
out = {}
for N, R in ((96, 32), (40, 5), (700, 16)):
    env = ok.BatchedEnvironment.from_track(t, N, R)
    env.init_bench_state(0, 0)
    for c in range(5):
        env.rollout_random(60, 77, 0, 60 * c)
    s = env.snapshot()
    for k in ("pos_x", "pos_y", "rot", "crashed", "timed_out", "disp_ctr", "hit_x", "hit_y", "rel_x", "rel_y", "dist"):
        out["%%d_%%s" %% (N, k)] = np.ascontiguousarray(s[k])
    out["%%d_fb" %% N] = np.array([env.info()["front_back_bytes"]])
    env.close()
np.savez(sys.argv[1], **out)
'''


def test_same_bits_with_and_without_the_split(gpu, tmp_path):
    files = []
    for fbv in ("1", "0"):
        f = str(tmp_path / ("fb%s.npz" % fbv))
        env = dict(os.environ, OKENV_FRONT_BACK=fbv)
        r = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), f], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        files.append(np.load(f))
    a, b = files
    assert a["96_fb"][0] > 0 and b["96_fb"][0] == 0
    for k in a.files:
        if not k.endswith("_fb"):
            assert np.array_equal(a[k].view(np.uint8), b[k].view(np.uint8)), k


def test_agents_where_the_back_image_is_needed(gpu, oracle):
    """Agents placed in the strip between inner and outer boundary, just outside the outer one, far outside, and on the track:
    their first steps (nobody has crashed yet when the first observation is made) against the oracle, ray by ray."""
    t, dev, orc = make_pair(gpu, oracle, "Silverstone", 256, 32)
    rng = np.random.default_rng(5)
    seg = t.segments
    sel = rng.integers(0, t.S, 256)
    a, b = seg[sel, 0:2].astype(np.float64), seg[sel, 2:4].astype(np.float64)
    d = b - a
    nrm = np.stack([-d[:, 1], d[:, 0]], axis=1) / np.maximum(np.hypot(d[:, 0], d[:, 1]), 1e-9)[:, None]
    off = rng.choice([-8.0, -3.5, -2.5, -1.5, 1.5, 2.5, 3.5, 8.0, 30.0, 400.0], 256)[:, None]
    p = a + rng.uniform(0, 1, 256)[:, None] * d + off * nrm
    rot = rng.uniform(-180, 180, 256).astype(np.float32)
    idx = np.arange(256, dtype=np.int32)
    for e in (dev, orc):
        e.reset_agents(idx, p[:, 0].astype(np.float32), p[:, 1].astype(np.float32), rot)
    thr = rng.uniform(0, 60, 256).astype(np.float32)
    steer = rng.uniform(-3, 3, 256).astype(np.float32)
    dev.set_actions(thr, steer)
    orc.set(oracle.F_THR, thr)
    orc.set(oracle.F_STEER, steer)
    for step in range(12):
        dev.step(1)
        orc.step(1)
        assert_same_state(dev.snapshot(), orc.snapshot(), "step %d" % step)
    ws = dev.work_stats_split()
    assert ws["rays"] > 0 and 0 < ws["back_walked"] <= ws["rays"] and ws["certified"] < ws["rays"]
    dev.close()


def test_where_it_pays(gpu):
    """The bench recipe's population: nearly every origin certified, about half the points per ray of the combined image."""
    t = gpu.Track("Silverstone")
    env = gpu.BatchedEnvironment.from_track(t, 4096, 64)
    env.init_bench_state(0, 0)
    env.rollout_random(300, 1234, 0, 0)
    s, c = env.work_stats_split(), env.work_stats()
    assert s["rays"] == c["rays"] > 100000
    assert s["certified"] > 0.97 * s["rays"] and s["back_walked"] < 0.03 * s["rays"] and s["ambiguous"] < 0.002 * s["rays"]
    assert s["points"] < 0.65 * c["points"] and s["tests"] < 0.75 * c["tests"]
    env.close()
