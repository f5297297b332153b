"""CPU check of the PRODUCT's traversal header (openkitchen_amd/csrc/ok_raycast.h + ok_grid.h, compiled for the
host by tests/cpp/grid_check.cpp): the uniform-grid walk returns the same first-hit bits as the oracle's
brute-force sweep, for realistic and adversarial rays, for several cell sizes."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

import _oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def gridcheck():
    src = os.path.join(HERE, "cpp", "grid_check.cpp")
    out_dir = os.path.join(HERE, "cpp", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libgridcheck.so")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, src], check=True)
    G = C.CDLL(so)
    G.gridcheck_cast.argtypes = [O.f32p, C.c_int, C.c_float, O.f32p, O.f32p, O.f32p, C.c_int, O.f32p, O.u32p, O.u32p, O.i32p, C.c_int, O.u32p]
    return G


def brute(t, ox, oy, ang):
    L = O.lib()
    seg = np.ascontiguousarray(t.segments.reshape(-1))
    return np.array([L.oracle_cast_ray(float(ox[i]), float(oy[i]), float(ang[i]), seg, t.S) for i in range(ox.size)], dtype=np.float32)


def make_rays(t, n, seed):
    rng = np.random.default_rng(seed)
    idx = rng.integers(0, t.P, n)
    ox = (t.x[idx] + rng.normal(0, 10, n)).astype(np.float32)
    oy = (t.y[idx] + rng.normal(0, 10, n)).astype(np.float32)
    ang = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    k = n // 5
    sel = rng.integers(0, t.S, k)
    ox[:k], oy[:k] = t.segments[sel, 0], t.segments[sel, 1]  # exactly on segment start points
    ang[:k // 2] = np.arctan2(t.segments[sel[:k // 2], 3] - t.segments[sel[:k // 2], 1],
                              t.segments[sel[:k // 2], 2] - t.segments[sel[:k // 2], 0]).astype(np.float32)  # along them
    ox[k:2 * k] = rng.uniform(-400, 2000, k).astype(np.float32)  # anywhere, including far outside the grid
    oy[k:2 * k] = rng.uniform(-400, 1800, k).astype(np.float32)
    ang[2 * k:2 * k + 8] = np.array([0, np.pi / 2, np.pi, -np.pi / 2, 1e-9, np.pi / 2 + 1e-7, -0.0, np.pi / 4], dtype=np.float32)
    return ox, oy, ang


@pytest.mark.parametrize("form", [0, 1, 2, 4, 7])
@pytest.mark.parametrize("name,cell", [("Silverstone", 16.0), ("Silverstone", 7.0), ("Spa", 16.0), ("Austin", 33.0), ("Monza", 300.0)])
def test_grid_walk_equals_brute_force(oracle, gridcheck, name, cell, form):
    t = O.Track(name)
    ox, oy, ang = make_rays(t, 20000, 42)
    got = np.zeros(ox.size, dtype=np.float32)
    tests = np.zeros(ox.size, dtype=np.uint32)
    cells = np.zeros(ox.size, dtype=np.uint32)
    points = np.zeros(ox.size, dtype=np.uint32)
    info = np.zeros(8, dtype=np.int32)
    seg = np.ascontiguousarray(t.segments.reshape(-1))
    assert gridcheck.gridcheck_cast(seg, t.S, cell, ox, oy, ang, ox.size, got, tests, cells, info, form, points) == 0
    want = brute(t, ox, oy, ang)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    assert (want < 200).mean() > 0.4
    assert tests.mean() < 0.2 * t.S  # the walk really culls
    if form == 1:
        assert info[2] + info[5] <= info[6] <= info[2] + 2 * info[5] + 2 + info[2] // 6  # segments + runs (+ padding, chunk overlap)
        assert tests.mean() < 0.25 * points.mean() + 1  # the side rule skips most registered segments
    assert cells.max() <= info[0] + info[1] + 2


def test_degenerate_segment_sets(oracle, gridcheck):
    """One segment; zero-length and non-finite segments mixed in; a segment set far from the origin."""
    rng = np.random.default_rng(1)
    for segs in (np.array([[10, -5, 10, 5]], dtype=np.float32),
                 np.array([[10, -5, 10, 5], [3, 3, 3, 3], [np.nan, 0, 1, 1], [np.inf, 0, 5, 5], [20, -50, 20, 50]], dtype=np.float32),
                 (rng.uniform(0, 300, (200, 4)) + 1.0e5).astype(np.float32)):
        n = 4000
        lo, hi = np.nanmin(np.where(np.isfinite(segs), segs, np.nan)), np.nanmax(np.where(np.isfinite(segs), segs, np.nan))
        ox = rng.uniform(lo - 50, hi + 50, n).astype(np.float32)
        oy = rng.uniform(lo - 50, hi + 50, n).astype(np.float32)
        ang = rng.uniform(-np.pi, np.pi, n).astype(np.float32)
        got = np.zeros(n, dtype=np.float32)
        info = np.zeros(8, dtype=np.int32)
        flat = np.ascontiguousarray(segs.reshape(-1))
        tests, cells, points = (np.zeros(n, dtype=np.uint32) for _ in range(3))
        want = np.array([O.lib().oracle_cast_ray(float(ox[i]), float(oy[i]), float(ang[i]), flat, segs.shape[0]) for i in range(n)],
                        dtype=np.float32)
        for form in (0, 1, 3):
            assert gridcheck.gridcheck_cast(flat, segs.shape[0], 16.0, ox, oy, ang, n, got, tests, cells, info, form, points) == 0
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), form
