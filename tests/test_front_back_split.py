"""The front / back split of the segment set (openkitchen_amd/csrc/ok_grid.h: okClassifyFrontBack; ok_raycast.h: okOriginChiScalar,
the ambiguity report of ok_first_hit_update) on the CPU, through tests/cpp/grid_check.cpp: a ray whose origin is certified to lie
where chi = 1 and whose front walk rejected nothing within rounding takes its first hit from the front segments alone -- and that
must be, bit for bit, the first hit over ALL segments, the reference's sweep (Environment/CollisionChecker.cu:43-70) as the oracle
restates it.  Rays that are not certified walk the back image too and must give the same.  Origins: where the bench recipe's agents
are; on, beside and between the boundary polylines (the 3 px strip between the inner and the outer one is exactly where the outer
one comes first); on vertices; far outside.  Directions: random, along segments, aimed exactly at vertices."""
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
    so = os.path.join(out_dir, "libgridcheck_fb_san.so" if O.SANITIZE else "libgridcheck_fb.so")
    subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared"] + (O.SAN_FLAGS if O.SANITIZE else []) + ["-o", so, src],
                   check=True)
    G = C.CDLL(so)
    G.gridcheck_cast_fb.argtypes = [O.f32p, C.c_int, C.c_float, O.f32p, O.f32p, O.f32p, C.c_int, O.f32p, O.u32p, O.i32p, C.c_int, C.c_int, O.u32p]
    return G


def brute(t, ox, oy, ang):
    L = O.lib()
    seg = np.ascontiguousarray(t.segments.reshape(-1))
    return np.array([L.oracle_cast_ray(float(ox[i]), float(oy[i]), float(ang[i]), seg, t.S) for i in range(ox.size)], dtype=np.float32)


def same_t(a, b):
    """bit-equal first-hit parameters; t = +0 and t = -0 count as equal (origins exactly on a vertex: which of the two zeros the
    reference's sequential sweep keeps depends on the segments' order -- the one divergence DESIGN.md section 6 documents, present
    with and without the split)"""
    return (a.view(np.uint32) == b.view(np.uint32)) | ((a == 0) & (b == 0))


def degenerate_hit(t, ox, oy, ang, want):
    """True when the sweep's first hit `want` of this ray comes from a segment the ray is (numerically) collinear with: direction
    within 1e-4 rad of the segment's AND origin within 1e-2 px of its supporting line.  There the reference's quotients are
    ratios of two rounding residues -- t and s can come out as anything, e.g. t = 0 on a segment 23 px down the ray -- and no
    broad phase that stops at the first geometric hit can reproduce them (second documented divergence;
    found by tests/tools/fuzz_front_back.py, present in every form of the grid walk since round 1).  An agent has to sit ON a
    boundary line, looking along it, for this: it has crashed long before."""
    f = np.float32
    s = O.lib()
    sn, cs = np.zeros(1, f), np.zeros(1, f)
    s.oracle_sincosf(np.array([ang], f), sn, cs, 1)
    rdx, rdy = cs[0], sn[0]
    seg = t.segments.astype(f)
    sdx, sdy = seg[:, 2] - seg[:, 0], seg[:, 3] - seg[:, 1]
    denom = rdx * sdy - rdy * sdx
    ex, ey = seg[:, 0] - f(ox), seg[:, 1] - f(oy)
    with np.errstate(all="ignore"):
        tt = (ex * sdy - ey * sdx) / denom
        ss = (ex * rdy - ey * rdx) / denom
    valid = (np.abs(denom) >= f(1e-8)) & (tt >= 0) & (tt <= f(200)) & (ss >= 0) & (ss <= 1)
    winners = np.flatnonzero(valid & (tt == f(want)))
    ln = np.maximum(np.hypot(sdx.astype(np.float64), sdy.astype(np.float64)), 1e-12)
    sin_angle = np.abs(denom.astype(np.float64)) / ln
    line_dist = np.abs(ex.astype(np.float64) * sdy - ey.astype(np.float64) * sdx) / ln
    return winners.size > 0 and bool(((sin_angle[winners] < 1e-4) & (line_dist[winners] < 1e-2)).all())


def mismatches(t, ox, oy, ang, got, ref):
    """indices of rays whose first hit differs from the sweep's, the degenerate collinear ones left out (and counted)"""
    bad = np.flatnonzero(~same_t(got, ref))
    real = [int(i) for i in bad if not degenerate_hit(t, ox[i], oy[i], ang[i], ref[i])]
    return np.array(real, dtype=np.int64), bad.size - len(real)


def cast_fb(G, t, cell, ox, oy, ang, parts, force_back=0):
    n = ox.size
    out, fl, info, pts = np.zeros(n, np.float32), np.zeros(n, np.uint32), np.zeros(10, np.int32), np.zeros(n, np.uint32)
    rc = G.gridcheck_cast_fb(np.ascontiguousarray(t.segments.reshape(-1)), t.S, cell, np.ascontiguousarray(ox, dtype=np.float32),
                             np.ascontiguousarray(oy, dtype=np.float32), np.ascontiguousarray(ang, dtype=np.float32), n, out, fl, info, parts, force_back, pts)
    assert rc == 0, rc
    return out, fl, info, pts


def adversarial_rays(t, n, seed):
    """Origins and directions chosen to sit where the argument could break."""
    rng = np.random.default_rng(seed)
    seg = t.segments
    k = n // 8
    ox, oy, ang = np.zeros(n, np.float32), np.zeros(n, np.float32), rng.uniform(-np.pi, np.pi, n).astype(np.float32)
    sel = rng.integers(0, t.S, n)
    a = seg[sel, 0:2].astype(np.float64)
    b = seg[sel, 2:4].astype(np.float64)
    d = b - a
    ln = np.maximum(np.hypot(d[:, 0], d[:, 1]), 1e-9)
    nrm = np.stack([-d[:, 1], d[:, 0]], axis=1) / ln[:, None]
    u = rng.uniform(0, 1, n)[:, None]
    on = a + u * d  # a point on a random segment (inner or outer)
    off = rng.choice([-3.5, -3.0, -2.0, -1.5, -0.5, -1e-3, -1e-5, 0.0, 1e-5, 1e-3, 0.5, 1.5, 2.0, 3.0, 3.5, 6.0], n)[:, None]
    p = on + off * nrm  # beside it: inside the track, in the strip between inner and outer boundary, beyond the outer one
    ox[:], oy[:] = p[:, 0], p[:, 1]
    # exactly on vertices
    ox[:k], oy[:k] = seg[sel[:k], 0], seg[sel[:k], 1]
    # along the segment the origin sits beside (grazing), both ways
    ang[k:2 * k] = np.arctan2(d[k:2 * k, 1], d[k:2 * k, 0]).astype(np.float32)
    ang[2 * k:3 * k] = np.arctan2(-d[2 * k:3 * k, 1], -d[2 * k:3 * k, 0]).astype(np.float32)
    # aimed exactly at some vertex nearby (as exactly as fp32 angles allow): the slip-through case
    tgt = rng.integers(0, t.S, k)
    far = seg[tgt, 0:2].astype(np.float64)
    src = far + rng.normal(0, 25, (k, 2))
    ox[3 * k:4 * k], oy[3 * k:4 * k] = src[:, 0], src[:, 1]
    ang[3 * k:4 * k] = np.arctan2(far[:, 1] - src[:, 1], far[:, 0] - src[:, 0]).astype(np.float32)
    # on the centre line and near it (ordinary driving), and far outside the track
    idx = rng.integers(0, t.P, k)
    ox[4 * k:5 * k] = (t.x[idx] + rng.normal(0, 6, k)).astype(np.float32)
    oy[4 * k:5 * k] = (t.y[idx] + rng.normal(0, 6, k)).astype(np.float32)
    ox[5 * k:6 * k] = rng.uniform(-500, 2100, k).astype(np.float32)
    oy[5 * k:6 * k] = rng.uniform(-500, 1900, k).astype(np.float32)
    return ox, oy, ang


@pytest.mark.parametrize("track_name,cell,parts", [("Austin", 24.0, 1), ("Silverstone", 24.0, 8), ("Monza", 20.0, 4), ("Spa", 24.0, 8), ("Spa", 16.0, 1),
                                                    ("Silverstone", 40.0, 2)])
def test_split_gives_the_sweeps_first_hit(gridcheck, oracle, track_name, cell, parts):
    t = O.Track(track_name)
    ox, oy, ang = adversarial_rays(t, 24000, 11 + int(cell) + parts)
    ref = brute(t, ox, oy, ang)
    got, fl, info, _ = cast_fb(gridcheck, t, cell, ox, oy, ang, parts)
    assert info[0] == 1 and info[2] > 0.9 * t.S / 2  # the split exists, nearly all of the outer polylines are back segments
    bad, collinear = mismatches(t, ox, oy, ang, got, ref)
    assert bad.size == 0 and collinear <= 2, (bad[:5], got[bad[:5]], ref[bad[:5]], fl[bad[:5]], ox[bad[:5]], oy[bad[:5]], ang[bad[:5]])
    # the population is what it was made to be: certified and uncertified origins, ambiguous walks, all in numbers
    cert, amb, backw = (fl & 1) != 0, (fl & 2) != 0, (fl & 4) != 0
    assert 0.2 < cert.mean() < 0.9 and (amb & cert).sum() > 3 and (backw == (~cert | amb)).all()
    # and every ray through the back image as well: front + back is the whole set
    got2, _, _, _ = cast_fb(gridcheck, t, cell, ox, oy, ang, parts, force_back=1)
    assert mismatches(t, ox, oy, ang, got2, ref)[0].size == 0


def test_the_recipes_agents_are_certified_and_see_half_the_points(gridcheck, oracle):
    """Where it pays: the bench recipe's population after 200 steps (some agents in the strip or outside by then)."""
    t = O.Track("Silverstone")
    N, R = 256, 64
    fan = O.default_ray_fan(R)
    env = O.OracleEnv(t.segments, N, R, fan, (t.x, t.y, t.heading))
    env.init_bench_state(0, 0)
    env.rollout_random(200, 1234, 0, 0, threads=8)
    s = env.snapshot()
    keep = s["crashed"] == 0
    px, py, rot = s["pos_x"][keep], s["pos_y"][keep], s["rot"][keep]
    ox, oy = np.repeat(px, R).astype(np.float32), np.repeat(py, R).astype(np.float32)
    ang = (np.float32(0.01745329238474369) * (np.repeat(rot, R) + np.tile(fan, px.size))).astype(np.float32)
    ref = brute(t, ox, oy, ang)
    got, fl, info, pts = cast_fb(gridcheck, t, 24.0, ox, oy, ang, 8)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))
    assert (fl & 1).mean() > 0.97 and ((fl >> 2) & 1).mean() < 0.03 and ((fl >> 1) & 1).mean() < 0.002
    assert info[4] > 0.98 * info[9] and info[6] + info[7] < 160 * 1024
    assert pts.mean() < 32  # the combined image shows such rays ~48 points


def test_inputs_that_are_not_a_track_get_no_split(gridcheck):
    class T:  # a star of unconnected spokes: no closed chains
        pass
    rng = np.random.default_rng(1)
    t = T()
    t.S = 64
    t.segments = np.concatenate([rng.uniform(100, 900, (64, 2)), rng.uniform(100, 900, (64, 2))], axis=1).astype(np.float32)
    n = 16
    out, fl, info, pts = np.zeros(n, np.float32), np.zeros(n, np.uint32), np.zeros(10, np.int32), np.zeros(n, np.uint32)
    z = np.zeros(n, np.float32)
    rc = gridcheck.gridcheck_cast_fb(np.ascontiguousarray(t.segments.reshape(-1)), t.S, 24.0, z, z, z, n, out, fl, info, 1, 0, pts)
    assert rc == 1 and info[0] == 0
