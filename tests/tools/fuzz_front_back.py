"""Development aid: the front / back split against the oracle's brute-force sweep on the CPU, randomised at length
(tests/test_front_back_split.py's comparison; tracks x cell sizes x decompositions x adversarial ray sets).
usage: python tests/tools/fuzz_front_back.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402
import test_front_back_split as T  # noqa: E402


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else int(time.time())
    rng = np.random.default_rng(seed)
    O.build_oracle(with_ref=False)
    G = T.gridcheck.__wrapped__() if hasattr(T.gridcheck, "__wrapped__") else None
    if G is None:  # build the harness the way the fixture does
        import ctypes as C
        import subprocess
        so = os.path.join(ROOT, "tests", "cpp", "_build", "libgridcheck_fb.so")
        os.makedirs(os.path.dirname(so), exist_ok=True)
        subprocess.run(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-fPIC", "-shared", "-o", so, os.path.join(ROOT, "tests", "cpp", "grid_check.cpp")], check=True)
        G = C.CDLL(so)
        G.gridcheck_cast_fb.argtypes = [O.f32p, C.c_int, C.c_float, O.f32p, O.f32p, O.f32p, C.c_int, O.f32p, O.u32p, O.i32p, C.c_int, C.c_int, O.u32p]
    t0, cases, rays, amb, cert, collinear = time.time(), 0, 0, 0, 0, 0
    tracks = {n: O.Track(n) for n in ("Austin", "Silverstone", "Monza", "Spa")}
    while time.time() - t0 < budget:
        name = rng.choice(list(tracks))
        t = tracks[name]
        cell = float(rng.choice([12.0, 16.0, 20.0, 24.0, 28.0, 40.0, 64.0]))
        parts = int(rng.choice([1, 2, 3, 4, 6, 8]))
        n = 12000
        ox, oy, ang = T.adversarial_rays(t, n, int(rng.integers(0, 2 ** 31)))
        ref = T.brute(t, ox, oy, ang)
        got, fl, info, _ = T.cast_fb(G, t, cell, ox, oy, ang, parts)
        bad, col = T.mismatches(t, ox, oy, ang, got, ref)
        collinear += col
        if bad.size:
            print("MISMATCH seed %d %s cell %g parts %d: %d rays, first: o=(%r, %r) ang=%r got %r want %r flags %d" %
                  (seed, name, cell, parts, bad.size, float(ox[bad[0]]), float(oy[bad[0]]), float(ang[bad[0]]), float(got[bad[0]]), float(ref[bad[0]]), int(fl[bad[0]])), flush=True)
            sys.exit(1)
        cases += 1
        rays += n
        amb += int(((fl >> 1) & 1).sum())
        cert += int((fl & 1).sum())
    print("fuzz_front_back: seed %d, %d cases, %d rays (certified %.3f, ambiguous front walks %d), no differing bit in %.0f s "
          "(%d rays collinear with a boundary segment within rounding left out: DESIGN.md section 6)" %
          (seed, cases, rays, cert / max(rays, 1), amb, time.time() - t0, collinear), flush=True)


if __name__ == "__main__":
    main()
