"""Development aid: lock-step trip counts of the cooperative raycast (tests/tools/wave_model.cpp) on bench-recipe poses.

usage: python tests/tools/wave_model.py [track] [agents] [rays]
Poses come from the CPU oracle's rollout of the bench recipe (test infrastructure used as a pose generator only).
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as O  # noqa: E402

SO = os.path.join(ROOT, "tools", "_build", "libwavemodel_san.so" if O.SANITIZE else "libwavemodel.so")


def build():
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-shared", "-fPIC"] + (O.SAN_FLAGS if O.SANITIZE else []) +
                   ["-I", os.path.join(ROOT, "include"),
                    "-I", os.path.join(ROOT, "openkitchen_amd", "csrc"), "-o", SO, os.path.join(ROOT, "tests", "tools", "wave_model.cpp")], check=True)
    L = C.CDLL(SO)
    f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
    L.wavemodel_run.argtypes = [f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, f32p, C.c_int, C.c_float, C.c_int, C.c_int,
                                np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")]
    return L


def poses(track, N, R, steps=60, seed=1234):
    O.build_oracle(with_ref=False)
    fan = O.default_ray_fan(R)
    env = O.OracleEnv(track.segments, N, R, fan, (track.x, track.y, track.heading))
    env.init_bench_state(0, 0)
    env.rollout_random(steps, seed, 0, 0, threads=8)
    s = env.snapshot()
    keep = s["crashed"] == 0
    return s["pos_x"][keep].copy(), s["pos_y"][keep].copy(), s["rot"][keep].copy(), fan


def main():
    tname = sys.argv[1] if len(sys.argv) > 1 else "Silverstone"
    N = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    R = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    L = build()
    track = O.Track(tname)
    px, py, rot, fan = poses(track, N, R)
    print("%s: %d live poses, %d rays" % (tname, px.size, R))
    if len(sys.argv) > 4 and sys.argv[4] == "tail":  # the tail kernel's arrangement: python wave_model.py Monza 256 32 tail
        f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
        L.wavemodel_tail.argtypes = [f32p, C.c_int, C.c_float, f32p, f32p, f32p, C.c_int, f32p, C.c_int, C.c_int, C.c_int, C.c_int]
        for cell in (20.0, 24.0, 16.0, 12.0):
            for split, strided in ((8, 0), (8, 1), (16, 0), (4, 0)):
                L.wavemodel_tail(track.segments, track.S, cell, px, py, rot, px.size, fan, R, split, strided, 1)
        return
    hdr = ("(pairs column = 8-slot rounds)\ncell  T1 split pb |  p1: cells chunks pairs exact (util: cell pair exact) |  p2: cells chunks pairs exact (util) | pend  p2frac | image")
    print(hdr)
    nums = [a for a in sys.argv[4:] if a.lstrip("-").isdigit()]
    mode = int(nums[0]) if len(nums) > 0 else 0
    goal = int(nums[1]) if len(nums) > 1 else 0
    p1cap = int(nums[2]) if len(nums) > 2 else 0
    L.wavemodel_set_p2_mode(mode, goal, p1cap)
    print("phase 1 capped at %d cells" % p1cap if p1cap else "phase 1 ends at its range")
    print("cells per round and ray (goal): %d" % goal)
    print("phase 2: %s" % {0: "equal parameter intervals, start cells owned by one walk (shipped)", 1: "cell tasks (lane j of a ray takes cells after phase 1; rounds until done)",
                           2: "equal parameter intervals, every walk processes every cell it touches (round 2)"}[mode])
    # further options: "front" = walk the front image of the front / back split; "nostop" = walks do not end on a hit (mock-up of
    # deferred exact tests: what the point loop then costs, against the passes the exact loop saves)
    front = 1 if "front" in sys.argv[4:] else 0
    nostop = 1 if "nostop" in sys.argv[4:] else 0
    L.wavemodel_set_options(front, nostop)
    predict = [a for a in sys.argv[4:] if a.startswith("predict=")]  # phase 2 cut at last step's hit distance + this margin [px]
    if predict:
        L.wavemodel_set_predict.argtypes = [C.c_float]
        L.wavemodel_set_predict(float(predict[0].split("=")[1]))
        print("phase 2 cut at the previous step's distance + %s px, second round for the rest" % predict[0].split("=")[1])
    print("image: %s; walks %s" % ("front segments only" if front else "all segments", "do not end on a hit" if nostop else "end on a hit inside the covered part"))
    for cell, t1, split, pb in ([(28, 48, 8, 1), (28, 32, 8, 1), (28, 64, 8, 1)] if front else
                                [(20, 48, 8, 1), (24, 48, 8, 1), (24, 40, 8, 1), (24, 32, 8, 1), (24, 24, 8, 1), (24, 56, 8, 1), (24, 200, 8, 1), (28, 48, 8, 1)]):
        out = np.zeros(20)
        rc = L.wavemodel_run(track.segments, track.S, float(cell), px, py, rot, px.size, fan, R, float(t1), split, pb, out)
        if rc != 0:
            print("cell %g: image not encodable" % cell)
            continue
        o = out
        print("%4g %3g %4d %3d | %6.2f %6.2f %6.2f %6.2f (%.2f %.2f %.2f) | %6.2f %6.2f %6.2f %6.2f (%.2f %.2f %.2f) | %5.1f %5.2f | %d B max %d"
              % (cell, t1, split, pb, o[0], o[1], o[2], o[3], o[4] / max(o[0], 1e-9) / 64, o[6] / max(o[2], 1e-9) / 64, o[7] / max(o[3], 1e-9) / 64,
                 o[8], o[9], o[10], o[11], o[12] / max(o[8], 1e-9) / 64, o[14] / max(o[10], 1e-9) / 64, o[15] / max(o[11], 1e-9) / 64,
                 o[16], o[17], int(o[18]), int(o[19])))


if __name__ == "__main__":
    main()
