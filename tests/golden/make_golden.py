"""Generates the committed golden fixtures.  Run in the development container (needs /root/reference for the
reference-derived part):   python tests/golden/make_golden.py

  ref_tracks.npz      -- FROM THE REFERENCE's compiled RaceTrack.cpp (oracle/_ref): per config track P, the fp32 bit
                         hashes (sha256) of the centre line, widths, headings and the four boundary polylines, the
                         first/last 8 centre-line points, headings and boundary points, and the reference's
                         findNearestTrackIndexBruteForce / getNearestDistanceToTrackBoundary / getDistanceToLaneCenter
                         answers for 256 probe points.
  ref_kinematics.npz  -- FROM THE REFERENCE's compiled Agent.cpp: Agent::move trajectories (VELOCITY and
                         ACCELERATION) under seeded actions, 400 steps each; plus the default sensor fan and the
                         Agent::reset probe.
  c1_trajectory.npz   -- from the ORACLE (parity definition, ok_sincosf trig): BASELINE config 1 shape (64 agents x
                         16 rays, Austin), bench driver loop, 450 steps: full state every 50 steps and at the end.
                         A regression pin for oracle and GPU path alike (the reference cannot produce it: its
                         raycast exists only as CUDA).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import _oracle as O  # noqa: E402

TRACKS = ["Austin", "Silverstone", "Monza", "Spa"]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    O.build_oracle(with_ref=True)
    assert O.have_ref(), "reference shim not built"
    out = {}
    rng = np.random.default_rng(20260220)
    for name in TRACKS:
        t = O.Track(name, "ref")
        out[name + "_P"] = np.int32(t.P)
        for k in O.Track.KEYS:
            a = getattr(t, k)
            out["%s_sha_%s" % (name, k)] = np.array(sha(a))
            out["%s_head_%s" % (name, k)] = a[:16].copy()
            out["%s_tail_%s" % (name, k)] = a[-16:].copy()
        qx = rng.uniform(0, 1600, 256).astype(np.float32)
        qy = rng.uniform(0, 1400, 256).astype(np.float32)
        qx[:32], qy[:32] = t.x[::max(1, t.P // 32)][:32], t.y[::max(1, t.P // 32)][:32]
        idx = np.zeros(256, dtype=np.int32)
        h = O.ref().ref_track_load(O.track_path(name).encode())
        O.ref().ref_nearest_track_idx(h, qx, qy, 256, idx)
        O.ref().ref_track_free(h)
        bd, lc = np.zeros(256, dtype=np.float32), np.zeros(256, dtype=np.float32)
        h = O.ref().ref_track_load(O.track_path(name).encode())
        O.ref().ref_track_queries(h, qx, qy, 256, bd, lc)
        O.ref().ref_track_free(h)
        out[name + "_probe_x"], out[name + "_probe_y"], out[name + "_probe_idx"] = qx, qy, idx
        out[name + "_probe_boundary"], out[name + "_probe_lane"] = bd, lc
    np.savez_compressed(os.path.join(HERE, "ref_tracks.npz"), **out)

    kin = {}
    for mode in (0, 1):
        n = 400
        thr = (rng.uniform(0, 100, n) if mode == 0 else rng.uniform(-0.4, 0.6, n)).astype(np.float32)
        steer = rng.uniform(-5, 6, n).astype(np.float32)
        res = [np.zeros(n, dtype=np.float32) for _ in range(5)]
        O.ref().ref_agent_rollout(mode, 700.0, 500.0, 33.0, thr, steer, n, *res)
        kin["m%d_thr" % mode], kin["m%d_steer" % mode] = thr, steer
        for nm, a in zip(("x", "y", "rot", "speed", "acc"), res):
            kin["m%d_%s" % (mode, nm)] = a
    rays = np.zeros(64, dtype=np.float32)
    n = O.ref().ref_agent_default_rays(rays, 64)
    kin["default_rays"] = rays[:n].copy()
    probe = np.zeros(9, dtype=np.float32)
    O.ref().ref_agent_reset_probe(11.0, 22.0, 33.0, probe)
    kin["reset_probe"] = probe
    np.savez_compressed(os.path.join(HERE, "ref_kinematics.npz"), **kin)

    t = O.Track("Austin", "oracle")
    fan = O.default_ray_fan(16)
    env = O.OracleEnv(t.segments, 64, 16, fan, (t.x, t.y, t.heading))
    env.init_bench_state(0, 0)
    traj = {"seed": np.uint32(1234), "N": np.int32(64), "R": np.int32(16)}
    for chunk in range(9):
        env.rollout_random(50, 1234, 0, chunk * 50)
        s = env.snapshot()
        for k in ("pos_x", "pos_y", "rot", "speed", "crashed", "timed_out", "disp_ctr", "dist", "rel_x", "rel_y"):
            traj["s%03d_%s" % ((chunk + 1) * 50, k)] = s[k]
    np.savez_compressed(os.path.join(HERE, "c1_trajectory.npz"), **traj)
    for f in ("ref_tracks.npz", "ref_kinematics.npz", "c1_trajectory.npz"):
        print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")


if __name__ == "__main__":
    main()
