"""Development aid: time library variants (tools/build_variant.sh) on the C2 workload and check each against the oracle.

usage (GPU box): python tests/tools/variant_bench.py v0 v1 "v1:OKENV_PHASE1_RANGE=32" "v1:cell=16" ...
Each variant runs in its own process (the library path is fixed at import).  bench.py stays the judged harness.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

CHILD = r'''
import os, sys, time
import numpy as np
sys.path.insert(0, %(root)r)
import openkitchen_amd.buildlib as bl
bl.LIB_PATH = %(lib)r
bl.needs_build = lambda: False
import openkitchen_amd as ok
sys.path.insert(0, os.path.join(%(root)r, "tests"))
import _oracle as O
cell = %(cell)r
N, R, track_name = %(N)d, %(R)d, %(track)r
t = ok.Track(track_name)
# parity: 192 agents, 60 steps of the bench recipe, every field bit for bit
fan = ok.default_ray_fan(R)
PN = %(PN)d
dev = ok.BatchedEnvironment(t.segments, PN, fan, device=0, centerline=(t.x, t.y, t.heading), grid_cell=cell)
orc = O.OracleEnv(t.segments, PN, R, fan, (t.x, t.y, t.heading))
dev.init_bench_state(0, 0); orc.init_bench_state(0, 0)
for c in range(6):
    dev.rollout_random(10, 1234, 0, c * 10)
orc.rollout_random(60, 1234, 0, 0, threads=32)
d, o = dev.snapshot(), orc.snapshot()
bad = [k for k in ("pos_x", "pos_y", "rot", "crashed", "timed_out", "hit_x", "hit_y", "rel_x", "rel_y", "dist")
       if not np.array_equal(np.ascontiguousarray(d[k]).view(np.uint8), np.ascontiguousarray(o[k]).view(np.uint8))]
dev.close()
env = ok.BatchedEnvironment.from_track(t, N, R, grid_cell=cell)
env.init_bench_state(0, 0)
env.rollout_random(200, 1234, 0, 0)
env.sync()
res = []
for spl, steps in ((100, 1000), (20, 200), (1, 200)):
    env.set_timing(True)
    for c in range(steps // spl):
        env.rollout_random(spl, 1234, 0, 200 + c * spl)
    env.sync()
    ms, n = env.get_timing()
    env.set_timing(False)
    res.append("spl%%d %%.2f us/step" %% (spl, ms * 1e3 / steps))
print("%(tag)-44s parity %%s | %%s" %% ("OK" if not bad else "MISMATCH " + ",".join(bad), " | ".join(res)), flush=True)
'''


def main():
    O_built = False
    for spec in sys.argv[1:]:
        name, _, opts = spec.partition(":")
        env = dict(os.environ)
        cell, N, R, track, PN = 0.0, 4096, 64, "Silverstone", 192
        for kv in filter(None, opts.split(",")):
            k, v = kv.split("=")
            if k == "cell":
                cell = float(v)
            elif k == "N":
                N = int(v)
            elif k == "R":
                R = int(v)
            elif k == "track":
                track = v
            elif k == "PN":
                PN = int(v)
            else:
                env[k] = v
        lib = os.path.join(ROOT, "tools", "_build", "libokenv_%s.so" % name)
        if not O_built:
            subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
            O_built = True
        code = CHILD % dict(root=ROOT, lib=lib, cell=cell, N=N, R=R, track=track, tag=spec, PN=PN)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        out = r.stdout.strip().splitlines()
        print(out[-1] if out else "%s: FAILED rc=%d %s" % (spec, r.returncode, r.stderr[-400:]), flush=True)


if __name__ == "__main__":
    main()
