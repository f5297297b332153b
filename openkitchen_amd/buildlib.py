"""Builds libokenv.so (HIP kernels + C ABI + C++ Environment facade) in-tree for gfx950.

hipcc cross-compiles without a GPU, so this runs in the CPU-only development container as well as on the
MI355X box.  The shared object is git-ignored but travels with the gpurun snapshot.
"""
import fcntl
import os
import shutil
import subprocess
import sys

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(PKG_DIR, "csrc")
LIB_PATH = os.path.join(PKG_DIR, "libokenv.so")

HIP_SOURCES = [os.path.join(CSRC, "okenv_capi.hip")]
CXX_SOURCES = [
    os.path.join(CSRC, "facade", "RaceTrack.cpp"),
    os.path.join(CSRC, "facade", "Agent.cpp"),
    os.path.join(CSRC, "facade", "Environment.cpp"),
    os.path.join(CSRC, "facade", "Visualizer.cpp"),
]
HEADERS = [
    os.path.join(CSRC, "ok_raycast.h"), os.path.join(CSRC, "ok_grid.h"), os.path.join(CSRC, "okenv_kernels.h"),
    os.path.join(ROOT, "include", "okenv.h"), os.path.join(ROOT, "include", "okenv_math.h"),
]

# -ffp-contract=off: crash/done flags must be bit-exact against the CPU oracle, and FMA contraction changes
# the cancellation-prone 2x2 determinants of the ray-segment test ("FMA contraction").
FLAGS = ["-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "--offload-arch=gfx950",
         "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-Wno-unused-function"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP extension cannot be built")


def sources():
    return HIP_SOURCES + [s for s in CXX_SOURCES if os.path.exists(s)]


def needs_build():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + HEADERS + [os.path.join(ROOT, "include", "Environment", f)
                                  for f in os.listdir(os.path.join(ROOT, "include", "Environment")) if f.endswith(".h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compiles libokenv.so if it is missing or older than its sources.  Safe to call from several processes at once
    (one rank per GPU): an exclusive file lock serialises them, the check is repeated under the lock, and the new
    library is moved into place atomically, so no rank ever loads a half-written file."""
    if not force and not needs_build():
        return LIB_PATH
    with open(os.path.join(PKG_DIR, ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if force or needs_build():
                tmp = "%s.tmp.%d" % (LIB_PATH, os.getpid())
                cmd = [_hipcc()] + FLAGS + ["-I", os.path.join(ROOT, "include"), "-I", CSRC, "-o", tmp] + sources()
                if verbose:
                    print(" ".join(cmd), file=sys.stderr)
                try:
                    subprocess.run(cmd, check=True, cwd=ROOT)
                    os.replace(tmp, LIB_PATH)
                finally:
                    if os.path.exists(tmp):
                        os.remove(tmp)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


APPS_DIR = os.path.join(CSRC, "apps")
BIN_DIR = os.path.join(PKG_DIR, "bin")
APPS = ["genetic_learner_sim", "q_racer_sim"]
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
# genetic_learner_sim runs its islands' per-generation fitness all-gather over RCCL itself (ncclAllGather on the handles' streams)
APP_EXTRA = {"genetic_learner_sim": ["-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROCM, "include"), "-pthread", "-L", os.path.join(ROCM, "lib"),
                                     "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROCM, "lib")]}


def build_apps(verbose=False):
    """Compiles the two applications the north star names (openkitchen_amd/csrc/apps/: the EvolutionaryRacer and the
    Q_Learning episode loops over the C ABI) with g++ against include/okenv.h and links them to libokenv.so.  Returns the
    paths of the executables (openkitchen_amd/bin/)."""
    build(verbose=verbose)
    os.makedirs(BIN_DIR, exist_ok=True)
    out = []
    for app in APPS:
        src, exe = os.path.join(APPS_DIR, app + ".cpp"), os.path.join(BIN_DIR, app)
        deps = [src, LIB_PATH] + [os.path.join(APPS_DIR, f) for f in os.listdir(APPS_DIR) if f.endswith(".h")]
        if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(d) for d in deps):
            cmd = ["g++", "-std=c++17", "-O2", "-Wall", "-I", os.path.join(ROOT, "include"), "-o", exe, src, "-L", PKG_DIR, "-lokenv",
                   "-Wl,-rpath,$ORIGIN/.."] + APP_EXTRA.get(app, [])  # finds libokenv.so next to bin/, wherever the tree is
            if verbose:
                print(" ".join(cmd), file=sys.stderr)
            subprocess.run(cmd, check=True, cwd=ROOT)
        out.append(exe)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
    if "--apps" in sys.argv:
        print(build_apps(verbose=True))
