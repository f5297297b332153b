"""The CMake face of the drop-in (include/Environment/CMakeLists.txt): put where the reference's Environment/ directory was, it
answers the applications' own three lines -- add_subdirectory(../Environment ...), ENVIRONMENT_INCLUDE_DIR = ../ and
target_link_libraries(... raylib dl rt Environment) (Template/CMakeLists.txt:10,11,34; Environment/CMakeLists.txt:29-88 defines
`CollisionChecker` and `Environment`) -- with the MI355X library.  CPU-only: configure, compile and link; running needs a GPU
(tests/test_facade.py, tests/test_reference_binding_dropin.py run such binaries)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


def cmake_build(src, build, *defs):
    gen = ["-G", "Ninja"] if shutil.which("ninja") else []
    r = subprocess.run(["cmake", "-S", src, "-B", build, "-DCMAKE_BUILD_TYPE=Release"] + gen + list(defs), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    r = subprocess.run(["cmake", "--build", build], capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    return r.stdout


def needed_libs(exe):
    out = subprocess.run(["readelf", "-d", exe], capture_output=True, text=True, check=True).stdout
    return [l.split("[")[1].rstrip("]") for l in out.splitlines() if "(NEEDED)" in l]


def test_template_shaped_consumer_builds_against_the_dropin(ok, tmp_path):
    """A project with the reference applications' CMake shape, in a tree whose Environment/ is a symlink to the drop-in."""
    tree = tmp_path / "OpenKitchen"
    (tree / "Template").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "cmake_consumer", "Paths.cmake"), tree / "Paths.cmake")
    shutil.copy(os.path.join(ROOT, "tests", "cmake_consumer", "CMakeLists.txt"), tree / "Template" / "CMakeLists.txt")
    shutil.copy(os.path.join(ROOT, "tests", "cpp", "test_facade.cpp"), tree / "Template" / "main.cpp")
    os.symlink(os.path.join(ROOT, "include", "Environment"), tree / "Environment")
    build = tmp_path / "build"
    cmake_build(str(tree / "Template"), str(build))
    for exe in ("template", "template_cc"):
        path = str(build / exe)
        assert os.path.exists(path)
        libs = needed_libs(path)
        assert "libokenv.so" in libs and not any("raylib" in l for l in libs), libs  # by name (no SONAME: no absolute path baked in)
        rp = subprocess.run(["readelf", "-d", path], capture_output=True, text=True).stdout
        assert os.path.join(ROOT, "openkitchen_amd") in rp  # build-tree RUNPATH finds the library where it lives


def test_a_copied_directory_needs_okenv_root(ok, tmp_path):
    """Copied instead of linked, the directory cannot find the tree it belongs to: that is an error message, and -DOKENV_ROOT fixes it."""
    tree = tmp_path / "OpenKitchen"
    (tree / "Template").mkdir(parents=True)
    shutil.copy(os.path.join(ROOT, "tests", "cmake_consumer", "Paths.cmake"), tree / "Paths.cmake")
    shutil.copy(os.path.join(ROOT, "tests", "cmake_consumer", "CMakeLists.txt"), tree / "Template" / "CMakeLists.txt")
    shutil.copy(os.path.join(ROOT, "tests", "cpp", "test_facade.cpp"), tree / "Template" / "main.cpp")
    shutil.copytree(os.path.join(ROOT, "include", "Environment"), tree / "Environment")
    r = subprocess.run(["cmake", "-S", str(tree / "Template"), "-B", str(tmp_path / "b1")], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "OKENV_ROOT" in r.stderr
    r = subprocess.run(["cmake", "-S", str(tree / "Template"), "-B", str(tmp_path / "b2"), "-DOKENV_ROOT=" + ROOT], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]


@pytest.mark.skipif(not os.path.isfile(os.path.join(REF, "Template", "CMakeLists.txt")), reason="reference tree not mounted")
def test_the_reference_template_project_builds_unchanged(ok, tmp_path):
    """The reference's own Template/CMakeLists.txt, Template/main.cpp and Paths.cmake, byte for byte (copied into a scratch tree at
    test time, never into the repository), next to a symlink Environment -> include/Environment: find_package(Torch) is served
    by the PyTorch wheel's CMake package; nothing else on the machine resembles the author's paths."""
    import torch
    tree = tmp_path / "OpenKitchen"
    (tree / "Template").mkdir(parents=True)
    shutil.copy(os.path.join(REF, "Paths.cmake"), tree / "Paths.cmake")
    for f in ("CMakeLists.txt", "main.cpp"):
        shutil.copy(os.path.join(REF, "Template", f), tree / "Template" / f)
    os.symlink(os.path.join(ROOT, "include", "Environment"), tree / "Environment")
    build = tmp_path / "build"
    cmake_build(str(tree / "Template"), str(build), "-DCMAKE_PREFIX_PATH=" + torch.utils.cmake_prefix_path)
    libs = needed_libs(str(build / "template"))
    assert "libokenv.so" in libs and any(l.startswith("libtorch") for l in libs) and not any("raylib" in l for l in libs), libs
