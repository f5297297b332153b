"""bench.py reports PMC-derived figures (roofline.traffic, valu_roofline) from a committed profile only while the step kernel's
sources are the ones the profile was collected on (CPU-only test of the check itself)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def make_tree(tmp_path):
    for rel in bench.KERNEL_SOURCES:
        dst = tmp_path / rel
        dst.parent.mkdir(parents=True, exist_ok=True)
        shutil.copy(os.path.join(ROOT, rel), dst)
    (tmp_path / "profiles").mkdir()
    return str(tmp_path)


def test_profile_is_used_only_for_the_sources_it_was_collected_on(tmp_path):
    root = make_tree(tmp_path)
    prof = {"agents": 4096, "rays": 64, "track": "Silverstone", "hbm_bytes_per_agent_step": 16.5,
            "kernel_source_sha256": bench.kernel_source_hash(root)}
    path = os.path.join(root, "profiles", "hbm_traffic.json")
    json.dump(prof, open(path, "w"))
    tj, note = bench.load_counter_profile(4096, 64, "Silverstone", root=root)
    assert tj is not None and note is None and tj["hbm_bytes_per_agent_step"] == 16.5
    # another workload: not this profile's business
    tj, note = bench.load_counter_profile(8192, 32, "Monza", root=root)
    assert tj is None and "another workload" in note
    # the kernel changes: the counters are stale
    with open(os.path.join(root, bench.KERNEL_SOURCES[1]), "a") as f:
        f.write("\n// edited\n")
    tj, note = bench.load_counter_profile(4096, 64, "Silverstone", root=root)
    assert tj is None and note.startswith("stale_profile")
    # a profile without a hash (round 2's) is stale by definition
    del prof["kernel_source_sha256"]
    json.dump(prof, open(path, "w"))
    tj, note = bench.load_counter_profile(4096, 64, "Silverstone", root=root)
    assert tj is None and note.startswith("stale_profile")
    os.remove(path)
    tj, note = bench.load_counter_profile(4096, 64, "Silverstone", root=root)
    assert tj is None and "no counter profile" in note


def test_committed_profile_matches_the_committed_kernel():
    """whoever changes the kernel either re-collects the counters or accepts `traffic: null` -- this test only states which of
    the two the tree is in, it never fails on a stale profile"""
    tj, note = bench.load_counter_profile(4096, 64, "Silverstone")
    assert (tj is None) == (note is not None)
    print("committed counter profile:", "current" if tj else note)
