"""The C-ABI library builds, loads without a GPU, and exports every symbol include/okenv.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "okenv.h")).read()
    return sorted(set(re.findall(r"OKENV_API\s+[\w\s\*]+?\b(okenv_\w+)\s*\(", text)))


def test_header_and_binding_agree(ok):
    decl = declared_symbols()
    assert len(decl) >= 30
    assert sorted(ok.capi.SYMBOLS) == decl


def test_library_exports_every_declared_symbol(ok):
    lib = C.CDLL(ok.capi.lib_path())
    for s in declared_symbols():
        assert hasattr(lib, s), s


def test_fails_loudly_without_gpu(ok):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    t = ok.Track("Austin")
    with pytest.raises(ok.capi.OkenvError) as e:
        ok.BatchedEnvironment.from_track(t, 4, 8)
    assert e.value.code == -3 and "no CPU fallback" in str(e.value)
    with pytest.raises(ok.capi.OkenvError):
        ok.debug_sincos(np.zeros(4, dtype=np.float32))


def test_invalid_arguments_are_reported(ok):
    L = ok.capi.load()
    h = C.c_void_p()
    seg = np.zeros(4, dtype=np.float32)
    rays = np.zeros(1, dtype=np.float32)
    assert L.okenv_create(C.byref(h), ok.capi.ptr(seg), 0, 1, 1, ok.capi.ptr(rays), 0, 0, 0.0) == -1  # no segments
    assert L.okenv_create(C.byref(h), ok.capi.ptr(seg), 1, 0, 1, ok.capi.ptr(rays), 0, 0, 0.0) == -1  # no agents
    assert b"agent" in L.okenv_last_error(None)
    tr = C.c_void_p()
    assert L.okenv_track_load(C.byref(tr), b"/nonexistent/track.csv") == -4
    assert L.okenv_step(None, 1) == -1


def test_product_has_no_oracle_dependency():
    """Nothing under openkitchen_amd/ or include/ may import, include or link the oracle."""
    for base in ("openkitchen_amd", "include"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".h", ".hip", ".cpp", ".c")):
                    text = open(os.path.join(dirpath, f), errors="ignore").read()
                    assert "_oracle" not in text and "oracle/" not in text.replace("under oracle/", ""), os.path.join(dirpath, f)


def test_every_handle_entry_point_stops_a_resident_kernel_first():
    """While a resident step kernel serves okenv_step_packed nothing else may run against the handle's state: every C-ABI
    function that takes the handle begins with OK_QUIESCE(h).  The exceptions are listed: okenv_step_packed (it IS the
    resident path), okenv_destroy (stops it itself), okenv_get_info / okenv_last_error (host-side data only) and
    okenv_set_sensor_offset (quiesces only when the value changes: the facade calls it before every step)."""
    src = open(os.path.join(ROOT, "openkitchen_amd", "csrc", "okenv_capi.hip")).read()
    exempt = {"okenv_step_packed", "okenv_destroy", "okenv_get_info", "okenv_last_error", "okenv_set_sensor_offset"}
    seen = 0
    for m in re.finditer(r"^    (?:__attribute__\(\(visibility\(\"default\"\)\)\) )?(?:int|const char \*)\s*(okenv_\w+)\(okenv_t h\b[^{]*\{\n(.*?)\n", src, re.M | re.S):
        name, first_line = m.group(1), m.group(2)
        seen += 1
        if name in exempt:
            continue
        assert "OK_QUIESCE(h);" in first_line, name
    assert seen >= 45
    body = src[src.index("int okenv_set_sensor_offset(okenv_t h"):]
    assert "OK_QUIESCE(h);" in body[:600]
    body = src[src.index("int okenv_destroy(okenv_t h"):]
    assert "stopResident(h)" in body[:600]
