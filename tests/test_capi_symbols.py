"""The C-ABI library loads (no GPU needed) and exports every symbol include/dgr_hip.h declares."""
import ctypes
import os
import re

from dgr_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dgr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgr_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(_capi.LIB_PATH)
    syms = declared_symbols()
    assert len(syms) >= 10
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/dgr_hip.h but not exported"


def test_binding_covers_the_header():
    assert set(_capi.exported_symbols()) == set(declared_symbols())
    lib = _capi.load()
    assert lib.dgr_version().startswith(b"dgr_hip")


def test_state_sizes_scale_as_documented():
    lib = _capi.load()
    assert lib.dgr_geometry_bytes(0) == 0
    g1, g2 = lib.dgr_geometry_bytes(1000), lib.dgr_geometry_bytes(2000)
    # 48 (rec) + 4 (depth) + 4 (radius) + 24 (cov3D) + 8 (rect) + 1 (clamped) + 4 (goff) B per Gaussian, plus the
    # per-256-Gaussian block totals and 256-byte alignment of each array
    assert 93 * 1000 <= g1 <= 93 * 1000 + 9 * 256 and g2 > g1
    assert lib.dgr_binning_bytes(1000, 64, 64) >= 12 * 1000
    assert lib.dgr_light_backward_scratch_bytes(1000, 64, 64) >= 64 * 1000
