"""The C-ABI library loads (no GPU needed) and exports every symbol include/dgr_hip.h declares."""
import ctypes
import os
import re

import pytest

from dgr_amd import _capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dgr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dgr_[a-z0-9_]+)\s*\(", text)))


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
    # 64 (rec: 48 bytes of data in a 64-byte slot, one L2 line per gather) + 4 (depth) + 4 (radius) + 8 (rect) + 1 (clamped) +
    # 4 (goff) + 48 (SH direction derivatives) B per Gaussian, plus the per-256-Gaussian block totals and 256-byte alignment of each array
    assert 133 * 1000 <= g1 <= 133 * 1000 + 10 * 256 and g2 > g1
    # 24 B per instance (list, key scratch, ranks / pair columns, pair keys) + the segment binning's tables: per tile row
    # of a 64x64 frame (4 tiles: one 16-tile segment) one word per bin_segments workgroup (256) for the run starts (+ one
    # closing row) and one for the running instance counts
    b = lib.dgr_binning_bytes(1000, 64, 64)
    assert 24 * 1000 + 256 * (5 + 4) * 4 <= b <= 24 * 1000 + 256 * (5 + 4) * 4 + 8 * 256
    assert lib.dgr_binning_bytes(2000, 64, 64) > b and lib.dgr_binning_bytes(1000, 1920, 1080) - b >= 256 * (68 * 8 * 2 - 9) * 4
    assert lib.dgr_light_backward_scratch_bytes(1000, 64, 64) >= 64 * 1000


def test_options_round_trip_and_reject_unknown_names():
    lib = _capi.load()
    assert lib.dgr_get_option(b"tight_cull") == 0
    assert lib.dgr_set_option(b"tight_cull", 1) == 0 and lib.dgr_get_option(b"tight_cull") == 1
    assert lib.dgr_set_option(b"tight_cull", 0) == 0
    assert lib.dgr_set_option(b"profile_every", 8) == 0 and lib.dgr_get_option(b"profile_every") == 8
    assert lib.dgr_set_option(b"profile_every", 1) == 0
    assert lib.dgr_get_option(b"fast_alpha") == 0  # the default alpha path carries the host's bits
    assert lib.dgr_set_option(b"fast_alpha", 1) == 0 and lib.dgr_get_option(b"fast_alpha") == 1
    assert lib.dgr_set_option(b"fast_alpha", 0) == 0
    assert lib.dgr_get_option(b"lds_count") in (0, 1, 2)
    keep = lib.dgr_get_option(b"lds_count")
    assert lib.dgr_set_option(b"lds_count", 2) == 0 and lib.dgr_get_option(b"lds_count") == 2
    assert lib.dgr_set_option(b"lds_count", keep) == 0
    keep = lib.dgr_get_option(b"lane_lists")   # 2 (the frame picks the blend kernels' lane lists) unless DGR_FWD_HALVES forced one
    assert keep in (0, 1, 2)
    for v, want in ((0, 0), (1, 1), (2, 2), (7, 2), (-3, 0)):
        assert lib.dgr_set_option(b"lane_lists", v) == 0 and lib.dgr_get_option(b"lane_lists") == want
    assert lib.dgr_set_option(b"lane_lists", keep) == 0
    assert lib.dgr_set_option(b"no_such_option", 1) == _capi.DGR_ERR_BAD_ARGUMENT
    assert b"no_such_option" in lib.dgr_last_error()
    with pytest.raises(ValueError):
        _capi.set_option("no_such_option", 1)
    assert lib.dgr_profile_select(b"no_such_stage") == _capi.DGR_ERR_BAD_ARGUMENT


def test_early_status_without_a_forward_reports_nothing_posted():
    import ctypes
    lib = _capi.load()
    buf = (ctypes.c_int * 4)(7, 7, 7, 7)
    assert lib.dgr_early_status_arm() == 0
    assert lib.dgr_early_status_wait(buf) == 1 and list(buf) == [0, 0, 0, 0]


def test_batch_entry_points_reject_bad_view_counts_before_touching_the_gpu():
    lib = _capi.load()
    views = (_capi.LightView * 1)()
    args = (0, 3, 16, None, 64, 48, None, None, None, None, None, 1.0, None, None, 0.6, 0.45, 0)
    assert lib.dgr_light_forward_batch(None, 0, views, *args) == _capi.DGR_ERR_BAD_ARGUMENT
    assert lib.dgr_light_forward_batch(None, _capi.MAX_BATCH_VIEWS + 1, views, *args) == _capi.DGR_ERR_BAD_ARGUMENT
    assert b"views per batch" in lib.dgr_last_error()
    grads = (_capi.LightViewGrad * 1)()
    assert lib.dgr_light_backward_batch(None, 0, grads, 0, 3, 16, None, 64, 48, None, None, None, None, 1.0, None, None, 0.6,
                                        0.45, None, None, None, None, None, None, None, 0, 0) == _capi.DGR_ERR_BAD_ARGUMENT
    # the ctypes structs mirror the C layout: 17 / 19 eight-byte slots (an int is padded to pointer alignment; round 9 added
    # dgr_light_view_grad.num_rendered behind scratch_bytes)
    assert ctypes.sizeof(_capi.LightView) == 17 * 8 and ctypes.sizeof(_capi.LightViewGrad) == 19 * 8
    assert lib.dgr_get_option(b"batch_streams") == 2
    assert lib.dgr_set_option(b"batch_streams", 1) == 0 and lib.dgr_get_option(b"batch_streams") == 1
    assert lib.dgr_set_option(b"batch_streams", 2) == 0
    assert lib.dgr_get_option(b"batch_order") == 0
    assert lib.dgr_set_option(b"batch_order", 1) == 0 and lib.dgr_get_option(b"batch_order") == 1
    assert lib.dgr_set_option(b"batch_order", 0) == 0


def test_thread_options_override_the_process_wide_ones_per_thread():
    """include/dgr_hip.h: dgr_set_thread_option / dgr_thread_options_swap -- host-side state only, no GPU needed."""
    import threading
    lib = _capi.load()
    assert lib.dgr_get_option(b"alpha_mode") == 0 and lib.dgr_get_thread_option(b"alpha_mode") == 0
    seen = {}

    def other():
        seen["before"] = lib.dgr_get_thread_option(b"alpha_mode")
        with _capi.thread_options(alpha_mode=2, deterministic_grads=1):
            seen["inside"] = (lib.dgr_get_thread_option(b"alpha_mode"), lib.dgr_get_thread_option(b"deterministic_grads"))
        seen["after"] = lib.dgr_get_thread_option(b"alpha_mode")

    with _capi.thread_options(alpha_mode=1, tight_cull=1):
        assert lib.dgr_get_thread_option(b"alpha_mode") == 1 and lib.dgr_get_thread_option(b"tight_cull") == 1
        assert lib.dgr_get_thread_option(b"fast_alpha") == 1
        assert lib.dgr_get_option(b"alpha_mode") == 0 and lib.dgr_get_option(b"tight_cull") == 0   # process-wide: untouched
        t = threading.Thread(target=other)
        t.start()
        t.join()
        word = lib.dgr_thread_options_effective()
        assert (word & 15) - 1 == 1 and ((word >> 4) & 15) - 1 == 1 and ((word >> 8) & 15) - 1 == 0
        with _capi.thread_options(alpha_mode=0):                      # nests
            assert lib.dgr_get_thread_option(b"alpha_mode") == 0 and lib.dgr_get_thread_option(b"tight_cull") == 1
        assert lib.dgr_get_thread_option(b"alpha_mode") == 1
    assert seen == {"before": 0, "inside": (2, 1), "after": 0}         # the other thread never saw this thread's values
    assert lib.dgr_get_thread_option(b"alpha_mode") == 0 and lib.dgr_get_thread_option(b"tight_cull") == 0
    # a forward's snapshot installed around a backward on another thread, then removed
    prev = lib.dgr_thread_options_swap(word)
    assert lib.dgr_get_thread_option(b"alpha_mode") == 1 and lib.dgr_get_thread_option(b"tight_cull") == 1
    lib.dgr_thread_options_swap(prev)
    assert lib.dgr_get_thread_option(b"alpha_mode") == 0
    assert lib.dgr_set_thread_option(b"lds_count", 1) != 0            # not a per-call option
    with pytest.raises(ValueError):
        with _capi.thread_options(alpha_mode=7):
            pass
    assert lib.dgr_get_thread_option(b"alpha_mode") == 0
