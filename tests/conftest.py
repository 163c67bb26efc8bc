import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "diff-gaussian-rasterization_amd")
for p in (ROOT, PKG, os.path.join(PKG, "light"), os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    """The CPU oracle (float-math build).  Only tests may import it."""
    from oracle import oracle as O
    O.build()
    return O


# Two environment switches select paths on which some tests have nothing to test (profiles/r9/matrix.sh runs the suite under each switch):
#   DGR_FORWARD_MODE=callback -- every forward goes through the reference's resize-callback entry points, which block the host: no lazy
#       status mode, no hipGraph capture, nothing left queued when the call returns;
#   DGR_BINDING=ctypes        -- the compiled extension is not loaded: the tests OF the compiled binding have no subject.
_NEEDS_PRESIZED = ("test_lazy_status_mode_matches_strict", "test_captured_step", "test_strict_mode_refuses_a_capturing_stream",
                   "test_inputs_made_on_the_callers_stream", "test_hip_lazy_safety.py", "test_tracking_iteration_replayed_from_a_hipgraph",
                   "test_a_replayed_graph_follows_the_frames_lane_lists")
_NEEDS_COMPILED = ("test_hip_binding_guard.py", "test_compiled_and_ctypes_bindings_agree",
                   "test_callback_entry_points_match_the_presized_path[compiled]")


def pytest_collection_modifyitems(config, items):
    callback = os.environ.get("DGR_FORWARD_MODE") == "callback"
    ctypes_only = os.environ.get("DGR_BINDING") == "ctypes"
    for item in items:
        if callback and any(k in item.nodeid for k in _NEEDS_PRESIZED):
            item.add_marker(pytest.mark.skip(reason="DGR_FORWARD_MODE=callback: the blocking entry points have no lazy mode and cannot be captured"))
        if ctypes_only and any(k in item.nodeid for k in _NEEDS_COMPILED):
            item.add_marker(pytest.mark.skip(reason="DGR_BINDING=ctypes: the compiled binding is not loaded"))
