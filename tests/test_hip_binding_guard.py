"""The compiled binding's outputs are windows of shared allocations built without the dispatcher (csrc/torch_ext.cpp: view_of).
What must not change for a caller: autograd's saved-tensor check.  An output the backward needs (opacity_map: the light backward
derives T_final from it, L/diff_gaussian_rasterization/__init__.py:101-102) that is edited in place between forward and backward
must raise exactly as it does through the Python autograd.Function over `_C`; an output the backward does not need (color) may be
edited, and the edit is differentiated.  The same through the dispatcher-view fall-back (what another PyTorch than the validated
one gets), whose results must be the raw views' bit for bit."""
import numpy as np
import pytest
import torch

from util import make_scene
import hip_helpers as hh

pytestmark = pytest.mark.gpu


def leaves(s, dev):
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    L = dict(means3D=t(s.means), shs=t(s.shs), opac=t(s.opac), scales=t(s.scales), rots=t(s.rots), view=t(s.view))
    for v in L.values():
        v.requires_grad_(True)
    L["means2D"] = torch.zeros((s.P, 3), device=dev, requires_grad=True)
    return L


def forward(s, L, dev):
    from dgr_amd import light
    from dgr_amd.multiview import make_settings
    rast = light.GaussianRasterizer(make_settings(s, 3, dev))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    return rast(means3D=L["means3D"], means2D=L["means2D"], opacities=L["opac"], shs=L["shs"], scales=L["scales"], rotations=L["rots"],
                viewmatrix=L["view"], gt_depth=t(s.gt))


@pytest.fixture(params=["compiled node", "python Function", "compiled node, dispatcher views"])
def path(request, monkeypatch):
    from dgr_amd import light
    assert light._C is light._CompiledC, "the compiled extension was not built / not importable"
    ext = light._CompiledC.ext
    if request.param == "python Function":
        monkeypatch.setattr(light, "_USE_NODE", False)
    elif request.param.endswith("dispatcher views"):
        ext.set_raw_views(0)
    yield request.param
    ext.set_raw_views(-1)  # (decided again by the next view)


def test_an_in_place_edit_of_a_saved_output_raises(path):
    dev = hh.dev()
    s = make_scene(3000, 96, 64, 7)
    L = leaves(s, dev)
    color, radii, depth, median, var, opacity_map, unc, px = forward(s, L, dev)
    opacity_map.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        (color.sum() + depth.sum()).backward()


def test_an_in_place_edit_of_an_unsaved_output_is_differentiated(path):
    dev = hh.dev()
    s = make_scene(3000, 96, 64, 7)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    gC = t(s.gC) * (s.W * s.H) ** 0.5
    res = []
    for edit in (False, True):
        L = leaves(s, dev)
        color, radii, depth, median, var, opacity_map, unc, px = forward(s, L, dev)
        if edit:
            color.mul_(2.0)   # a gradient function of its own on top of the rasterizer's node
            depth.detach().add_(1.0)  # an edit autograd does not see, of an output the backward does not read
        (color * gC).sum().backward()
        torch.cuda.synchronize()
        res.append({k: v.grad.detach().cpu().numpy().astype(np.float64) for k, v in L.items() if v.grad is not None})
    for k in res[0]:
        a, b = res[0][k], res[1][k]
        scale = max(np.abs(a).max(), 1e-30)
        assert np.abs(2.0 * a - b).max() <= 2e-5 * scale, (path, k)  # (float atomics: two runs differ by their arrival order)


def test_raw_and_dispatcher_views_give_the_same_tensors():
    from dgr_amd import light
    ext = light._CompiledC.ext
    dev = hh.dev()
    s = make_scene(5000, 128, 96, 3)
    outs = []
    try:
        for mode in (1, 0):
            ext.set_raw_views(mode)
            L = leaves(s, dev)
            o = forward(s, L, dev)
            assert ext.raw_views() == mode
            outs.append([x.detach().cpu().numpy() for x in o])
            # outputs of one forward share allocations either way; shapes, dtypes and contiguity are the reference's
            assert o[0].shape == (3, s.H, s.W) and o[5].shape == (1, s.H, s.W) and o[1].dtype == torch.int32 and all(x.is_contiguous() for x in o)
    finally:
        ext.set_raw_views(-1)
    for i, (a, b) in enumerate(zip(*outs)):
        if i == 6:  # gau_uncertainty: a float-atomic sum per Gaussian, equal up to the arrival order
            assert np.allclose(a, b, rtol=1e-5, atol=1e-12)
        else:
            assert np.array_equal(a, b), i
    # decided by itself again: on the PyTorch this was validated on the raw views pass their self-check
    L = leaves(s, dev)
    forward(s, L, dev)
    assert ext.raw_views() == (1 if torch.__version__.startswith("2.10") else 0)
