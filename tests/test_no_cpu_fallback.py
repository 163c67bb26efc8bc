"""The product path has no CPU fallback: with CPU tensors (or without a GPU) every surface raises instead of computing
something else -- the one-view mirrors of the reference modules, the batched surface, markVisible.  (The reference has no CPU
path either: its buffers are hard-wired to torch::kCUDA, L/rasterize_points.cu:78-82.)  Nothing here imports oracle/."""
import sys

import pytest
import torch

from dgr_amd import batch as B
from dgr_amd import light as L


def _inputs(P=8, V=2, H=16, W=16):
    g = torch.Generator().manual_seed(0)
    r = lambda *s: torch.rand(*s, generator=g)  # noqa: E731
    return dict(means3D=r(P, 3), means2D=torch.zeros(P, 3), sh=r(P, 16, 3), opac=r(P, 1), scales=r(P, 3), rots=r(P, 4),
                views=torch.eye(4).repeat(V, 1, 1), projs=torch.eye(4).repeat(V, 1, 1), campos=torch.zeros(V, 3),
                gts=r(V, H, W), bg=torch.zeros(3), persp=torch.eye(4))


def test_one_view_surface_refuses_cpu_tensors():
    x = _inputs()
    rs = L.GaussianRasterizationSettings(16, 16, 0.6, 0.6, x["bg"], 1.0, x["views"][0], x["projs"][0], 3, x["campos"][0], False,
                                         False, x["persp"], False, False)
    with pytest.raises(RuntimeError, match="GPU only|no CPU|cuda|CUDA|HIP|hip"):
        L.GaussianRasterizer(rs)(x["means3D"], x["means2D"], x["opac"], shs=x["sh"], scales=x["scales"], rotations=x["rots"],
                                 viewmatrix=x["views"][0], gt_depth=x["gts"][0])


def test_batched_surface_refuses_cpu_tensors_and_bad_view_counts():
    x = _inputs()
    rs = B.BatchRasterizationSettings(16, 16, 0.6, 0.6, x["bg"], 1.0, x["views"], x["projs"], 3, x["campos"], False, False,
                                      x["persp"], False, False)
    with pytest.raises(RuntimeError, match="GPU only|no CPU|cuda|CUDA|HIP|hip"):
        B.GaussianRasterizerBatch(rs)(x["means3D"], torch.zeros(2, 8, 3), x["opac"], shs=x["sh"], scales=x["scales"],
                                      rotations=x["rots"], viewmatrices=x["views"], gt_depths=x["gts"])
    with pytest.raises(Exception, match="excatly one of"):  # (the reference's message, typo included)
        B.GaussianRasterizerBatch(rs)(x["means3D"], None, x["opac"], scales=x["scales"], rotations=x["rots"])
    with pytest.raises(Exception, match="exactly one of"):
        B.GaussianRasterizerBatch(rs)(x["means3D"], None, x["opac"], shs=x["sh"], scales=x["scales"])


def test_the_product_modules_do_not_import_the_oracle():
    assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules if "dgr_amd" in getattr(sys.modules[m], "__name__", "") )
    import dgr_amd.batch, dgr_amd.light, dgr_amd.full, dgr_amd.slam, dgr_amd.multiview, dgr_amd.optim  # noqa: E401,F401
    src = "".join(open(m.__file__).read() for m in (dgr_amd.batch, dgr_amd.light, dgr_amd.full, dgr_amd.slam, dgr_amd.multiview,
                                                    dgr_amd.optim))
    assert "import oracle" not in src and "from oracle" not in src
