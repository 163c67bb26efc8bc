#!/usr/bin/env python3
"""Is a soak_batch.py miss the batch's doing?  Re-runs ONE-VIEW backward passes of a failing draw several times and prints how
far two runs of the same kernel on the same inputs are apart (the blend backward's float atomics arrive in a different
order every run) next to the batch-vs-loop difference.   usage: python tests/tools/debug_soak_batch.py P W H V deg pre scene_seed"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import hip_helpers as hh  # noqa: E402
from util import make_scene  # noqa: E402
from dgr_amd import light as L  # noqa: E402

P, W, H, V, deg, pre, seed = (int(a) for a in sys.argv[1:8])
mode = sys.argv[8] if len(sys.argv) > 8 else "as drawn"
T = hh.T
ss = [make_scene(P, W, H, seed, view_index=v) for v in range(V)]
if mode == "translucent":
    ss = [s._replace(opac=(s.opac * 0.12).astype(np.float32)) for s in ss]
elif mode == "opaque":
    ss = [s._replace(opac=np.minimum(1.0, s.opac * 0.2 + 0.85).astype(np.float32)) for s in ss]
s = ss[0]
cov = None
if pre in (3, 5):
    c = torch.empty((P, 6), device=hh.dev())
    L._capi.load().dgr_cov3d_forward(L._capi.stream_handle(), P, T(s.scales).data_ptr(), T(s.rots).data_ptr(), 1.0, c.data_ptr())
    cov = c.cpu().numpy()
for v, x in enumerate(ss):
    one, d1 = hh.hip_forward(x, deg, cov3D_precomp=cov)
    gr = tuple(g * (W * H) ** 0.5 for g in (x.gC, x.gD, x.gM, x.gV))
    runs = [hh.hip_backward(x, deg, one, grads=gr, cov3D_precomp=cov) for _ in range(4)]
    for k in ("dL_dmeans3D", "dL_dcov3D", "dL_dscales", "dL_dopacity", "dL_dmeans2D"):
        scale = np.abs(runs[0][k]).max()
        if scale == 0:
            continue
        d = max(np.abs(runs[0][k].astype(np.float64) - r[k]).max() for r in runs[1:]) / scale
        row = int(np.abs(runs[0][k].astype(np.float64) - runs[1][k]).reshape(P, -1).max(1).argmax())
        print(f"view {v} {k}: two runs of the one-view backward differ by {d:.2e} of scale (row {row}, value scale {scale:.3e})")
