"""Fused sparse Adam for the Gaussian parameters.

`torch.optim.Adam` semantics (no weight decay, no amsgrad) with one HIP launch per tensor through the C ABI
(`dgr_sparse_adam`, csrc/optim.hip).  `step(visible=radii)` updates only the Gaussians some view saw -- parameter and
both moments of the other rows are left untouched, as in 3DGS's sparse Adam; `step()` updates every row.
"""
import torch

from . import _capi


@torch.no_grad()
def add_densification_stats(dmeans2D, radii, xyz_gradient_accum=None, denom=None, max_radii2D=None):
    """3DGS's per-view densification bookkeeping in one launch (`dgr_densification_stats`): for rows with radii > 0,
    `xyz_gradient_accum += |dmeans2D[:, :2]|`, `denom += 1`, `max_radii2D = max(max_radii2D, radii)`.
    `dmeans2D`: the `.grad` of the `means2D` tensor handed to the rasterizer ([P, 3]); the three accumulators are float32
    tensors with P elements ([P] or [P, 1]), updated in place; pass None to skip one."""
    P = radii.numel()
    for t in (xyz_gradient_accum, denom, max_radii2D):
        if t is not None and (t.numel() != P or t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda):
            raise RuntimeError("add_densification_stats: accumulators must be contiguous float32 GPU tensors with P elements")
    if radii.dtype != torch.int32 or not radii.is_cuda or dmeans2D.shape != (P, 3) or dmeans2D.dtype != torch.float32:
        raise RuntimeError("add_densification_stats: radii must be int32 [P] and dmeans2D float32 [P, 3] on the GPU")
    ptr = lambda t: None if t is None else t.data_ptr()
    rc = _capi.load().dgr_densification_stats(_capi.stream_handle(), P, dmeans2D.contiguous().data_ptr(),
                                              radii.contiguous().data_ptr(), ptr(xyz_gradient_accum), ptr(denom),
                                              ptr(max_radii2D))
    if rc:
        raise RuntimeError(_capi.last_error())


class SparseAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, capturable=False):
        """`params`: tensors [P, ...] or torch-style groups `{"params": [...], "lr": ...}`.  `capturable=True` keeps the
        step count on the device (one extra tiny launch per step) so that `step()` can be recorded into a hipGraph."""
        groups = list(params)
        if groups and isinstance(groups[0], torch.Tensor):
            groups = [{"params": groups}]
        self.param_groups = []
        for g in groups:
            g = dict(g)
            g.setdefault("lr", lr)
            g.setdefault("betas", betas)
            g.setdefault("eps", eps)
            g["params"] = list(g["params"])
            self.param_groups.append(g)
        self.state = {}
        self.steps = 0
        self.capturable = bool(capturable)
        self._step_dev = None

    def zero_grad(self, set_to_none=True):
        for g in self.param_groups:
            for p in g["params"]:
                if set_to_none:
                    p.grad = None
                elif p.grad is not None:
                    p.grad.zero_()

    @torch.no_grad()
    def step(self, visible=None):
        """visible: int32 [P] (e.g. the forward's radii; a row is updated where > 0) or None for every row."""
        lib = _capi.load()
        self.steps += 1
        if self.capturable:
            if self._step_dev is None:
                dev = self.param_groups[0]["params"][0].device
                self._step_dev = torch.full((1,), self.steps - 1, dtype=torch.int32, device=dev)
            self._step_dev.add_(1)
        if visible is not None:
            if visible.dtype == torch.bool:
                visible = visible.to(torch.int32)
            visible = visible.contiguous()
        st = _capi.stream_handle()
        for g in self.param_groups:
            b1, b2 = g["betas"]
            for p in g["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise RuntimeError("SparseAdam: parameters must be contiguous float32 tensors on the GPU")
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                s = self.state.get(p)
                if s is None:
                    s = self.state[p] = (torch.zeros_like(p), torch.zeros_like(p))
                rows = p.shape[0]
                k = p.numel() // max(rows, 1)
                if visible is not None and visible.numel() != rows:
                    raise RuntimeError("SparseAdam: `visible` must have one entry per row")
                vis = None if visible is None else visible.data_ptr()
                if self.capturable:
                    rc = lib.dgr_sparse_adam_capturable(st, rows, k, p.data_ptr(), grad.data_ptr(), s[0].data_ptr(),
                                                        s[1].data_ptr(), vis, float(g["lr"]), float(b1), float(b2),
                                                        float(g["eps"]), self._step_dev.data_ptr())
                else:
                    rc = lib.dgr_sparse_adam(st, rows, k, p.data_ptr(), grad.data_ptr(), s[0].data_ptr(), s[1].data_ptr(),
                                             vis, float(g["lr"]), float(b1), float(b2), float(g["eps"]), self.steps)
                if rc:
                    raise RuntimeError(_capi.last_error())
