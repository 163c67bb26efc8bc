"""Fused sparse Adam for the Gaussian parameters.

`torch.optim.Adam` semantics (no weight decay, no amsgrad) with one HIP launch per tensor through the C ABI
(`dgr_sparse_adam`, csrc/optim.hip).  `step(visible=radii)` updates only the Gaussians some view saw -- parameter and
both moments of the other rows are left untouched, as in 3DGS's sparse Adam; `step()` updates every row.
"""
import torch

from . import _capi


class SparseAdam:
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        """`params`: tensors [P, ...] or torch-style groups `{"params": [...], "lr": ...}`."""
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
                rc = lib.dgr_sparse_adam(st, rows, k, p.data_ptr(), grad.data_ptr(), s[0].data_ptr(), s[1].data_ptr(),
                                         None if visible is None else visible.data_ptr(), float(g["lr"]), float(b1),
                                         float(b2), float(g["eps"]), self.steps)
                if rc:
                    raise RuntimeError(_capi.last_error())
