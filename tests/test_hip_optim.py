"""dgr_amd.optim.SparseAdam (csrc/optim.hip) against torch.optim.Adam."""
import numpy as np
import pytest
import torch

from dgr_amd.optim import SparseAdam

pytestmark = pytest.mark.gpu


def make(P, dev, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    shapes = [(P, 3), (P, 16, 3), (P, 1), (P, 4)]
    return [torch.randn(s, generator=g).to(dev) for s in shapes]


def test_dense_steps_match_torch_adam():
    dev = torch.device("cuda:0")
    P = 3000
    a = [t.clone().requires_grad_() for t in make(P, dev, 0)]
    b = [t.clone().requires_grad_() for t in make(P, dev, 0)]
    lrs = [1.6e-4, 2.5e-3, 5e-2, 1e-3]
    ours = SparseAdam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], eps=1e-15)
    ref = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], eps=1e-15)
    for it in range(5):
        grads = make(P, dev, 100 + it)
        for p, q, g in zip(a, b, grads):
            p.grad, q.grad = g.clone(), g.clone()
        ours.step()
        ref.step()
        for p, q in zip(a, b):
            np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)


def test_invisible_rows_are_left_alone():
    dev = torch.device("cuda:0")
    P = 2000
    params = [t.clone().requires_grad_() for t in make(P, dev, 1)]
    before = [p.detach().clone() for p in params]
    opt = SparseAdam(params, lr=1e-2)
    radii = (torch.arange(P, device=dev) % 3 != 0).to(torch.int32) * 7  # every third Gaussian unseen
    for it in range(3):
        for p, g in zip(params, make(P, dev, 50 + it)):
            p.grad = g
        opt.step(visible=radii)
    unseen = (radii == 0).cpu().numpy()
    for p, b0 in zip(params, before):
        now, was = p.detach().cpu().numpy(), b0.cpu().numpy()
        assert np.array_equal(now[unseen], was[unseen])
        assert not np.allclose(now[~unseen], was[~unseen])
        m, v = opt.state[p]
        assert not m.cpu().numpy()[unseen].any() and not v.cpu().numpy()[unseen].any()
    # the visible rows follow plain Adam on those rows
    ref_p = [b0[~torch.from_numpy(unseen).to(dev)].clone().requires_grad_() for b0 in before]
    ref = torch.optim.Adam(ref_p, lr=1e-2)
    for it in range(3):
        for q, g in zip(ref_p, make(P, dev, 50 + it)):
            q.grad = g[~torch.from_numpy(unseen).to(dev)]
        ref.step()
    for p, q in zip(params, ref_p):
        np.testing.assert_allclose(p.detach().cpu().numpy()[~unseen], q.detach().cpu().numpy(), rtol=2e-6, atol=1e-7)
