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


def test_densification_stats_match_the_indexed_torch_ops():
    """3DGS's add_densification_stats + max_radii2D update, one launch, three views in a row."""
    from dgr_amd.optim import add_densification_stats
    dev = torch.device("cuda:0")
    P = 5001
    g = torch.Generator(device="cpu").manual_seed(3)
    accum, denom, maxr = (torch.zeros((P, 1), device=dev), torch.zeros((P, 1), device=dev), torch.zeros(P, device=dev))
    accum_ref, denom_ref, maxr_ref = accum.clone(), denom.clone(), maxr.clone()
    for view in range(3):
        dmeans2D = torch.randn((P, 3), generator=g).to(dev)
        radii = torch.randint(-1, 40, (P,), generator=g).to(torch.int32).to(dev)
        radii[torch.rand(P, generator=g).to(dev) < 0.3] = 0
        add_densification_stats(dmeans2D, radii, accum, denom, maxr)
        seen = radii > 0
        accum_ref[seen] += torch.norm(dmeans2D[seen, :2], dim=-1, keepdim=True)
        denom_ref[seen] += 1
        maxr_ref[seen] = torch.max(maxr_ref[seen], radii[seen].float())
    np.testing.assert_allclose(accum.cpu().numpy(), accum_ref.cpu().numpy(), rtol=1e-6, atol=1e-7)
    assert torch.equal(denom, denom_ref) and torch.equal(maxr, maxr_ref)
    # optional outputs
    before = accum.clone()
    add_densification_stats(dmeans2D, radii, None, denom, None)
    assert torch.equal(accum, before) and torch.equal(denom, denom_ref + (radii > 0).float().unsqueeze(1))


def test_capturable_steps_match_and_replay_from_a_graph():
    """capturable=True reads the step count from device memory: same updates as torch Adam, also when the step is recorded
    into a hipGraph once and replayed (the bias correction must follow the replays)."""
    dev = torch.device("cuda:0")
    P = 700
    a = [t.clone().requires_grad_() for t in make(P, dev, 7)]
    b = [t.clone().requires_grad_() for t in make(P, dev, 7)]
    ours = SparseAdam(a, lr=2e-3, capturable=True)
    ref = torch.optim.Adam(b, lr=2e-3)
    grads = make(P, dev, 8)
    for p, q, g in zip(a, b, grads):
        p.grad, q.grad = g.clone(), g.clone()
    for _ in range(3):  # eager steps (they also create the optimiser state before the capture)
        ours.step()
        ref.step()
    graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream(device=dev)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        with torch.cuda.graph(graph, stream=side):
            ours.step()
    torch.cuda.current_stream(dev).wait_stream(side)
    for _ in range(5):  # (recording a graph executes nothing)
        graph.replay()
        ref.step()
    torch.cuda.synchronize()
    for p, q in zip(a, b):
        np.testing.assert_allclose(p.detach().cpu().numpy(), q.detach().cpu().numpy(), rtol=5e-6, atol=2e-7)
