"""cProfile of examples/tracking.py's eager fused iteration (where the host's time goes).  GPU box."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
os.environ.setdefault("DGR_SYNC_MODE", os.environ.get("DGR_SYNC_MODE", "strict"))
import numpy as np, torch
from dgr_amd import slam
from dgr_amd.optim import SparseAdam
from dgr_amd.synth import camera, make_scene
from test_slam_render import Model, rot_to_quat
dev = torch.device("cuda:0")
W, H, P = 640, 480, 100000
s = make_scene(P, W, H, 3)
pc = Model(s, dev)
tanfovx, tanfovy, Rm, t_true, *_ = camera(W, H, 0.05)
bg, gt_depth = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
kw = dict(fov=(tanfovx, tanfovy), HW=(H, W), gt_depth=gt_depth, track_off=False, map_off=True)
q_true = torch.tensor(rot_to_quat(Rm), dtype=torch.float32, device=dev)
t_true = torch.tensor(t_true, dtype=torch.float32, device=dev)
with torch.no_grad():
    obs = slam.render(None, pc, None, bg, viewmatrix=slam.pose_to_camera(q_true, t_true, tanfovx, tanfovy)[0], **kw)
obs_c, obs_d = obs["render"].detach(), obs["depth"].detach()
q = (q_true + torch.tensor([0.0, 0.004, -0.006, 0.003], device=dev)).requires_grad_()
t = (t_true + torch.tensor([0.012, -0.009, 0.015], device=dev)).requires_grad_()
opt = SparseAdam([{"params": [q], "lr": 5e-4}, {"params": [t], "lr": 1.5e-3}])
fast = "pose_tensors" in slam.render.__code__.co_varnames
def iteration():
    opt.zero_grad(set_to_none=True)
    cam = slam.pose_to_camera(q, t, tanfovx, tanfovy)
    if fast:
        out = slam.render(None, pc, None, bg, viewmatrix=cam[0], pose_tensors=cam, **kw)
    else:
        out = slam.render(None, pc, None, bg, viewmatrix=cam[0], **kw)
    loss = slam.l1_loss(out["render"], out["depth"], obs_c, obs_d, 1.0, 0.5)
    loss.backward()
    opt.step()
    return loss
for _ in range(20): iteration()
torch.cuda.synchronize()
for mt in (True, False):
    torch.autograd.set_multithreading_enabled(mt)
    t0 = time.perf_counter()
    for _ in range(300): iteration()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"autograd engine thread {mt}: host issue {(t1 - t0) / 300 * 1e3:.3f} ms per iteration, with the GPU drained {(t2 - t0) / 300 * 1e3:.3f}")
pr = cProfile.Profile(); pr.enable()
for _ in range(300): iteration()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
