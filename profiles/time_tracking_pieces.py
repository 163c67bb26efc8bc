import os, sys, time
ROOT = "/root/repo"
sys.path[:0] = [ROOT, ROOT + "/diff-gaussian-rasterization_amd", ROOT + "/tests"]
import numpy as np, torch
from dgr_amd import slam
from dgr_amd.synth import camera, make_scene
from test_slam_render import Model, rot_to_quat
dev = torch.device("cuda:0")
W, H = 640, 480
s = make_scene(100000, W, H, 3)
pc = Model(s, dev)
tanfovx, tanfovy, Rm, t_true, *_ = camera(W, H, 0.05)
bg, gt_depth = torch.from_numpy(s.bg).to(dev), torch.from_numpy(s.gt).to(dev)
kw = dict(fov=(tanfovx, tanfovy), HW=(H, W), gt_depth=gt_depth, track_off=False, map_off=True)
q = torch.tensor(rot_to_quat(Rm), dtype=torch.float32, device=dev).requires_grad_()
t = torch.tensor(t_true, dtype=torch.float32, device=dev).requires_grad_()
opt = torch.optim.Adam([q, t], lr=1e-4)
obs_c = torch.rand((3, H, W), device=dev); obs_d = torch.rand((1, H, W), device=dev)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0; return time.perf_counter()
for it in range(60):
    if it == 10: T.clear()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    opt.zero_grad(set_to_none=True); t0 = tick("zero_grad", t0)
    w2c = slam.w2c_from_quat_trans(q, t); t0 = tick("w2c_from_quat", t0)
    vm = slam.camera_tensors(w2c, tanfovx, tanfovy)[0]; t0 = tick("camera_tensors", t0)
    out = slam.render(None, pc, None, bg, viewmatrix=vm, **kw); t0 = tick("render", t0)
    loss = (out["render"] - obs_c).abs().mean() + 0.5 * (out["depth"] - obs_d).abs().mean(); t0 = tick("loss", t0)
    loss.backward(); t0 = tick("backward", t0)
    opt.step(); t0 = tick("adam", t0)
for k, v in T.items():
    print(f"{k:16s} {v / 50 * 1e3:7.3f} ms")

# kernel stages of the rasterizer inside one tracking iteration (tracking mode: map_off)
from dgr_amd import _capi
_capi.set_option("profile_every", 1)
_capi.profile_select("all")
for it in range(20):
    opt.zero_grad(set_to_none=True)
    vm = slam.camera_tensors(slam.w2c_from_quat_trans(q, t), tanfovx, tanfovy)[0]
    out = slam.render(None, pc, None, bg, viewmatrix=vm, **kw)
    loss = (out["render"] - obs_c).abs().mean() + 0.5 * (out["depth"] - obs_d).abs().mean()
    loss.backward()
torch.cuda.synchronize()
tot = 0.0
for name in _capi.profile_stages():
    ms, n = _capi.profile_read(name)
    if n:
        print(f"  stage {name:16s} {1e3 * ms / n:7.1f} us")
        tot += ms / n
print(f"  rasterizer kernels per iteration: {1e3 * tot:.1f} us")
_capi.profile_select("")
