"""Round 7 (after profiles/r6/slot_probe.py; PROBE_PRIORITY=-1 puts the probe on a high-priority stream, DGR_BLEND_WGS_PER_CU=7 caps the
blend kernels with the LDS-leaving claim).  Round 6: does a kernel that FITS into one freed blend slot (<= 64 VGPRs, 256 threads, no LDS) make progress while a blend
kernel holds the chip?  Stream A loops the light backward (or the forward); stream B loops a torch streaming kernel
(c = a + b over 3 x 64 MB: a few dozen VGPRs).  Each alone, then together.  If B's iterations disappear in A's shadow the front
end of another view would too, were it shaped like B; preprocess_fwd (96 VGPRs, 27 KB of LDS) is known to starve
(profiles/r6/timeline.txt).  Usage (GPU box): python profiles/r6/slot_probe.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
os.environ["DGR_SYNC_MODE"] = "lazy"
import numpy as np, torch
from dgr_amd import light as L
from dgr_amd.synth import make_scene
import hip_helpers as hh

P, W, H, deg = 500000, 1920, 1080, 3
s = make_scene(P, W, H, 0)
out, d = hh.hip_forward(s, deg)
T, E = hh.T, hh.E
fargs = (T(s.bg), T(s.means), E(), T(s.opac), T(s.scales), T(s.rots), 1.0, E(), T(s.view), T(s.gt), T(s.proj), s.tanfovx,
         s.tanfovy, s.H, s.W, T(s.shs), deg, T(s.campos), False, False)
(R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
bargs = (T(s.bg), T(s.means), radii, E(), T(s.scales), T(s.rots), 1.0, E(), T(s.view), T(s.proj), s.tanfovx, s.tanfovy,
         T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None]), T(s.gt), T(s.shs), deg, T(s.campos), geom, R, binning, img,
         alpha, False, T(s.persp), False, False)
fwd = lambda: L._C.rasterize_gaussians(*fargs)
bwd = lambda: L._C.rasterize_gaussians_backward(*bargs)
n16 = 16 * 1024 * 1024
xa, xb, xc = (torch.ones(n16, device="cuda") for _ in range(3))
probe = lambda: torch.add(xa, xb, out=xc)   # 3 x 64 MB moved
for _ in range(5):
    fwd(); bwd(); probe()
torch.cuda.synchronize()
A, B = torch.cuda.Stream(), torch.cuda.Stream(priority=int(os.environ.get("PROBE_PRIORITY", "0")))
print("probe priority", os.environ.get("PROBE_PRIORITY", "0"), "blend cap", os.environ.get("DGR_BLEND_WGS_PER_CU", "0"), flush=True)


def run(fa, na, nb):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(max(na, nb)):
        if i < na:
            with torch.cuda.stream(A):
                fa()
        if i < nb:
            with torch.cuda.stream(B):
                probe()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for fa in (bwd, fwd):  # warm both streams (allocator pools are per stream)
    run(fa, 20, 20)
for name, fa in (("backward (zero_fill + render_bwd + preprocess_bwd)", bwd), ("forward (preprocess_fwd + binning + render_fwd)", fwd)):
    na = 100
    ta = min(run(fa, na, 0) for _ in range(3))
    tb1 = min(run(fa, 0, 100) for _ in range(3)) / 100
    for ratio in (2,):
        nb = na * ratio
        tb = tb1 * nb
        tab = min(run(fa, na, nb) for _ in range(3))
        print(f"{name}: A alone {ta / na:.3f} ms/iter; probe alone {tb1 * 1e3:.1f} us/iter; {ratio} probes per A-iteration: "
              f"A {ta:.1f} ms + B {tb:.1f} ms = {ta + tb:.1f} serial, together {tab:.1f} ms (hidden {100 * (ta + tb - tab) / tb:.0f} % of B)")
