"""Round 6: how much of the blend kernels' time is the tail of the tile schedule?  Experiment build
(-DDGR_EXPERIMENT_TILE_ORDER: block -> tile through a table) with tables computed on the host from the exported ranges:
default XCD bands; heaviest tile first (LPT) globally; LPT inside each XCD band; lightest first (the worst case).
Usage (GPU box): DGR_HIP_LIB=.../libdgr_hip_tileorder.so python profiles/r6/tile_order_probe.py [clustered]"""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "tests")]
os.environ["DGR_SYNC_MODE"] = "lazy"
import numpy as np, torch
from dgr_amd import _capi, light as L
from dgr_amd.synth import make_scene
import hip_helpers as hh

P, W, H, deg = 500000, 1920, 1080, 3
s = make_scene(P, W, H, 0)
if len(sys.argv) > 1 and sys.argv[1] == "clustered":
    # a non-uniform scene: pull 60 % of the Gaussians towards one third of the frame (x, y scaled about a corner point)
    rng = np.random.default_rng(1)
    pick = rng.random(P) < 0.6
    m = s.means.copy()
    c = m[pick].mean(axis=0)
    m[pick, :2] = c[:2] + 0.35 * (m[pick, :2] - c[:2]) + np.array([0.5, 0.3], np.float32)
    s = s._replace(means=m.astype(np.float32))
out, d = hh.hip_forward(s, deg)
T, E = hh.T, hh.E
fargs = (T(s.bg), T(s.means), E(), T(s.opac), T(s.scales), T(s.rots), 1.0, E(), T(s.view), T(s.gt), T(s.proj), s.tanfovx,
         s.tanfovy, s.H, s.W, T(s.shs), deg, T(s.campos), False, False)
(R, color, depth, median, var, alpha, radii, geom, binning, img, _, _) = out
bargs = (T(s.bg), T(s.means), radii, E(), T(s.scales), T(s.rots), 1.0, E(), T(s.view), T(s.proj), s.tanfovx, s.tanfovy,
         T(s.gC), T(s.gD[None]), T(s.gM[None]), T(s.gV[None]), T(s.gt), T(s.shs), deg, T(s.campos), geom, R, binning, img,
         alpha, False, T(s.persp), False, False)
rg = hh.hip_state("ranges", s, d).reshape(-1, 2).astype(np.int64)
cnt = rg[:, 1] - rg[:, 0]
tiles = len(cnt)
print(f"R = {R}, tiles = {tiles}, list length mean {cnt.mean():.1f} max {cnt.max()} p99 {np.percentile(cnt, 99):.0f} min {cnt.min()}")


def xcd_tile(b, n):
    xcd, local = b & 7, b >> 3
    q, r = n >> 3, n & 7
    base = xcd * (q + 1) if xcd < r else r * (q + 1) + (xcd - r) * q
    return base + local


default = np.array([xcd_tile(b, tiles) for b in range(tiles)], np.uint32)
lpt = np.argsort(-cnt, kind="stable").astype(np.uint32)
lpt_xcd = default.copy()
for x in range(8):
    blocks = np.arange(x, tiles, 8)
    band = default[blocks]
    lpt_xcd[blocks] = band[np.argsort(-cnt[band], kind="stable")]
worst = np.argsort(cnt, kind="stable").astype(np.uint32)
lib = _capi.load()
lib.dgr_debug_set_tile_order.argtypes = [C.c_void_p]
_capi.set_option("profile_every", 1)
_capi.profile_select("all")


def measure(name, table):
    keep = None
    if table is None:
        lib.dgr_debug_set_tile_order(None)
    else:
        assert sorted(table.tolist()) == list(range(tiles))
        keep = torch.from_numpy(table.astype(np.int32)).cuda()
        lib.dgr_debug_set_tile_order(keep.data_ptr())
    torch.cuda.synchronize()
    for st in _capi.profile_stages():
        _capi.profile_read(st)
    for _ in range(30):
        o = L._C.rasterize_gaussians(*fargs)
        b = list(bargs); b[20], b[21], b[22], b[23], b[24] = o[7], o[0], o[8], o[9], o[5]
        L._C.rasterize_gaussians_backward(*b)
    torch.cuda.synchronize()
    r = {}
    for st in ("render_fwd", "render_bwd"):
        tot, n = _capi.profile_read(st)
        r[st] = tot / max(n, 1) * 1e3
    print(f"{name:28s} render_fwd {r['render_fwd']:.1f} us  render_bwd {r['render_bwd']:.1f} us")


for rep in range(2):
    measure("built-in XCD bands", None)
    measure("table: XCD bands", default)
    measure("table: LPT inside XCD bands", lpt_xcd)
    measure("table: LPT global", lpt)
    measure("table: lightest first", worst)
