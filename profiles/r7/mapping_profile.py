"""Round 7: where the host time of examples/mapping.py --fused (eager) goes: cProfile over 60 iterations.
usage (GPU box): python profiles/r7/mapping_profile.py"""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "diff-gaussian-rasterization_amd"), os.path.join(ROOT, "examples")]
os.environ.setdefault("DGR_SYNC_MODE", "lazy")
import torch
import mapping
torch.autograd.set_multithreading_enabled(False)
dev = torch.device("cuda:0")
pr = cProfile.Profile()
pr.enable()
(l0, l1), pc, dt = mapping.mapping_loop(dev, 100000, 640, 480, 4, 63, fused=True)
pr.disable()
print(f"{dt * 1e3:.3f} ms per iteration under the profiler")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
