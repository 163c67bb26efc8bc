"""sha256 over the kernel sources (csrc/*.hip, *.h), first 16 hex digits: profiles/make_pmc_traffic.py stores it with the PMC
counters, bench.py recomputes it and reports whether the committed counters were taken on the kernels it is running."""
import glob
import hashlib
import os


def kernel_sources_sha():
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diff-gaussian-rasterization_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]
