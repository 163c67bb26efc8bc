#!/usr/bin/env python3
"""Round 7: how much of bench.py's timed region has NO blend kernel resident, from a rocprofv3 kernel trace (csv).

usage: blend_residency.py <kernel_trace.csv> <steps>
The blend kernels (render_fwd / render_bwd) are VALU-bound and hold every wave slot; the other kernels of a view are
bandwidth- or latency-bound.  Wall time = (time with >= 1 blend kernel resident) + (time with none).  If the first term
equals the blend kernels' stand-alone total, concurrent blends share the chip without loss and ALL of the distance to
perfect overlap is the second term: intervals in which every view in flight is in its front end or its per-Gaussian
backward at the same time."""
import collections, csv, sys

path, steps = sys.argv[1], int(sys.argv[2])
rows = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(path)))


def kind(n):
    for k in ("render_bwd", "render_fwd", "preprocess_bwd", "preprocess_fwd", "bin_tiles", "bin_segments", "tile_schedule"):
        if k in n:
            return k
    return "other"


bwd = [i for i, r in enumerate(rows) if kind(r[2]) == "render_bwd"]
i0 = bwd[-steps]
while i0 > 0 and rows[i0][0] - max(r[1] for r in rows[max(0, i0 - 12):i0]) < 100_000 and i0 > bwd[-steps - 1]:
    i0 -= 1
i1 = bwd[-1]
while i1 + 1 < len(rows) and kind(rows[i1 + 1][2]) == "preprocess_bwd" and rows[i1 + 1][0] - rows[bwd[-1]][1] < 200_000:
    i1 += 1
reg = [r for r in rows[i0:i1 + 1] if kind(r[2]) != "other"]
t0, t1 = reg[0][0], max(r[1] for r in reg)
ev = sorted([(s, 1, kind(n)) for s, e, n in reg] + [(e, -1, kind(n)) for s, e, n in reg])
act, tp, share, gaps = collections.Counter(), t0, collections.Counter(), []
for t, d, k in ev:
    if t > tp:
        b = act["render_fwd"] + act["render_bwd"]
        share[b] += t - tp
        if b == 0:
            if gaps and abs(gaps[-1][1] - tp) < 10:
                gaps[-1][1] = t
            else:
                gaps.append([tp, t])
    act[k] += d
    tp = t
wall = t1 - t0
alone = {"render_fwd": 0.0, "render_bwd": 0.0}
print(f"timed region: {len(reg)} kernels, {wall / 1e6:.3f} ms = {wall / 1e3 / steps:.1f} us per view")
print("blend kernels resident -> share of wall time:", {k: round(v / wall, 3) for k, v in sorted(share.items())})
print(f"time with >= 1 blend kernel resident: {(wall - share[0]) / 1e6:.3f} ms = {(wall - share[0]) / 1e3 / steps:.1f} us per view"
      f"   (stand-alone render_fwd + render_bwd: see profiles/r7_kernel_stats.txt)")
print(f"time with none: {share[0] / 1e6:.3f} ms = {share[0] / 1e3 / steps:.1f} us per view, in {len(gaps)} intervals; the long ones (us from the start, length):")
print("  ", [(round((a - t0) / 1e3), round((b - a) / 1e3)) for a, b in gaps if b - a > 30_000])
per = collections.defaultdict(list)
for s, e, n in reg:
    per[kind(n)].append((e - s) / 1e3)
print("average duration under overlap (us):", {k: round(sum(v) / len(v)) for k, v in sorted(per.items())})
