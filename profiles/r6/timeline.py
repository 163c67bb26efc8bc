#!/usr/bin/env python3
"""Round 6: GPU timeline of bench.py's timed region from a rocprofv3 kernel trace (csv).

usage: timeline.py <kernel_trace.csv> <steps>
Prints, for the last <steps> views (= the timed region: the last <steps> render_bwd launches and what belongs to them):
 * per group of 5 views: wall time from the first kernel start to the last kernel end, GPU idle time inside, average
   duration of every kernel kind (under overlap);
 * how much of the wall time had 1 / 2 / 3+ dgr kernels resident, and which pairs of kinds overlapped most.
"""
import csv, sys, collections

path, steps = sys.argv[1], int(sys.argv[2])
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), name, r.get("Queue_Id", "?")))
rows.sort()
def kind(n):
    for k in ("render_bwd", "render_fwd", "preprocess_bwd", "preprocess_fwd", "bin_tiles", "bin_segments", "tile_schedule", "zero_fill"):
        if k in n: return k
    return "other"
bwd = [i for i, r in enumerate(rows) if kind(r[2]) == "render_bwd"]
first_bwd = bwd[-steps]
# the timed region starts with the preprocess_fwd of the view whose backward is bwd[-steps]: walk back to the 1st preprocess_fwd
# after the previous region's last kernel (a gap > 100 us separates them: drain + barrier)
i0 = first_bwd
while i0 > 0 and rows[i0][0] - max(r[1] for r in rows[max(0, i0 - 12):i0]) < 100_000 and i0 > bwd[-steps - 1]:
    i0 -= 1
last = bwd[-1]
i1 = last
while i1 + 1 < len(rows) and kind(rows[i1 + 1][2]) == "preprocess_bwd" and rows[i1 + 1][0] - rows[last][1] < 200_000:
    i1 += 1
reg = rows[i0:i1 + 1]
t0 = reg[0][0]
print(f"region: {len(reg)} kernels, {(max(r[1] for r in reg) - t0) / 1e6:.3f} ms, queues {sorted(set(r[3] for r in reg))}")
# sweep
ev = []
for s, e, n, q in reg:
    ev.append((s, 1, kind(n))); ev.append((e, -1, kind(n)))
ev.sort()
active = collections.Counter(); tprev = t0
occ = collections.Counter(); pair = collections.Counter()
seg = []  # (t, t_next, frozenset kinds)
for t, d, k in ev:
    if t > tprev:
        ks = tuple(sorted(k_ for k_ in active.elements() if k_ != "other"))
        occ[len(ks)] += t - tprev
        pair[ks] += t - tprev
        seg.append((tprev, t, ks))
    active[k] += d
    if active[k] == 0: del active[k]
    tprev = t
tot = sum(occ.values())
print("resident dgr kernels -> share of wall time:", {k: round(v / tot, 3) for k, v in sorted(occ.items())})
print("most common resident sets:")
for ks, v in pair.most_common(12):
    print(f"  {v / tot:6.3f}  {' + '.join(ks) if ks else '(idle)'}")
# by group of 5 views (by render_bwd end)
ends = [rows[i][1] for i in bwd[-steps:]]
prev = t0
for g in range(0, steps, 5):
    tend = ends[min(g + 4, steps - 1)]
    ks = [r for r in reg if prev <= r[1] <= tend]
    idle = sum(b - a for a, b, s in seg if not s and a >= prev and b <= tend)
    d = collections.defaultdict(list)
    for s, e, n, q in ks: d[kind(n)].append((e - s) / 1e3)
    print(f"views {g:3d}-{g + 4:3d}: {(tend - prev) / 5e6:.4f} ms/view, idle {idle / 1e3:6.1f} us; " +
          " ".join(f"{k}={sum(v) / len(v):.0f}" for k, v in sorted(d.items()) if k != "other"))
    prev = tend

# per-view intervals (a view = the kernels between one preprocess_fwd and the next preprocess_bwd on the same queue)
if len(sys.argv) > 3:
    nshow = int(sys.argv[3])
    byq = collections.defaultdict(list)
    for s, e, n, q in reg:
        if kind(n) != "other": byq[q].append((s, e, kind(n)))
    views = []
    for q, ks in byq.items():
        cur = []
        for s, e, k in ks:
            cur.append((s, e, k))
            if k == "preprocess_bwd":
                views.append((cur[0][0], q, cur)); cur = []
    views.sort()
    print("view  queue  start_ms | per kernel: start-end (ms from region start)")
    for i, (s0, q, ks) in enumerate(views[:nshow]):
        print(f"{i:3d}  q{q}  " + "  ".join(f"{k[:7]}{'' if not k.startswith('render') else k[6:10]}:{(s - t0) / 1e6:.3f}-{(e - t0) / 1e6:.3f}" for s, e, k in ks))
