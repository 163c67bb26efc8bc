#!/bin/bash
# A/B of the tracking step's lane mapping on one box at config 3 (bench.py --tracking: map_off, pose gradient only): DGR_FWD_HALVES=0
# (one list per quadrant wave in forward and backward) against 1 (one per half-wave in both).  Alternating, twice, 100 steps.
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); print("halves", sys.argv[1], sys.argv[2], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for rep in 1 2; do
for m in 0 1; do
  DGR_FWD_HALVES=$m python bench.py --no-cpu-baseline --steps 100 --tracking 2>/dev/null | tail -1 | python -c "$P" $m tracking
  DGR_FWD_HALVES=$m python bench.py --no-cpu-baseline --steps 100 --tracking --lean-loss 2>/dev/null | tail -1 | python -c "$P" $m tracking-lean
done
done
