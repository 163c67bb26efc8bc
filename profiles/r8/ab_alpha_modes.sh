#!/bin/bash
# A/B of the exact alpha paths on one box at config 3: alpha_mode 0 (fp32 polynomial expf, round 8's default) against
# alpha_mode 2 (glibc's expf in the double pipe, rounds 5-7's default) and 1 (fast).  Stage times one view at a time
# (dispatch-packet events), ms per view with views in flight over 100 steps; alternating, twice.
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); print("alpha_mode", sys.argv[1], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
for rep in 1 2; do
for m in 2 0 1; do
  DGR_ALPHA_MODE=$m python bench.py --no-cpu-baseline --steps 100 $AB_EXTRA 2>/dev/null | tail -1 | python -c "$P" $m
done
done
