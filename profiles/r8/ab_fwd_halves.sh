#!/bin/bash
# A/B of the light forward's lane mapping on one box at config 3: DGR_FWD_HALVES=0 (one list per quadrant wave) against 1 (one list
# per half-wave).  Stage times one view at a time, ms per view with views in flight over 100 steps; alternating, twice.
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); print("fwd_halves", sys.argv[1], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "err", d["config"].get("grad_max_abs_err",{}).get("max"))'
for rep in 1 2; do
for m in 0 1; do
  DGR_FWD_HALVES=$m python bench.py --no-cpu-baseline --steps 100 $AB_EXTRA 2>/dev/null | tail -1 | python -c "$P" $m
done
done
