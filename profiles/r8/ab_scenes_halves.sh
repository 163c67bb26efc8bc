#!/bin/bash
# The round's lane mappings (DGR_FWD_HALVES=1: half-wave forward, paired mapping backward) against rounds 1-7's (=0) on the two
# non-uniform scenes of bench.py, one box, alternating, 100 steps.
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "halves", sys.argv[2], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render")})'
for rep in 1 2; do for sc in heavy_tail clustered; do for m in 0 1; do
  DGR_FWD_HALVES=$m python bench.py --no-cpu-baseline --steps 100 --scene $sc 2>/dev/null | tail -1 | python -c "$P" $sc $m
done; done; done
