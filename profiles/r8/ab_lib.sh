#!/bin/bash
# A/B of library builds at config 3 on one box: bash profiles/r8/ab_lib.sh <name|default> ...   (names: lib/libdgr_hip_<name>.so,
# profiles/r6/build_variant.sh).  Stage times one view at a time, ms per view with views in flight over 100 steps; alternating, twice.
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
for rep in 1 2; do
for n in "$@"; do
  if [ "$n" = default ]; then export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip.so; else export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip_$n.so; fi
  python bench.py --no-cpu-baseline --steps 100 $AB_EXTRA 2>/dev/null | tail -1 | python -c "$P" $n
done
done
