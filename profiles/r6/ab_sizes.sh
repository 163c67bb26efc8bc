#!/bin/bash
# A/B of two library builds at the sizes of BASELINE configs 3, 4, 5 (stage times one view at a time, three views in flight)
cd "$(dirname "$0")/../.."
P='import sys,json; d=json.loads(sys.stdin.read()); print(sys.argv[1], sys.argv[2], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for wl in "config3 100" "config4 30" "config5 20"; do
  set -- $wl
  for rep in 1 2; do
  for n in r5 default; do
    if [ "$n" = default ]; then export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip.so; else export DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip_$n.so; fi
    python bench.py --no-cpu-baseline --workload $1 --steps $2 2>/dev/null | tail -1 | python -c "$P" $1 $n
  done
  done
done
