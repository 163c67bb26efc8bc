#!/bin/bash
# A/B of builds of the library, one view at a time AND with the default three views in flight (config 3):
#   bash profiles/lib_sweep2.sh <lib.so> [<lib.so> ...]
P='import sys,json; d=json.loads(sys.stdin.read()); print("   3 in flight", round(d["ms_per_step"],4), "| one stream", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for lib in "$@"; do
  echo "$lib"
  for i in 1 2; do DGR_HIP_LIB=$PWD/$lib python bench.py --workload config3 --steps 100 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"; done
done
