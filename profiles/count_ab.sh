#!/bin/bash
# LDS count (forced: DGR_LDS_COUNT=2) against round 2's global-atomic count on the same box: bash profiles/count_ab.sh [workloads...]
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print("   3 in flight", round(d["ms_per_step"],4), "| one stream", round(c["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in c["stage_ms"].items() if k in ("preprocess_fwd","count_rank","scan_tiles","emit_instances","zero_counters")})'
for w in "${@:-config3}"; do
  for m in 2 0 2 0; do
    echo "$w DGR_LDS_COUNT=$m"
    DGR_LDS_COUNT=$m python bench.py --workload $w --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"
  done
done
