#!/bin/bash
# Sweep of the tile-counter padding (DGR_COUNT_STRIDE builds under lib/exp/): prints ms/view, count_rank, scan_tiles.
for s in "$@"; do
  for w in config2 config3; do
    DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/exp/libdgr_s$s.so python bench.py --workload $w --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > /tmp/line.json
    python - "$s" "$w" <<'PY'
import json, sys
d = json.load(open("/tmp/line.json")); st = d["config"]["stage_ms"]
print("stride", sys.argv[1], sys.argv[2], round(d["ms_per_step"], 4), "count_rank", st["count_rank"], "scan_tiles", st["scan_tiles"])
PY
  done
done
# Result (r1, MI355X, us): stride  config2 count_rank/scan_tiles   config3 count_rank/scan_tiles
#                            1        76 / 7                          97 / 13
#                            4        29 / 6.5                        71.5 / 12.8
#                            8        28 / 6.3                        70.7 / 13.8
#                           16        23 / 6.7                        70.7 / 16.7
#                           32        20 / 6.4                        69.9 / 17.2
# Build a variant with: hipcc ... -DDGR_COUNT_STRIDE=<s> -shared -o lib/exp/libdgr_s<s>.so csrc/*.hip
