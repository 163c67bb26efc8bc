#!/bin/bash
# bin_tiles experiments: block -> segment map (DGR_BT_MAP 1 = XCD-contiguous, 0 = identity), register-sort rule threshold
# (DGR_BT_RULE_MAX 1024 / 512), segment size
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("bin")})'
for scene in clustered synth-v1 heavy_tail; do
for ss in 4 3 2; do for map in 1 0; do for rule in 1024 512; do
  echo "== scene $scene SEG_SHIFT=$ss MAP=$map RULE=$rule"
  DGR_SEG_SHIFT=$ss DGR_BT_MAP=$map DGR_BT_RULE_MAX=$rule python bench.py --no-cpu-baseline --steps 40 --warmup 10 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done; done; done; done 2>&1 | tee gpurun_out/r9/bt_ab.txt
