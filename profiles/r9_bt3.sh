#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("bin") or k.startswith("tile")})'
for scene in synth-v1 clustered heavy_tail; do for map in "" 0 1; do
  echo "== scene $scene (policy) MAP=${map:-policy}"
  DGR_BT_MAP=$map python bench.py --no-cpu-baseline --steps 40 --warmup 10 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done; done 2>&1 | tee gpurun_out/r9/bt_ab3.txt
for wl in config2 config4 config5; do echo "== $wl"; python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload $wl 2>/dev/null | tail -1 | python -c "$P"; done 2>&1 | tee -a gpurun_out/r9/bt_ab3.txt
