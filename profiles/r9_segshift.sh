#!/bin/bash
# bin_tiles / bin_segments by segment size (DGR_SEG_SHIFT = 4, 3, 2: 16, 8, 4 tiles per segment) on the three scenes
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
for scene in synth-v1 clustered heavy_tail; do
for ss in 4 3 2; do
  echo "== scene $scene DGR_SEG_SHIFT=$ss"
  DGR_SEG_SHIFT=$ss python bench.py --no-cpu-baseline --steps 60 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done; done 2>&1 | tee gpurun_out/r9/segshift.txt
for ss in 4 3 2; do echo "== config2 light DGR_SEG_SHIFT=$ss"; DGR_SEG_SHIFT=$ss python bench.py --no-cpu-baseline --steps 60 --workload config2 2>/dev/null | tail -1 | python -c "$P"; done 2>&1 | tee -a gpurun_out/r9/segshift.txt
