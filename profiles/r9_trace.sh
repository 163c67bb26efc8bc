#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
timeout 900 python -m pytest tests/test_hip_front_end.py tests/test_hip_heavy_tail.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -5
for cfg in "4 0" "2 0" "3 0"; do set -- $cfg
  echo "=== DGR_SEG_SHIFT=$1 DGR_BT_MAP=$2"
  DGR_SEG_SHIFT=$1 DGR_BT_MAP=$2 python profiles/r9/bin_tiles_trace.py clustered 2>&1 | grep -v amdgpu.ids | grep -v "^  "
done | tee gpurun_out/r9/bt_trace2.txt
echo "=== synth-v1 default"; python profiles/r9/bin_tiles_trace.py synth-v1 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tee -a gpurun_out/r9/bt_trace2.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("bin")})'
for scene in synth-v1 clustered heavy_tail; do for ss in 4 3 2; do
  echo "== scene $scene SEG_SHIFT=$ss MAP=0"
  DGR_SEG_SHIFT=$ss DGR_BT_MAP=0 python bench.py --no-cpu-baseline --steps 40 --warmup 10 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done; done 2>&1 | tee gpurun_out/r9/bt_ab2.txt
