#!/bin/bash
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
timeout 900 python -m pytest tests/test_hip_front_end.py tests/test_hip_heavy_tail.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
DGR_SEG_SHIFT=2 timeout 900 python -m pytest tests/test_hip_front_end.py tests/test_hip_heavy_tail.py tests/test_hip_light_parity.py -x -q -m gpu -k "not config4 and not config5" 2>&1 | grep -v amdgpu.ids | tail -3
python profiles/r9/bin_tiles_trace.py clustered 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tee gpurun_out/r9/bt_trace5.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("bin") or k.startswith("tile")})'
for scene in synth-v1 clustered heavy_tail; do
  echo "== scene $scene"
  python bench.py --no-cpu-baseline --steps 40 --warmup 10 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done 2>&1 | tee gpurun_out/r9/bt_ab5.txt
echo "== synth-v1 DGR_SEG_SHIFT=2 (grouped)"; DGR_SEG_SHIFT=2 python bench.py --no-cpu-baseline --steps 40 --warmup 10 2>/dev/null | tail -1 | python -c "$P" | tee -a gpurun_out/r9/bt_ab5.txt
