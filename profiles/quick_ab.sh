#!/bin/bash
# Quick check of a kernel change: targeted parity tests, per-stage kernel durations with one view at a time (dispatch-
# packet events), and the default bench.  Usage: bash profiles/quick_ab.sh [pytest -k expression | skip]
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/ab
K=${1:-"not config5 and not config4"}
if [ "$K" != "skip" ]; then
timeout 1500 python -m pytest tests/test_hip_wave_reduce.py tests/test_hip_light_parity.py tests/test_hip_full_parity.py tests/test_hip_random_sweep.py tests/test_hip_edge_cases.py -x -q -m gpu -k "$K" 2>&1 | grep -v amdgpu.ids | tail -12
fi
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | tee gpurun_out/ab/bench.json | python -c "$P"
python bench.py --no-cpu-baseline --steps 100 2>/dev/null | tail -1 | python -c "$P"
