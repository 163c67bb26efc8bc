#!/bin/bash
# Quick check of a kernel change: targeted parity tests, per-stage kernel durations with one view at a time (dispatch-
# packet events), and the default bench twice.  Usage: bash profiles/quick_ab.sh
cd "$(dirname "$0")/.."
timeout 1500 python -m pytest tests/test_hip_wave_reduce.py tests/test_hip_light_parity.py tests/test_hip_full_parity.py tests/test_hip_random_sweep.py tests/test_hip_edge_cases.py -x -q -m gpu 2>&1 | tail -15
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
python bench.py --no-cpu-baseline --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P"
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"; done
python bench.py --no-cpu-baseline --variant full --workload config2 --graph 2>/dev/null | tail -1 | python -c "$P"
