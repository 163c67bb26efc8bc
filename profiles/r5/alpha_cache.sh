#!/bin/bash
# Round 5, experiment 5: the alpha cache (the forward keeps o G per blended pair, the backward reads it): parity and time.
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_alpha_cache; mkdir -p $O
DGR_ALPHA_CACHE=1 timeout 1500 python -m pytest tests/test_hip_light_parity.py tests/test_hip_edge_cases.py tests/test_hip_batch.py tests/test_hip_random_sweep.py tests/test_hip_guarded_buffers.py tests/test_slam_render.py -q -m gpu -k "not config5" 2>&1 | grep -v amdgpu.ids | tail -15 > $O/pytest_cache_on.txt
DGR_ALPHA_CACHE=1 python tests/tools/error_budget.py 500000 1920 1080 2>/dev/null | tail -1 > $O/error_budget_cache_on.json
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
for c in 0 1; do for rep in 1 2; do
  DGR_ALPHA_CACHE=$c python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "$P" > $O/bench_cache${c}_$rep.txt
done; done
for c in 0 1; do
  DGR_ALPHA_CACHE=$c python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "$P" > $O/bench_drivercmd_cache${c}.txt
  DGR_ALPHA_CACHE=$c DGR_FAST_ALPHA=1 python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | python -c "$P" > $O/bench_fast_cache${c}.txt
  DGR_ALPHA_CACHE=$c python bench.py --no-cpu-baseline --steps 100 --tracking 2>/dev/null | tail -1 | python -c "$P" > $O/bench_tracking_cache${c}.txt
  DGR_ALPHA_CACHE=$c python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config4_cache${c}.txt
done
cat $O/pytest_cache_on.txt; python -c "
import json; d=json.load(open('$O/error_budget_cache_on.json')); print({k:(d[k]['differing_values']) for k in d if k.startswith('img_')}, {k:'%.1e'%x['max_abs'] for k,x in d['end_to_end'].items()})"
for f in $O/bench_*.txt; do echo $f; cat $f; done
