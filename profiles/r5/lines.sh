#!/bin/bash
# the bench lines of the round (gpurun_out/r5_lines/) + the batch-vs-oracle test at config 3's size
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5_lines
timeout 900 python -m pytest tests/test_hip_batch.py -q -m gpu -k "against_the_oracle" 2>&1 | grep -v amdgpu.ids | tail -5 > gpurun_out/r5_lines/pytest_batch_oracle.txt
python bench.py --steps 200 2>gpurun_out/r5_lines/err.txt | tail -1 > gpurun_out/r5_lines/bench_config3_light.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_driver_cmd.json
python bench.py --steps 200 --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_one_stream.json
python bench.py --steps 100 --sync-mode strict --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_strict.json
python bench.py --steps 100 --sync-mode strict --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_strict_one_stream.json
DGR_FAST_ALPHA=1 python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_fast_alpha.json
python bench.py --steps 100 --tracking --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_tracking.json
python bench.py --steps 100 --tight-cull --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_tight_cull.json
python bench.py --steps 50 --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_batch4.json
python bench.py --steps 30 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_batch8.json
python bench.py --steps 30 --workload config4 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config4_light_view.json
python bench.py --steps 20 --workload config5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config5_light_view.json
python bench.py --steps 100 --variant full --workload config2 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config2_full.json
python bench.py --steps 100 --variant full --workload config2 --graph --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config2_full_graph.json
cat gpurun_out/r5_lines/pytest_batch_oracle.txt; tail -3 gpurun_out/r5_lines/err.txt
for f in gpurun_out/r5_lines/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", round(d["ms_per_step"],4), "serial", d["config"].get("ms_per_view_one_stream"), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, (d.get("roofline_valu") or {}).get("render_bwd",{}).get("frac"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
