#!/bin/bash
# Round 6, evidence at HEAD: rocprofv3 kernel trace + PMC passes (profiles/run_profile.sh), the bench lines of the round
# (uniform benchmark scene and the clustered one), the whole GPU suite.
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash profiles/run_profile.sh r6 > gpurun_out/prof_r6_run.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize.py gpurun_out/prof_r6 gpurun_out/r6 > gpurun_out/prof_r6_summary.log 2>&1
L=gpurun_out/r6_lines; mkdir -p $L gpurun_out/r6_final
python bench.py --steps 200 2>$L/err.txt | tail -1 > $L/bench_config3_light.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $L/bench_config3_light_driver_cmd.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $L/bench_config3_light_driver_cmd_2.json
python bench.py --steps 200 --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_one_stream.json
python bench.py --steps 100 --sync-mode strict --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_strict.json
python bench.py --steps 100 --sync-mode strict --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_strict_one_stream.json
DGR_FAST_ALPHA=1 python bench.py --steps 200 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_fast_alpha.json
python bench.py --steps 100 --tracking --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_tracking.json
python bench.py --steps 100 --tight-cull --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_tight_cull.json
python bench.py --steps 50 --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_batch4.json
python bench.py --steps 30 --batch 8 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_batch8.json
python bench.py --steps 200 --graph --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_graph.json
python bench.py --steps 100 --scene clustered --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_clustered.json
python bench.py --steps 100 --scene clustered --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config3_light_clustered_one_stream.json
python bench.py --steps 30 --workload config4 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config4_light_view.json
python bench.py --steps 20 --workload config5 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config5_light_view.json
python bench.py --steps 100 --variant full --workload config2 --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config2_full.json
python bench.py --steps 100 --variant full --workload config2 --graph --no-cpu-baseline 2>/dev/null | tail -1 > $L/bench_config2_full_graph.json
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 > gpurun_out/r6_final/pytest.txt
tail -3 $L/err.txt
for f in $L/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], "ms/step", round(d["ms_per_step"],4), "serial", d["config"].get("ms_per_view_one_stream"), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, (d.get("roofline_valu") or {}).get("render_bwd",{}).get("frac"))
except Exception as e: print(sys.argv[1], "FAILED", e)
PY
done
cat gpurun_out/r6_final/pytest.txt; head -14 gpurun_out/r6_kernel_stats.txt
