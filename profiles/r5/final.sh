#!/bin/bash
# Round 5, final evidence at HEAD: rocprofv3 kernel trace + PMC passes, the bench lines, the whole GPU suite
cd "$(dirname "$0")/../.."
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-$PWD}
bash profiles/run_profile.sh r5 > gpurun_out/prof_r5_run.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize.py gpurun_out/prof_r5 gpurun_out/r5 > gpurun_out/prof_r5_summary.log 2>&1
bash profiles/r5/lines.sh > gpurun_out/r5_lines_run.log 2>&1
python bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_driver_cmd_graph.json
python bench.py --steps 200 --graph --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r5_lines/bench_config3_light_graph.json
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 > gpurun_out/r5_final/pytest.txt
tail -22 gpurun_out/r5_lines_run.log
for f in gpurun_out/r5_lines/bench_config3_light_driver_cmd_graph.json gpurun_out/r5_lines/bench_config3_light_graph.json; do python -c "
import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],4))"; done
cat gpurun_out/r5_final/pytest.txt; head -12 gpurun_out/r5_kernel_stats.txt
