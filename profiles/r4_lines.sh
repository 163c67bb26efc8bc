#!/bin/bash
# round-4 bench lines other than the default one (run on the GPU box; copies land in gpurun_out/lines/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lines
B="python bench.py --no-cpu-baseline"
$B --tight-cull 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config3_light_tight_cull.json
$B --tracking 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config3_light_tracking.json
$B --variant full --workload config2 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config2_full.json
$B --variant full --workload config2 --graph 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config2_full_graph.json
$B --workload config4 --steps 60 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config4_light_view.json
$B --workload config5 --steps 30 --warmup 6 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config5_light_view.json
$B --workload config4 --batch 4 --steps 15 --warmup 4 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config4_batch4.json
$B --workload config5 --batch 4 --steps 10 --warmup 3 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config5_batch4.json
$B --workload config2 --batch 4 --steps 100 2>/dev/null | tail -1 > gpurun_out/lines/r4_bench_config2_light_batch4.json
for f in gpurun_out/lines/r4_*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],4), round(d['config']['ms_per_view'],4), d['config'].get('ms_per_view_one_stream') and round(d['config']['ms_per_view_one_stream'],4))"; done
