#!/bin/bash
# regenerates the committed bench lines other than the default one (run on the GPU box; copy gpurun_out/lines/* to profiles/)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/lines
B="python bench.py --no-cpu-baseline"
$B --views-in-flight 1 2>/dev/null | tail -1 > gpurun_out/lines/r1_bench_config3_light_one_stream.json
$B --sync-mode strict 2>/dev/null | tail -1 > gpurun_out/lines/r1_bench_config3_light_strict.json
$B --tight-cull 2>/dev/null | tail -1 > gpurun_out/lines/r1_bench_config3_light_tight_cull.json
$B --variant full --workload config2 2>/dev/null | tail -1 > gpurun_out/lines/r1_bench_config2_full.json
$B --variant full --workload config2 --graph 2>/dev/null | tail -1 > gpurun_out/lines/r1_bench_config2_full_graph.json
for f in gpurun_out/lines/*.json; do python -c "import json,sys; d=json.load(open('$f')); print('$f'.split('/')[-1], round(d['ms_per_step'],4))"; done
