#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5_examples
timeout 600 python examples/tracking.py --fused --graph 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r5_examples/tracking_fused_graph.txt
timeout 600 python examples/tracking.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r5_examples/tracking.txt
timeout 600 python examples/mapping.py --fused --graph 2>&1 | grep -v amdgpu.ids | tail -4 > gpurun_out/r5_examples/mapping_fused_graph.txt
timeout 600 python examples/mapping.py 2>&1 | grep -v amdgpu.ids | tail -3 > gpurun_out/r5_examples/mapping.txt
python bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("graph, driver steps:", round(d["ms_per_step"],4))' > gpurun_out/r5_examples/graph_driver_cmd.txt
for f in gpurun_out/r5_examples/*.txt; do echo "== $f"; cat $f; done
