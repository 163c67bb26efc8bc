#!/bin/bash
# whole GPU suite (no -x)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5_suite
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -8 > gpurun_out/r5_suite/pytest.txt
python bench.py --steps 20 --warmup 5 --graph --no-cpu-baseline 2>&1 | tail -3 > gpurun_out/r5_suite/graph_driver_cmd.txt
cat gpurun_out/r5_suite/pytest.txt; tail -c 600 gpurun_out/r5_suite/graph_driver_cmd.txt
