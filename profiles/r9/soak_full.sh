#!/bin/bash
# After the full variant got tag bytes per half and paired lists: its draws of the soaks again (seeds 91-93), the node soak and the thread soak
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r9
( timeout 900 python tests/tools/soak_parity.py 0 250 91
  DGR_SOAK_HEAVY=1 timeout 900 python tests/tools/soak_parity.py 0 200 92
  DGR_SOAK_HEAVY=1 DGR_DETERMINISTIC_GRADS=1 timeout 900 python tests/tools/soak_parity.py 0 120 93
  timeout 300 python tests/tools/soak_node.py --seconds 60 --seed 91
  timeout 300 python tests/tools/soak_threads.py --seconds 40 --threads 4 --seed 9 ) 2>&1 | grep -v amdgpu.ids | grep "soak_node\|soak_threads\|PROBLEM\|FAIL\|AMBIG\|draws in\|draws,\|MISMATCH\|Error\|error" | cut -c1-400 | tee gpurun_out/r9/soak_full.txt
