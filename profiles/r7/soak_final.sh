#!/bin/bash
# Round 7, final binaries: every soak once more.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_soak_final; mkdir -p $O
( timeout 400 python tests/tools/soak_node.py --seconds 120 --seed 31
  timeout 400 python tests/tools/soak_node.py --seconds 120 --seed 32 --drop-inputs
  timeout 700 python tests/tools/soak_parity.py 300 120 31
  timeout 700 python tests/tools/soak_batch.py ) 2>&1 | grep -v amdgpu.ids | grep "soak_node\|FAIL\|AMBIG\|draws in\|draws,\|MISMATCH\|Error" | cut -c1-400 | tee $O/soak.txt
