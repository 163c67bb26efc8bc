#!/bin/bash
# Soak on round 9's FINAL binaries (lane lists by the frame, both mappings in one kernel; 64-byte record slots; tag bytes only): the
# draws of profiles/r8/soak_final.sh on new seeds; heavy-tailed thirds (frames that take quadrant lists by themselves) and both
# mappings forced.
cd "$(dirname "$0")/../.."
O=gpurun_out/r9_soak; mkdir -p $O
( timeout 300 python tests/tools/soak_node.py --seconds 60 --seed 81
  timeout 300 python tests/tools/soak_node.py --seconds 60 --seed 82 --drop-inputs
  timeout 900 python tests/tools/soak_parity.py 250 100 81
  DGR_SOAK_HEAVY=1 timeout 900 python tests/tools/soak_parity.py 250 80 82
  DGR_SOAK_HEAVY=1 DGR_FWD_HALVES=0 timeout 900 python tests/tools/soak_parity.py 150 0 83
  DGR_SOAK_HEAVY=1 DGR_FWD_HALVES=1 timeout 900 python tests/tools/soak_parity.py 150 0 84
  DGR_SOAK_HEAVY=1 DGR_DETERMINISTIC_GRADS=1 timeout 900 python tests/tools/soak_parity.py 150 0 85
  timeout 700 python tests/tools/soak_batch.py ) 2>&1 | grep -v amdgpu.ids | grep "soak_node\|FAIL\|AMBIG\|draws in\|draws,\|MISMATCH\|Error\|error" | cut -c1-400 | tee $O/soak.txt
