#!/bin/bash
# Soak on the round's FINAL binaries (half-wave forward, exact tags per half, paired mapping backward, half-wave tracking backward):
# profiles/r8/soak.sh's draws on new seeds, + heavy-tailed draws (DGR_SOAK_HEAVY=1), + the same draws with DGR_FWD_HALVES=0.
cd "$(dirname "$0")/../.."
O=gpurun_out/r8_soak; mkdir -p $O
( timeout 400 python tests/tools/soak_node.py --seconds 80 --seed 71
  timeout 400 python tests/tools/soak_node.py --seconds 80 --seed 72 --drop-inputs
  timeout 900 python tests/tools/soak_parity.py 300 120 71
  DGR_SOAK_HEAVY=1 timeout 900 python tests/tools/soak_parity.py 300 100 72
  DGR_SOAK_HEAVY=1 DGR_FWD_HALVES=0 timeout 900 python tests/tools/soak_parity.py 200 60 73
  DGR_DETERMINISTIC_GRADS=1 timeout 900 python tests/tools/soak_parity.py 200 0 74
  timeout 700 python tests/tools/soak_batch.py ) 2>&1 | grep -v amdgpu.ids | grep "soak_node\|FAIL\|AMBIG\|draws in\|draws,\|MISMATCH\|Error\|error" | cut -c1-400 | tee $O/soak_final2.txt
