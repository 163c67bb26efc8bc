#!/bin/bash
# one more pass of the oracle soak on the round's LAST binaries (seeds 101-104; both variants, a third heavy-tailed)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r9
( DGR_SOAK_HEAVY=1 timeout 900 python tests/tools/soak_parity.py 300 150 101
  timeout 900 python tests/tools/soak_parity.py 300 150 102
  DGR_SOAK_HEAVY=1 DGR_FWD_HALVES=1 timeout 900 python tests/tools/soak_parity.py 200 0 103
  timeout 700 python tests/tools/soak_batch.py ) 2>&1 | grep -v amdgpu.ids | grep "FAIL\|AMBIG\|draws in\|draws,\|MISMATCH\|Error\|error" | cut -c1-420 | tee gpurun_out/r9/soak_last.txt
