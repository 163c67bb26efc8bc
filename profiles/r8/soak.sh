#!/bin/bash
# Round 8 soak on the current binaries: kernels against the oracle on random draws (new expf on both sides, loop diet, queue walk in
# bin_segments), the compiled node against the Python Function, the batched entry points; then the same parity draws with
# deterministic_grads on.
cd "$(dirname "$0")/../.."
O=gpurun_out/r8_soak; mkdir -p $O
( timeout 400 python tests/tools/soak_node.py --seconds 100 --seed 41
  timeout 400 python tests/tools/soak_node.py --seconds 100 --seed 42 --drop-inputs
  timeout 900 python tests/tools/soak_parity.py 300 120 41
  timeout 900 python tests/tools/soak_parity.py 250 100 42
  DGR_DETERMINISTIC_GRADS=1 timeout 900 python tests/tools/soak_parity.py 250 0 43
  timeout 700 python tests/tools/soak_batch.py ) 2>&1 | grep -v amdgpu.ids | grep "soak_node\|FAIL\|AMBIG\|draws in\|draws,\|MISMATCH\|Error\|error" | cut -c1-400 | tee $O/soak.txt
