#!/bin/bash
# float64 arbitration of the two draws of profiles/r9/soak.txt beyond the soak's bar (tests/tools/arbitrate_fp64.py, which re-creates
# draws with precomputed colours / covariances since round 9)
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r9
( timeout 900 python tests/tools/arbitrate_fp64.py 81 246
  DGR_SOAK_HEAVY=1 DGR_DETERMINISTIC_GRADS=1 timeout 900 python tests/tools/arbitrate_fp64.py 85 132 ) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r9/soak_arbitrate.txt
