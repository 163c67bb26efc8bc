#!/bin/bash
# The GPU suite under the environment switches that select other code paths (final binaries of round 9): one line per setting.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r9
run() { echo -n "$* : "; env "$@" timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -E "passed|failed|error" | tail -1; }
( run DGR_FWD_HALVES=0
  run DGR_FWD_HALVES=1
  run DGR_LDS_COUNT=0
  run DGR_SEG_SHIFT=2
  run DGR_SEG_SHIFT=4
  run DGR_TILE_SCHEDULE=0
  run DGR_TILE_SCHEDULE=1
  run DGR_BINDING=ctypes
  run DGR_FORWARD_MODE=callback ) 2>&1 | tee gpurun_out/r9/matrix.txt
