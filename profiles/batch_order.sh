#!/bin/bash
# a batch's per-view stages: pipelined (binning stream + blend stream, option batch_order 1) against round robin over 2 streams
cd "$(dirname "$0")/.."
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print("ms/step", round(d["ms_per_step"],4), "ms/view", round(c["ms_per_view"],4))'
for W in config3 config2; do for V in 2 4 8; do for O in 1 0; do
  echo "$W --batch $V --batch-order $O"; python bench.py --no-cpu-baseline --workload $W --batch $V --batch-order $O --steps 60 2>/dev/null | tail -1 | python -c "$P"
done; done; done
