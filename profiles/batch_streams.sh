#!/bin/bash
# --batch V with 1..8 internal streams, same box
cd "$(dirname "$0")/.."
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print("ms/step", round(d["ms_per_step"],4), "ms/view", round(c["ms_per_view"],4))'
W=${1:-config3}
for V in 3 4 8; do for K in 1 2 3 4 8; do
  echo "--batch $V --batch-streams $K"; python bench.py --no-cpu-baseline --workload $W --batch $V --batch-streams $K --steps 60 2>/dev/null | tail -1 | python -c "$P"
done; done
