#!/bin/bash
# Round 6: how far may the host run ahead of the status words (lazy mode)?  DGR_LAZY_DEPTH = status words left unread when a
# forward is issued (1 = the shipped value: view i is issued once view i-2's forward has reported)
cd "$(dirname "$0")/../.."
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step'],4))"; }
for rep in 1 2; do
for D in 1 2 3 5; do
  for K in 20 100; do
    DGR_LAZY_DEPTH=$D python bench.py --steps $K --warmup 5 --no-cpu-baseline 2>/dev/null | line "depth=$D steps=$K"
  done
done
done
