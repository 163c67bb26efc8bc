#!/bin/bash
# Views in flight x hardware queues, after the host path got 2.5x cheaper (round 7).
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_sweep; mkdir -p $O
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4))'
for q in default 8; do
  for k in 3 5 7 9 11; do
    if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
    a=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 200 2>/dev/null | python -c "$P")
    b=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 20 --warmup 5 2>/dev/null | python -c "$P")
    c=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 20 --warmup 5 2>/dev/null | python -c "$P")
    echo "hw_queues=$q views=$k  200 steps: $a   20 steps: $b $c"
  done
done | tee $O/sweep.txt
