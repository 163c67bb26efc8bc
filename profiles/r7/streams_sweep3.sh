#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_sweep; mkdir -p $O
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4), d["config"]["ms_per_view_one_stream"] and round(d["config"]["ms_per_view_one_stream"],4))'
for k in 7 21 32 48; do
    a=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 200 2>/dev/null | python -c "$P")
    b=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 20 --warmup 5 2>/dev/null | python -c "$P")
    echo "views=$k  200 steps: $a   20 steps: $b"
done | tee $O/sweep3.txt
for k in 7 21; do
  a=$(python bench.py --no-cpu-baseline --views-in-flight $k --workload config2 --variant full 2>/dev/null | python -c "$P")
  echo "config2 full views=$k: $a"
done | tee -a $O/sweep3.txt
