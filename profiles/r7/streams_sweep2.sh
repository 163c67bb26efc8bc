#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_sweep; mkdir -p $O
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4))'
for k in 7 11 15 21; do
    a=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 200 2>/dev/null | python -c "$P")
    b=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 20 --warmup 5 2>/dev/null | python -c "$P")
    c=$(python bench.py --no-cpu-baseline --views-in-flight $k --steps 20 --warmup 5 2>/dev/null | python -c "$P")
    echo "views=$k  200 steps: $a   20 steps: $b $c"
done | tee $O/sweep2.txt
DGR_BENCH_TRACE=1 python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>&1 | grep trace | tee -a $O/sweep2.txt
DGR_BENCH_TRACE=1 python bench.py --no-cpu-baseline --steps 40 --warmup 5 2>&1 | grep trace | tee -a $O/sweep2.txt
