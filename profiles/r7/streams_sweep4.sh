#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_sweep; mkdir -p $O
P='import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d["ms_per_step"],4))'
for i in 1 2; do
for cfg in "7 1" "7 32" "21 1" "11 32"; do
    set -- $cfg
    b=$(DGR_LAZY_DEPTH=$2 DGR_BENCH_TRACE=1 python bench.py --no-cpu-baseline --views-in-flight $1 --steps 20 --warmup 5 2>$O/trace.tmp | python -c "$P")
    echo "views=$1 lazy_depth_floor=$2  20 steps: $b   $(grep trace $O/trace.tmp | cut -c1-120)"
done; done | tee $O/sweep4.txt
