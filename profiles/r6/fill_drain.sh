#!/bin/bash
# Round 6: where does the intercept of T(K) = slope * K + intercept come from at the driver's --steps 20?
cd "$(dirname "$0")/../.."
O=gpurun_out/r6_fill; mkdir -p $O
line() { python -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('$1', 'ms_per_step', round(d['ms_per_step'],4), 'one_stream', d['config'].get('ms_per_view_one_stream'))"; }
for K in 5 10 20 40 80; do
  DGR_BENCH_TRACE=1 python bench.py --steps $K --warmup 5 --no-cpu-baseline 2>$O/trace_$K.err | line "steps=$K"
  grep "\[trace\]" $O/trace_$K.err
done
for PE in 2 1000; do
  DGR_BENCH_PROFILE_EVERY=$PE DGR_BENCH_TRACE=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>$O/trace_pe$PE.err | line "steps=20 profile_every=$PE"
  grep "\[trace\]" $O/trace_pe$PE.err
done
for V in 2 4 5; do
  DGR_BENCH_PROFILE_EVERY=1000 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --views-in-flight $V 2>/dev/null | line "steps=20 pe=1000 views_in_flight=$V"
done
