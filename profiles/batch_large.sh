#!/bin/bash
# batched entry points at the sizes of BASELINE configs 4 and 5 (per-GPU views): 4 views per step as a batch against the same
# views one call at a time with .grad accumulating, and against the default line (three independent views in flight)
cd "$(dirname "$0")/.."
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print("ms/step", round(d["ms_per_step"],4), "ms/view", round(c["ms_per_view"],4), "one-stream", c["ms_per_view_one_stream"] and round(c["ms_per_view_one_stream"],4), {k: round(v*1e3,1) for k,v in c["stage_ms"].items()})'
for W in config4 config5; do
  echo "== $W default"; python bench.py --no-cpu-baseline --workload $W --steps 40 --warmup 6 2>/dev/null | tail -1 | python -c "$P"
  echo "== $W --batch 4"; python bench.py --no-cpu-baseline --workload $W --batch 4 --steps 12 --warmup 4 2>/dev/null | tail -1 | python -c "$P"
  echo "== $W --group 4"; python bench.py --no-cpu-baseline --workload $W --group 4 --steps 40 --warmup 8 2>/dev/null | tail -1 | python -c "$P"
done
