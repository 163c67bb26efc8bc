#!/bin/bash
# batched entry points against the one-view surface, same box: full GPU suite, default line, V views per step as a batch
# (--batch V) against the same views one call at a time with .grad accumulating (--group V, one stream) and against
# three independent views in flight (the default line, no accumulation)
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print("ms/step", round(d["ms_per_step"],4), "ms/view", round(c["ms_per_view"],4), "one-stream", c["ms_per_view_one_stream"] and round(c["ms_per_view_one_stream"],4), {k: round(v*1e3,1) for k,v in c["stage_ms"].items()})'
W=${1:-config3}
python bench.py --no-cpu-baseline --workload $W 2>/dev/null | tail -1 | python -c "$P"
for V in 2 4 8; do
  echo "--batch $V"; python bench.py --no-cpu-baseline --workload $W --batch $V --steps 60 2>/dev/null | tail -1 | python -c "$P"
  echo "--group $V"; python bench.py --no-cpu-baseline --workload $W --group $V --steps 240 2>/dev/null | tail -1 | python -c "$P"
done
