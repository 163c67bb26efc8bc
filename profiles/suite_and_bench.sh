#!/bin/bash
# full GPU suite, then per-stage times (one view at a time) and the default bench twice
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
python bench.py --no-cpu-baseline --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P"
for i in 1 2; do python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P"; done
python bench.py --no-cpu-baseline --variant full --workload config2 --graph 2>/dev/null | tail -1 | python -c "$P"
