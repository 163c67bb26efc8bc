#!/bin/bash
# Round 9 (driver round 6), first call: the Appendix C test on the kernels as they stood at the start of the round, and
# same-box baselines for the three scenes / variants this round works on.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
timeout 900 python -m pytest tests/test_hip_appendix_c.py -x -q -m gpu -s 2>&1 | grep -v amdgpu.ids | tail -15 | tee gpurun_out/r9/appendix_c.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print(d["config"]["workload"][:40], "ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
for args in "" "--scene clustered" "--scene heavy_tail" "--workload config2 --variant full" "--workload config2" "--tracking"; do
  echo "== bench $args"
  python bench.py --no-cpu-baseline --steps 100 $args 2>/dev/null | tail -1 | tee -a gpurun_out/r9/baseline_lines.json | python -c "$P"
done 2>&1 | tee gpurun_out/r9/baseline.txt
