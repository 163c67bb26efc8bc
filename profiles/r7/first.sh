#!/bin/bash
# Round 7 (driver round 4), first GPU call: the suite on the compiled autograd node, host breakdown A/B, config 2 lines.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_first; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu -x 2>&1 | grep -v "^$\|amdgpu.ids" | tail -15 > $O/pytest.log
for v in light full; do
  python profiles/host_breakdown.py $v 2>&1 | grep -v amdgpu.ids > $O/host_${v}_node.txt
  DGR_AUTOGRAD=python python profiles/host_breakdown.py $v 2>&1 | grep -v amdgpu.ids | head -3 > $O/host_${v}_pyfunc.txt
done
B="python bench.py --no-cpu-baseline --workload config2 --variant full"
$B --views-in-flight 1 2>/dev/null | tail -1 > $O/c2_full_one.json
$B 2>/dev/null | tail -1 > $O/c2_full_seven.json
$B --views-in-flight 3 2>/dev/null | tail -1 > $O/c2_full_three.json
DGR_AUTOGRAD=python $B --views-in-flight 1 2>/dev/null | tail -1 > $O/c2_full_one_pyfunc.json
$B --graph 2>/dev/null | tail -1 > $O/c2_full_graph.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c3_driver.json
python bench.py --no-cpu-baseline --views-in-flight 1 2>/dev/null | tail -1 > $O/c3_one.json
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(sys.argv[1].split('/')[-1], "ms/step", round(d["ms_per_step"],4), "one_stream", c.get("ms_per_view_one_stream"), "K", c["views_in_flight"], {k:v for k,v in c["stage_ms"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt
cat $O/pytest.log $O/host_*.txt $O/summary.txt
