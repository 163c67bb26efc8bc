#!/bin/bash
# Round 7: suite at the tightened bars + the metric's second half for every BASELINE configuration.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_parity; mkdir -p $O
timeout 2400 python -m pytest tests -q -s -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | grep -v "UserWarning\|return Variable\|assert abs\|Docs:\|Consider using" | tee $O/pytest_full.log | tail -30 > $O/pytest.log; grep -h "dL_dview," $O/pytest_full.log | sort -u > $O/dview_errors.txt
python bench.py --workload config2 --variant full 2>/dev/null | tail -1 > $O/bench_config2_full.json
python bench.py --workload config4 2>/dev/null | tail -1 > $O/bench_config4_light_view.json
python bench.py --workload config5 --steps 50 --warmup 5 2>/dev/null | tail -1 > $O/bench_config5_light_view.json
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config3_light_driver_cmd.json
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    e=c.get("grad_max_abs_err",{})
    print(sys.argv[1].split('/')[-1], "ms/step", round(d["ms_per_step"],4), "one", c.get("ms_per_view_one_stream") and round(c["ms_per_view_one_stream"],4), "graph", c.get("ms_per_step_hipgraph_replay"), "err max", e.get("max"), {k:("%.1e"%v) for k,v in e.items() if k.startswith("dL")}, "cpu", d.get("cpu_baseline",{}).get("sample","")[:90])
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt
cat $O/pytest.log $O/summary.txt
