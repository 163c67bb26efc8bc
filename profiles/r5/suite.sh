#!/bin/bash
# whole GPU suite (no -x) + the default bench, the driver's command and the one-stream figure
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_suite; mkdir -p $O
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/pytest.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()}, "sum", round(sum(d["config"]["stage_ms"].values())*1e3,1))'
python bench.py --no-cpu-baseline --steps 200 2>/dev/null | tail -1 | tee $O/bench_default.json | python -c "$P" > $O/bench_default.txt
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_driver_cmd.json
python bench.py --no-cpu-baseline --workload config4 --steps 30 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config4.txt
python bench.py --no-cpu-baseline --workload config5 --steps 20 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config5.txt
cat $O/pytest.txt $O/bench_*.txt; python -c "
import json; d=json.load(open('$O/bench_driver_cmd.json')); print('driver cmd', d['ms_per_step'], d['value'], d['config'].get('grad_max_abs_err'), d.get('roofline'))"
