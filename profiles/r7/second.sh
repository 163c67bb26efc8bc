#!/bin/bash
# Round 7, second GPU call: suite, host breakdown, config 2 / config 3 lines after the front-end changes.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_fifth; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_front_end.py tests/test_hip_light_parity.py -q -m gpu 2>&1 | tail -3 > $O/pytest.log; python profiles/r7/pyprof.py 2>&1 | grep -v amdgpu.ids | tail -32 > $O/pyprof.txt
for v in light full; do
  DGR_HOST_PROF=1 python profiles/host_breakdown.py $v 2>&1 | grep -v amdgpu.ids > $O/host_${v}_node.txt
done
B="python bench.py --no-cpu-baseline --workload config2 --variant full"
$B --views-in-flight 1 2>/dev/null | tail -1 > $O/c2_full_one.json
$B 2>/dev/null | tail -1 > $O/c2_full_seven.json
$B --views-in-flight 3 2>/dev/null | tail -1 > $O/c2_full_three.json
DGR_BENCH_AUTOGRAD_THREADS=1 $B 2>/dev/null | tail -1 > $O/c2_full_seven_threads.json
$B --graph 2>/dev/null | tail -1 > $O/c2_full_graph.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c3_driver.json
python bench.py --no-cpu-baseline --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/c3_driver_b.json
python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/c3_200.json
python bench.py --no-cpu-baseline --views-in-flight 1 2>/dev/null | tail -1 > $O/c3_one.json
DGR_TILE_SCHEDULE=1 python bench.py --no-cpu-baseline --views-in-flight 1 2>/dev/null | tail -1 > $O/c3_one_sched_always.json
DGR_TILE_SCHEDULE=1 python bench.py --no-cpu-baseline 2>/dev/null | tail -1 > $O/c3_200_sched_always.json
python bench.py --no-cpu-baseline --scene clustered --views-in-flight 1 2>/dev/null | tail -1 > $O/c3_clustered_one.json
for f in $O/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]
    print(sys.argv[1].split('/')[-1], "ms/step", round(d["ms_per_step"],4), "one_stream", c.get("ms_per_view_one_stream") and round(c["ms_per_view_one_stream"],4), "K", c["views_in_flight"], {k:round(v*1e3,1) for k,v in c["stage_ms"].items()})
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt
cat $O/pytest.log $O/pyprof.txt $O/host_*.txt $O/summary.txt
