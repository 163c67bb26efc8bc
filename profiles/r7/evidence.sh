#!/bin/bash
# Round 7 (driver round 4) evidence at HEAD: the GPU suite, smoke, rocprofv3 stats + PMC, every bench line quoted in DESIGN.md.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_evidence; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | grep -v "UserWarning\|return Variable\|assert abs\|Docs:\|Consider using" | tail -12 > $O/pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 > $O/smoke.log
bash profiles/final_round.sh r7 > $O/final_round.log 2>&1
cp gpurun_out/prof_r7/r7_* $O/ 2>/dev/null
B="python bench.py"
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config3_light_driver_cmd.json
$B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config3_light_driver_cmd_2.json
$B 2>/dev/null | tail -1 > $O/bench_config3_light.json
$B --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_one_stream.json
$B --sync-mode strict --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_strict.json
$B --tracking --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_tracking.json
$B --tight-cull --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_tight_cull.json
$B --scene clustered --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_clustered.json
$B --scene clustered --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_clustered_one_stream.json
$B --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_batch4.json
$B --workload config2 --variant full 2>/dev/null | tail -1 > $O/bench_config2_full.json
$B --workload config2 --variant full --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2_full_one_stream.json
$B --workload config2 --variant full --graph --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2_full_graph.json
$B --workload config4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4_light_view.json
$B --workload config5 --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config5_light_view.json
python examples/tracking.py --fused 2>&1 | grep -v amdgpu.ids | tail -4 > $O/tracking_example_eager.txt
python examples/tracking.py --fused --graph 2>&1 | grep -v amdgpu.ids | tail -4 > $O/tracking_example_graph.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split('/')[-1][6:-5], "ms/step", round(d["ms_per_step"],4), "one", c.get("ms_per_view_one_stream") and round(c["ms_per_view_one_stream"],4), "K", c["views_in_flight"], "frac", round(r["frac"],4), r["kernel"], {k:round(v*1e3,1) for k,v in c["stage_ms"].items()}, "err", c.get("grad_max_abs_err",{}).get("max"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt
cat $O/pytest.log $O/smoke.log $O/summary.txt $O/tracking_example_*.txt
