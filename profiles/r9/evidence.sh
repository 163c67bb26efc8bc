#!/bin/bash
# Round 9 (driver round 6) evidence at HEAD: the GPU suite, smoke, rocprofv3 stats + PMC, every bench line quoted in DESIGN.md.
cd "$(dirname "$0")/../.."
O=gpurun_out/r9_evidence; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" | grep -v "UserWarning\|return Variable\|assert abs\|Docs:\|Consider using" | tail -12 > $O/pytest.log
timeout 600 python -m pytest tests/test_hip_appendix_c.py -q -m gpu -s 2>&1 | grep "^light\|^full\|passed\|failed" > $O/appendix_c.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -2 > $O/smoke.log
bash profiles/final_round.sh r9 > $O/final_round.log 2>&1
cp gpurun_out/prof_r9/r9_* $O/ 2>/dev/null
# the traffic file the bench lines below read, from THIS run's counter passes (the snapshot's file may predate a kernel change)
cp gpurun_out/prof_r9/r9_pmc.txt profiles/r9_pmc.txt && python profiles/make_pmc_traffic.py profiles/r9_pmc.txt ${DGR_EVIDENCE_COMMIT:-worktree} > /dev/null && cp profiles/pmc_traffic.json $O/
# the clustered frame: counters and a trace of bin_tiles (what bounds it now)
bash profiles/run_profile.sh r9_clustered --scene clustered > /dev/null 2>&1; python profiles/summarize.py gpurun_out/prof_r9_clustered gpurun_out/prof_r9_clustered/r9_clustered > /dev/null 2>&1
cp gpurun_out/prof_r9_clustered/r9_clustered_kernel_stats.txt gpurun_out/prof_r9_clustered/r9_clustered_pmc.txt $O/ 2>/dev/null
python profiles/r9/bin_tiles_trace.py clustered 2>&1 | grep -v amdgpu.ids > $O/bin_tiles_trace_after.txt
(cd profiles/microbench && ./wave_sort 0 && ./wave_sort 37) > $O/wave_sort.txt 2>&1
B="python bench.py"
for i in 1 2 3; do $B --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_config3_light_driver_cmd_$i.json; done
$B 2>/dev/null | tail -1 > $O/bench_config3_light.json
$B --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_one_stream.json
$B --sync-mode strict --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_strict.json
$B --tracking --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_tracking.json
$B --tracking --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_tracking_one_stream.json
$B --tight-cull --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_tight_cull.json
$B --lean-loss --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_lean_loss.json
$B --scene clustered --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_clustered.json
$B --scene heavy_tail --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_heavy_tail.json
DGR_DETERMINISTIC_GRADS=1 $B --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_light_deterministic.json
$B --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_batch4.json
DGR_DETERMINISTIC_GRADS=1 $B --batch 4 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_batch4_deterministic.json
$B --workload config2 --variant full 2>/dev/null | tail -1 > $O/bench_config2_full.json
$B --workload config2 --variant full --views-in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2_full_one_stream.json
DGR_DETERMINISTIC_GRADS=1 $B --workload config2 --variant full --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2_full_deterministic.json
$B --workload config2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2_light.json
$B --workload config4 --cpu-runs 1 2>/dev/null | tail -1 > $O/bench_config4_light_view.json
$B --workload config5 --steps 50 --warmup 5 --cpu-runs 1 2>/dev/null | tail -1 > $O/bench_config5_light_view.json
python examples/tracking.py --fused 2>&1 | grep -v amdgpu.ids | tail -4 > $O/tracking_example_eager.txt
python examples/tracking.py --fused --graph 2>&1 | grep -v amdgpu.ids | tail -4 > $O/tracking_example_graph.txt
python examples/mapping.py 2>&1 | grep -v amdgpu.ids | tail -6 > $O/mapping_example.txt
for f in $O/bench_*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1])); c=d["config"]; r=d["roofline"]
    print(sys.argv[1].split('/')[-1][6:-5], "ms/step", round(d["ms_per_step"],4), "one", c.get("ms_per_view_one_stream") and round(c["ms_per_view_one_stream"],4), "strict-one", c.get("ms_per_view_strict_one_stream") and round(c["ms_per_view_strict_one_stream"],4), "K", c["views_in_flight"], "frac", round(r["frac"],4), r["kernel"], {k:round(v*1e3,1) for k,v in c["stage_ms"].items()}, "graph", c.get("ms_per_step_hipgraph_replay"), "err", (c.get("grad_max_abs_err") or {}).get("max"), "cpu", (d.get("cpu_baseline") or {}).get("value"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
done > $O/summary.txt
find gpurun_out -name "*.db" -size +1M -delete   # (the raw profiler databases: gpurun merges at most 64 MiB back)
cat $O/pytest.log $O/smoke.log $O/appendix_c.txt $O/summary.txt $O/tracking_example_*.txt $O/mapping_example.txt
