#!/bin/bash
# dist paths on the one-GPU box, config 2 after the small-frame segment rule, and the whole GPU suite
cd "$(dirname "$0")/../.."
O=gpurun_out/r5_final; mkdir -p $O
bash profiles/dist_paths_check.sh > $O/dist_paths_check.txt 2>&1
DGR_BENCH_FORCE_DIST=1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --workload config4 --blend-wgs-per-cu 7 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print("one-rank RCCL, config4, blend_wgs_per_cu 7:", round(d["ms_per_step"],4), "ms/step;", d["config"]["workload"][:120])' >> $O/dist_paths_check.txt 2>&1
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})'
python bench.py --steps 100 --variant full --workload config2 --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_config2_full.json | python -c "$P" > $O/bench_config2_full.txt
python bench.py --steps 100 --variant full --workload config2 --graph --no-cpu-baseline 2>/dev/null | tail -1 | tee $O/bench_config2_full_graph.json | python -c "$P" > $O/bench_config2_full_graph.txt
python bench.py --steps 100 --workload config2 --tracking --no-cpu-baseline 2>/dev/null | tail -1 | python -c "$P" > $O/bench_config2_light_tracking.txt
timeout 3000 python -m pytest tests -q -m gpu 2>&1 | grep -v "amdgpu.ids" | tail -6 > $O/pytest.txt
cat $O/dist_paths_check.txt $O/bench_*.txt $O/pytest.txt
