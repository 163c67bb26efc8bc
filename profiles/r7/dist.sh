#!/bin/bash
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_dist; mkdir -p $O
timeout 600 python profiles/r7/collective_under_blend.py 2>&1 | grep -v "amdgpu.ids\|NCCL\|RCCL version\|Warning" > $O/collective_under_blend.txt
bash profiles/dist_paths_check.sh > $O/dist_paths_check.txt 2>&1
DGR_BENCH_FORCE_DIST=1 MASTER_PORT=29512 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $O/one_rank_rccl_line.json
DGR_BENCH_SHARE_GPU=1 DGR_BENCH_BACKEND=gloo timeout 600 python bench.py --gpus 2 --steps 12 --warmup 3 --workload config2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/two_rank_gloo_line.json
cat $O/collective_under_blend.txt $O/dist_paths_check.txt
python - <<'PY'
import json
for f in ("one_rank_rccl_line","two_rank_gloo_line"):
    try:
        d=json.load(open(f"gpurun_out/r7_dist/{f}.json")); print(f, d["ms_per_step"], d["config"]["allreduce_model"])
    except Exception as e: print(f, "ERR", e)
PY
