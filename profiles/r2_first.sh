#!/bin/bash
# Round 2, first GPU call: the whole GPU suite with the reworked parity tests, the default bench line (as the driver runs
# it), the same through the self-spawner at N=1, and the N=2 code path on one GPU (two ranks, gloo) via plain
# `python bench.py --gpus 2`.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r2_first
timeout 3000 python -m pytest tests -q -m gpu -x -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r2_first/pytest.log
tail -5 gpurun_out/r2_first/pytest.log
python bench.py --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r2_first/bench_default.json
DGR_BENCH_SPAWN=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r2_first/bench_spawn_n1.json
DGR_BENCH_SHARE_GPU=1 DGR_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 2 --steps 12 --warmup 3 --workload config2 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -3 > gpurun_out/r2_first/bench_two_ranks_one_gpu.log
for f in bench_default.json bench_spawn_n1.json; do python - "$f" <<'PY'
import json,sys
d=json.load(open("gpurun_out/r2_first/"+sys.argv[1]))
print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "serial", d["config"]["ms_per_view_one_stream"], "roofline", {k: d["roofline"][k] for k in ("kernel","frac","avg_ms","launches","frac_under_overlap","launches_under_overlap")}, d.get("cpu_baseline",{}).get("sample"), d["config"].get("grad_max_abs_err",{}).get("max"))
print({k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items()})
PY
done
tail -c 600 gpurun_out/r2_first/bench_two_ranks_one_gpu.log
