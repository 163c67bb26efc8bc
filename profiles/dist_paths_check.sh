#!/bin/bash
# Every N>1 code path of bench.py on a one-GPU box: (a) plain `python bench.py --gpus 2` self-spawn, two ranks sharing the GPU
# over gloo (DGR_BENCH_SHARE_GPU / DGR_BENCH_BACKEND test hooks), default exchange pattern; (b) the same with
# --views-per-allreduce 4; (c) with --batch 4 (one all-reduce per batched step); (d) a one-rank RCCL group (real RCCL init,
# blocking and overlapped all-reduce).  Throughput is meaningless here; the point is that each runs and prints its line.
cd "$(dirname "$0")/.."
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); c=d["config"]; print(d["n_gpus"], "ranks", round(d["ms_per_step"],4), "ms/step", "rccl_ranks", c["rccl_ranks"], c["gradient_allreduce"], "| views/step", c["views_per_step"])'
export DGR_BENCH_SHARE_GPU=1 DGR_BENCH_BACKEND=gloo
for extra in "" "--views-per-allreduce 4" "--batch 4"; do
  echo "== --gpus 2 $extra (gloo, shared GPU)"
  timeout 600 python bench.py --gpus 2 --steps 12 --warmup 3 --workload config2 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "$P" || echo FAILED
done
unset DGR_BENCH_SHARE_GPU DGR_BENCH_BACKEND
for extra in "--allreduce blocking" "--allreduce overlap" "--batch 4"; do
  echo "== one-rank RCCL group $extra"
  DGR_BENCH_FORCE_DIST=1 MASTER_PORT=29511 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline $extra 2>/dev/null | tail -1 | python -c "$P" || echo FAILED
done
