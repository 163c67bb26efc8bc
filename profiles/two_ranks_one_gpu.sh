#!/bin/bash
# Exercises bench.py's N>1 logic (rank-0 printing, barriers, grouped gradient reduce across processes) with two ranks that
# share the box's single GPU over gloo.  Throughput is meaningless here; the point is that it runs and prints one line.
DGR_BENCH_SHARE_GPU=1 DGR_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 12 --warmup 3 --workload config2 --no-cpu-baseline 2>&1 | grep -v -i "warn\|amdgpu.ids" | tail -3 | cut -c1-700
