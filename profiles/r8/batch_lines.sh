#!/bin/bash
# BASELINE configs 4 and 5 as they are written for one GPU's share: batches of 4 camera views through the batched entry points,
# eager and replayed from a hipGraph (config 5: "32-view batch over 8 GPUs ... hipGraph-captured render loop" = 4 views per GPU
# per step), with the error figure and the CPU baseline; config 3 batches of 4 and 8 refreshed on the current kernels.
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r8
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; e=c.get("grad_max_abs_err") or {}; print(sys.argv[1], "ms/step", round(d["ms_per_step"],4), "ms/view", round(c["ms_per_view"],4), "graph", c["hipgraph_replay"], "err max", e.get("max"), "dL_dview", e.get("dL_dview"), "cpu", (d.get("cpu_baseline") or {}).get("sample", "")[:90])'
run() { name=$1; shift; python bench.py "$@" 2>gpurun_out/r8/$name.err | tail -1 > gpurun_out/r8/$name.json; python -c "$P" $name < gpurun_out/r8/$name.json || tail -5 gpurun_out/r8/$name.err; }
run r8_bench_config3_batch4 --steps 50 --warmup 5 --batch 4 --cpu-runs 2
run r8_bench_config3_batch8 --steps 30 --warmup 5 --batch 8 --no-cpu-baseline
run r8_bench_config3_batch4_graph --steps 50 --warmup 5 --batch 4 --graph --no-cpu-baseline
run r8_bench_config4_batch4 --workload config4 --steps 20 --warmup 3 --batch 4
run r8_bench_config4_batch4_graph --workload config4 --steps 20 --warmup 3 --batch 4 --graph
run r8_bench_config5_batch4 --workload config5 --steps 10 --warmup 2 --batch 4
run r8_bench_config5_batch4_graph --workload config5 --steps 10 --warmup 2 --batch 4 --graph
