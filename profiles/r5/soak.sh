#!/bin/bash
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r5_soak
for seed in 7 11; do
  timeout 2400 python tests/tools/soak_parity.py 250 100 $seed 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r5_soak/soak_seed$seed.txt
done
DGR_LDS_COUNT=0 timeout 1200 python tests/tools/soak_parity.py 100 30 13 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r5_soak/soak_global_counters_seed13.txt
timeout 1500 python tests/tools/soak_batch.py 200 17 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r5_soak/soak_batch_seed17.txt
tail -4 gpurun_out/r5_soak/*.txt
