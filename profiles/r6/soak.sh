#!/bin/bash
# Round 6 soak: random draws against the oracle, 40 % of them clustered (tests/tools/soak_parity.py), both binning paths, batches
cd "$(dirname "$0")/../.."
mkdir -p gpurun_out/r6_soak
for seed in 21 22; do
  timeout 2400 python tests/tools/soak_parity.py 250 100 $seed 2>&1 | grep -v amdgpu.ids | tail -12 > gpurun_out/r6_soak/soak_seed$seed.txt
done
DGR_LDS_COUNT=0 timeout 1200 python tests/tools/soak_parity.py 100 30 23 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r6_soak/soak_global_counters_seed23.txt
timeout 1500 python tests/tools/soak_batch.py 200 27 2>&1 | grep -v amdgpu.ids | tail -6 > gpurun_out/r6_soak/soak_batch_seed27.txt
for f in gpurun_out/r6_soak/*.txt; do echo "== $f"; tail -n 5 $f; done
