cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7_soak
for seed in 7 8; do timeout 900 python tests/tools/soak_parity.py 250 100 $seed 2>&1 | grep -v amdgpu.ids | tail -6; done | tee gpurun_out/r7_soak/soak_parity.txt
timeout 900 python tests/tools/soak_batch.py 2>&1 | grep -v amdgpu.ids | tail -5 | tee gpurun_out/r7_soak/soak_batch.txt
