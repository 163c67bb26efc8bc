cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r7_soak
for seed in 2 3 4 5 6; do timeout 400 python tests/tools/soak_node.py --seconds 60 --seed $seed 2>&1 | grep -v amdgpu.ids | cut -c1-300 | tail -3; done | tee gpurun_out/r7_soak/soak_node.txt
