cd $GRAFT_REPO_ROOT
for a in "" "--fused" "--graph" "--fused --graph"; do python examples/mapping.py $a 2>&1 | grep -v amdgpu.ids | tail -1; done
for a in "--fused" "--fused --graph"; do python examples/tracking.py $a 2>&1 | grep -v amdgpu.ids | tail -1; done
python -m pytest tests/test_slam_render.py tests/test_hip_optim.py -q -m gpu 2>&1 | tail -1
