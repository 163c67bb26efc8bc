cd $GRAFT_REPO_ROOT
DGR_SYNC_MODE=lazy python profiles/r7/tracking_profile.py 2>&1 | grep -v amdgpu.ids | head -30
python examples/tracking.py --fused 2>&1 | grep -v amdgpu.ids | tail -2
DGR_SYNC_MODE=lazy python examples/tracking.py --fused 2>&1 | grep -v amdgpu.ids | tail -1
python examples/tracking.py --fused --graph 2>&1 | grep -v amdgpu.ids | tail -1
python examples/tracking.py 2>&1 | grep -v amdgpu.ids | tail -1
python -m pytest tests/test_slam_render.py -q -m gpu 2>&1 | tail -2
