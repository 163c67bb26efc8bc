cd /root/repo 2>/dev/null || cd $GRAFT_REPO_ROOT
O=gpurun_out/r7_parity; mkdir -p $O
timeout 2400 python -m pytest tests/test_hip_random_sweep.py tests/test_hip_light_parity.py -q -s -m gpu 2>&1 | grep -v "^$\|amdgpu.ids" > $O/pytest_two.log; tail -5 $O/pytest_two.log; grep -h "dL_dview," $O/pytest_two.log | sort -u | grep -E "P=(2000000|5000000|500000) "
