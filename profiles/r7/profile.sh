#!/bin/bash
# Round 7 evidence: exact-math / front-end tests, rocprofv3 kernel stats + PMC passes, the bench lines.
cd "$(dirname "$0")/../.."
O=gpurun_out/r7_prof; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_exact_math.py tests/test_hip_front_end.py tests/test_hip_light_parity.py tests/test_hip_random_sweep.py tests/test_hip_full_parity.py -q -m gpu 2>&1 | tail -4 > $O/pytest.log
bash profiles/final_round.sh r7 > $O/final_round.log 2>&1
cp gpurun_out/prof_r7/r7_* $O/ 2>/dev/null
cat $O/pytest.log; tail -20 $O/final_round.log
