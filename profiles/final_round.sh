#!/bin/bash
# End-of-round evidence: full GPU test suite, rocprofv3 passes (run_profile.sh), default bench line.
cd "$(dirname "$0")/.."
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -3 | tee gpurun_out/pytest_gpu_tail.txt
bash profiles/run_profile.sh v14 > /dev/null 2>&1
python profiles/summarize.py gpurun_out/prof_v14 gpurun_out/prof_v14/r1_v14_final
python bench.py > gpurun_out/bench_default.json 2>/dev/null; tail -1 gpurun_out/bench_default.json | cut -c1-200
