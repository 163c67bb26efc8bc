#!/bin/bash
cd "$(dirname "$0")/.."
timeout 1200 python -m pytest tests/test_hip_lazy_safety.py tests/test_hip_thread_options.py tests/test_hip_edge_cases.py tests/test_hip_deterministic.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -15
