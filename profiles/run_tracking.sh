#!/bin/bash
# tracking example: eager / hipGraph, torch ops / fused pose + loss + Adam
cd "$(dirname "$0")/.."
timeout 600 python -m pytest tests/test_slam_render.py tests/test_hip_optim.py -x -q -m gpu 2>&1 | tail -3
for f in "" "--fused" "--graph" "--graph --fused"; do timeout 300 python examples/tracking.py $f 2>&1 | tail -2; done
