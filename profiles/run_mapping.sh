#!/bin/bash
# GPU check of the optimiser additions and the mapping example
cd "$(dirname "$0")/.."
timeout 900 python -m pytest tests/test_hip_optim.py tests/test_slam_render.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python examples/mapping.py 2>&1 | tail -1
timeout 600 python examples/mapping.py --views-in-flight 1 2>&1 | tail -1
timeout 300 python examples/mapping.py --graph 2>&1 | tail -3
timeout 600 python examples/mapping.py --fused 2>&1 | tail -1
timeout 300 python examples/mapping.py --fused --graph 2>&1 | tail -1
