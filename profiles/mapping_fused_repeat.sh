#!/bin/bash
# examples/mapping.py --fused, eager, both bindings, three runs each on one box (the eager figure is host-bound)
cd "$(dirname "$0")/.."
for i in 1 2 3; do
  timeout 300 python examples/mapping.py --fused 2>&1 | tail -1 | cut -c1-110
  DGR_BINDING=ctypes timeout 300 python examples/mapping.py --fused 2>&1 | tail -1 | cut -c1-110
done
timeout 300 python examples/mapping.py --views-in-flight 1 2>&1 | tail -1 | cut -c1-110
timeout 300 python examples/mapping.py --fused --graph 2>&1 | tail -1 | cut -c1-110
