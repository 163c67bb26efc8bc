#!/bin/bash
cd "$(dirname "$0")/../.."
L=$PWD/diff-gaussian-rasterization_amd/lib
for i in 1 2 3; do
for v in libdgr_hip.so libdgr_hip_tag1.so; do
  DGR_HIP_LIB=$L/$v python bench.py --no-cpu-baseline --views-in-flight 1 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['config']['stage_ms'].items() if k.startswith('render')})"
done; done
DGR_HIP_LIB=$L/libdgr_hip_tag1.so DGR_BINDING=ctypes python -m pytest tests/test_hip_light_parity.py -q -m gpu -k "forward_images or contribution_tags" 2>&1 | tail -2
