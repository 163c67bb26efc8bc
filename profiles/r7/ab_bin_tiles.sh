#!/bin/bash
cd "$(dirname "$0")/../.."
L=$PWD/diff-gaussian-rasterization_amd/lib
for i in 1 2; do
for v in libdgr_hip.so libdgr_hip_nolongest.so; do
  DGR_HIP_LIB=$L/$v python bench.py --no-cpu-baseline --views-in-flight 1 --steps 100 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('$v', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['config']['stage_ms'].items()})"
done; done
python bench.py --no-cpu-baseline --views-in-flight 1 --steps 100 --sync-mode strict 2>/dev/null | tail -1 | python -c "import json,sys; d=json.load(sys.stdin); print('strict (unarmed)', round(d['ms_per_step'],4), {k:round(v*1e3,1) for k,v in d['config']['stage_ms'].items()})"
