#!/bin/bash
# Round 7: bin_segments with at most 512 workgroups instead of 256 (two per CU): stage times at configs 3, 4, 5, one view at a time.
cd "$(dirname "$0")/../.."
L=$PWD/diff-gaussian-rasterization_amd/lib
for cfg in config3 config4 config5; do
  for lib in libdgr_hip.so libdgr_hip_wgs512.so; do
    for i in 1 2; do
      DGR_HIP_LIB=$L/$lib python bench.py --workload $cfg --views-in-flight 1 --no-cpu-baseline --steps 30 --warmup 5 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']; print('$cfg $lib', round(d['ms_per_step'],4), {k: round(1e3*v,1) for k,v in s.items() if k.startswith('bin') or k=='preprocess_fwd'})"
    done
  done
done
