#!/bin/bash
# Round 7: do the per-Gaussian kernels hide better under the blends when they fit the slots a blend workgroup frees?  Variants of
# preprocess.hip: SH rows staged 16 at a time (LDS 26.9 / 28.2 -> 16.7 / 14.8 KB) and the kernels compiled for 6 / 5 or 8 / 8 waves
# per SIMD (78 / 96 or 64 / 64 VGPRs, with spills).  Each variant takes the place of lib/libdgr_hip.so ON THIS BOX (compiled binding).
cd "$(dirname "$0")/../.."
L=diff-gaussian-rasterization_amd/lib
cp $L/libdgr_hip.so /tmp/libdgr_hip_base.so
for rep in 1 2; do
for v in base r16 r16f8 r16f6b5 r16f8b8; do
  if [ $v = base ]; then cp /tmp/libdgr_hip_base.so $L/libdgr_hip.so; else cp $L/libdgr_hip_$v.so $L/libdgr_hip.so; fi
  python bench.py --no-cpu-baseline --steps 150 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d['config']['stage_ms']; print('$v', 'in flight', round(d['ms_per_step'],4), 'one stream', round(d['config']['ms_per_view_one_stream'],4), {k: round(1e3*v,1) for k,v in s.items() if k.startswith('preprocess')})"
done; done
cp /tmp/libdgr_hip_base.so $L/libdgr_hip.so
