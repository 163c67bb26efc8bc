#!/bin/bash
# the forward's lane-private bit-mask walk (DGR_FWD_WALK=1) against the half-wave lists: parity, then stage times alternating
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
DGR_FWD_WALK=1 timeout 900 python -m pytest tests/test_hip_light_parity.py tests/test_golden.py tests/test_hip_heavy_tail.py -x -q -m gpu -k "not config4 and not config5" 2>&1 | grep -v amdgpu.ids | tail -3
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render")})'
for rep in 1 2; do for w in 0 1; do for scene in synth-v1 clustered heavy_tail; do
  echo -n "walk=$w $scene: "; DGR_FWD_WALK=$w python bench.py --no-cpu-baseline --steps 60 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done; done; done 2>&1 | tee gpurun_out/r9/ab_fwd_walk.txt
for w in 0 1; do echo -n "walk=$w config2 light: "; DGR_FWD_WALK=$w python bench.py --no-cpu-baseline --steps 60 --workload config2 2>/dev/null | tail -1 | python -c "$P"; done | tee -a gpurun_out/r9/ab_fwd_walk.txt
