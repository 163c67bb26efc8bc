#!/bin/bash
# lane lists by the frame (option "lane_lists" = 2, decided by bin_tiles on the device) -- parity, then stage times against the
# forced mappings and against the library of the commit before (lib_head/: separate kernels per mapping, DGR_FWD_HALVES per process)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
PKG=diff-gaussian-rasterization_amd
timeout 1500 python -m pytest tests/test_hip_lane_mappings.py tests/test_hip_light_parity.py tests/test_hip_front_end.py tests/test_golden.py tests/test_hip_heavy_tail.py tests/test_hip_deterministic.py -x -q -m gpu -k "not config4 and not config5" 2>&1 | grep -v amdgpu.ids | tail -4 | tee gpurun_out/r9/lists_pytest.txt
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render") or k.startswith("bin_tiles")})'
run() { python bench.py --no-cpu-baseline --steps 60 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2; do for scene in synth-v1 heavy_tail clustered; do
  echo -n "head  halves=1 $scene: "; LD_LIBRARY_PATH=$PWD/$PKG/lib_head DGR_HIP_LIB=$PWD/$PKG/lib_head/libdgr_hip.so DGR_FWD_HALVES=1 run --scene $scene
  echo -n "head  halves=0 $scene: "; LD_LIBRARY_PATH=$PWD/$PKG/lib_head DGR_HIP_LIB=$PWD/$PKG/lib_head/libdgr_hip.so DGR_FWD_HALVES=0 run --scene $scene
  echo -n "fused lists=1  $scene: "; DGR_FWD_HALVES=1 run --scene $scene
  echo -n "fused lists=0  $scene: "; DGR_FWD_HALVES=0 run --scene $scene
  echo -n "fused by frame $scene: "; run --scene $scene
done; done 2>&1 | tee gpurun_out/r9/ab_lists.txt
echo -n "tracking step, head: "; LD_LIBRARY_PATH=$PWD/$PKG/lib_head DGR_HIP_LIB=$PWD/$PKG/lib_head/libdgr_hip.so run --tracking 2>&1 | tee -a gpurun_out/r9/ab_lists.txt
echo -n "tracking step, fused: "; run --tracking 2>&1 | tee -a gpurun_out/r9/ab_lists.txt
