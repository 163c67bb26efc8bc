#!/bin/bash
# paired lists + tag bytes in the FULL variant's blend kernels against the commit before: parity, then config 2 (full) stage times with
# the library file swapped (lib_prev/, lib_pair/), alternating
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
PKG=diff-gaussian-rasterization_amd
timeout 1500 python -m pytest tests/test_hip_full_parity.py tests/test_hip_deterministic.py tests/test_hip_appendix_c.py tests/test_hip_random_sweep.py tests/test_hip_heavy_tail.py tests/test_hip_tile_schedule.py tests/test_golden.py tests/test_hip_edge_cases.py tests/test_slam_render.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3 | tee gpurun_out/r9/ab_full_paired.txt
P='import sys,json; d=json.loads(sys.stdin.read()); c=d["config"]; print("ms/view", round(d["ms_per_step"],4), "one", round(c["ms_per_view_one_stream"] or 0,4), "strict", round(c.get("ms_per_view_strict_one_stream") or 0,4), {k: round(v*1e3,1) for k,v in c["stage_ms"].items() if k.startswith("render")})'
run() { python bench.py --no-cpu-baseline --steps 200 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2 3; do for v in lib_prev lib_pair; do
cp $PKG/$v/libdgr_hip.so $PKG/lib/libdgr_hip.so
echo -n "$v config2 full     : "; run --workload config2 --variant full
echo -n "$v config3 full     : "; run --workload config3 --variant full
done; done 2>&1 | tee -a gpurun_out/r9/ab_full_paired.txt
cp $PKG/lib_pair/libdgr_hip.so $PKG/lib/libdgr_hip.so
