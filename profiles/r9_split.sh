#!/bin/bash
# the split binning (bin_tiles places keys in global memory, sort_tile_lists sorts a wave per tile) against the one-kernel form
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
timeout 1200 python -m pytest tests/test_hip_front_end.py tests/test_hip_heavy_tail.py tests/test_hip_light_parity.py tests/test_hip_edge_cases.py tests/test_hip_batch.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -3
for ss in 2 3; do DGR_SEG_SHIFT=$ss timeout 900 python -m pytest tests/test_hip_front_end.py tests/test_hip_heavy_tail.py -x -q -m gpu 2>&1 | grep -v amdgpu.ids | tail -1; done
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("bin") or k.startswith("sort") or k.startswith("tile")})'
for rep in 1 2; do for sp in 1 0; do for scene in synth-v1 clustered heavy_tail; do
  echo -n "split=$sp $scene: "; DGR_BIN_SPLIT=$sp python bench.py --no-cpu-baseline --steps 40 --warmup 10 --scene $scene 2>/dev/null | tail -1 | python -c "$P"
done; done; done 2>&1 | tee gpurun_out/r9/ab_bin_split.txt
for sp in 1 0; do for wl in config2 config4 config5; do echo -n "split=$sp $wl: "; DGR_BIN_SPLIT=$sp python bench.py --no-cpu-baseline --steps 20 --warmup 5 --workload $wl 2>/dev/null | tail -1 | python -c "$P"; done; done 2>&1 | tee -a gpurun_out/r9/ab_bin_split.txt
for ss in 4 3; do echo -n "split=1 clustered DGR_SEG_SHIFT=$ss: "; DGR_SEG_SHIFT=$ss python bench.py --no-cpu-baseline --steps 40 --warmup 10 --scene clustered 2>/dev/null | tail -1 | python -c "$P"; done 2>&1 | tee -a gpurun_out/r9/ab_bin_split.txt
