#!/bin/bash
# fused kernels (one per mode, the frame's flag picks the lane lists) against the commit before, through the compiled binding:
# the library file itself is swapped (lib_head/ = commit c6cbd1b's build, lib_new/ = this tree's)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
PKG=diff-gaussian-rasterization_amd
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), "strict", round(d["config"].get("ms_per_view_strict_one_stream") or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render") or k.startswith("bin_tiles")})'
run() { python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2; do for lib in head new; do
cp $PKG/lib_$lib/libdgr_hip.so $PKG/lib/libdgr_hip.so
echo -n "$lib tracking   : "; run --tracking
echo -n "$lib mapping    : "; run
echo -n "$lib heavy_tail : "; run --scene heavy_tail
done; done 2>&1 | tee gpurun_out/r9/ab_lists_fused.txt
cp $PKG/lib_new/libdgr_hip.so $PKG/lib/libdgr_hip.so
