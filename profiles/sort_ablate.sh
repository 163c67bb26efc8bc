#!/bin/bash
# Where sort_tiles' 30 us go: builds without the register compare-exchange steps (DGR_ABLATE_SORT=1) and without the LDS merge
# stages as well (=2) -- wrong order, right cost of what is left.  bash profiles/sort_ablate.sh build ; gpurun -- 'bash profiles/sort_ablate.sh run'
cd "$(dirname "$0")/../diff-gaussian-rasterization_amd"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize"
if [ "$1" = build ]; then
  for a in 1 2; do
    hipcc $FLAGS -DDGR_ABLATE_SORT=$a -c csrc/binning.hip -o build/binning_sortablate$a.o
    hipcc --offload-arch=gfx950 -shared -o lib/libdgr_hip_sortablate$a.so build/api.o build/preprocess.o build/binning_sortablate$a.o build/render_light.o build/render_light_rows.o build/render_full.o build/optim.o build/slam.o
  done
  exit 0
fi
cd ..
P='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k in ("sort_tiles","emit_instances")})'
for a in "" _sortablate1 _sortablate2; do
  echo "lib$a:"; DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip$a.so python bench.py --no-cpu-baseline --steps 20 --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P"
done
