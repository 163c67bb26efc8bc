#!/bin/bash
# Where the blend kernels' time goes: builds of render_light.hip with the pair loop removed (DGR_ABLATE=1) and with the
# backward's butterfly removed (DGR_ABLATE=2) -- wrong results, right cost of what is left.  Build here (hipcc cross-
# compiles), run on the GPU box:  bash profiles/ablate.sh build ; gpurun -- 'bash profiles/ablate.sh run'
cd "$(dirname "$0")/../diff-gaussian-rasterization_amd"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize"
if [ "$1" = build ]; then
  for a in 1 2 3; do
    hipcc $FLAGS -DDGR_ABLATE=$a -c csrc/render_light.hip -o build/render_light_ablate$a.o
    hipcc $FLAGS -DDGR_ABLATE=$a -c csrc/render_light_rows.hip -o build/render_light_rows_ablate$a.o
    hipcc --offload-arch=gfx950 -shared -o lib/libdgr_hip_ablate$a.so build/api.o build/preprocess.o build/binning.o build/render_light_ablate$a.o build/render_light_rows_ablate$a.o build/render_full.o build/optim.o build/slam.o
  done
  exit 0
fi
cd ..
P='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render")})'
for a in "" _ablate1 _ablate2 _ablate3; do
  echo "lib$a:"; DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip$a.so python bench.py --no-cpu-baseline --steps 20 --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P"
done
