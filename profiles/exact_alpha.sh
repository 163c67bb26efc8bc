#!/bin/bash
# The exact-alpha measurement build (csrc/render_common.h: DGR_EXACT_ALPHA) and the error budget it answers:
#   bash profiles/exact_alpha.sh build          (here: hipcc cross-compiles)
#   bash profiles/exact_alpha.sh run [P W H]    (on the GPU box; default config 3) -> gpurun_out/error_budget_*.json
cd "$(dirname "$0")/../diff-gaussian-rasterization_amd"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize"
if [ "$1" = build ]; then
  for f in render_light render_light_rows render_full; do hipcc $FLAGS -DDGR_EXACT_ALPHA=1 -c csrc/$f.hip -o build/${f}_exact.o || exit 1; done
  hipcc --offload-arch=gfx950 -shared -o lib/libdgr_hip_exact.so build/api.o build/preprocess.o build/binning.o build/render_light_exact.o build/render_light_rows_exact.o build/render_full_exact.o build/optim.o build/slam.o
  exit $?
fi
cd ..
shift
ARGS=${*:-500000 1920 1080}
TAG=$(echo $ARGS | tr ' ' '_')
mkdir -p gpurun_out
python tests/tools/error_budget.py $ARGS 2>/dev/null | tail -1 > gpurun_out/error_budget_fast_$TAG.json
DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip_exact.so python tests/tools/error_budget.py $ARGS 2>/dev/null | tail -1 > gpurun_out/error_budget_exact_$TAG.json
python - <<PY
import json
for v in ("fast", "exact"):
    d = json.load(open("gpurun_out/error_budget_%s_$TAG.json" % v))
    print(v, "integer path exact:", d["integer_path_exact"], "n_contrib mismatches:", d["n_contrib_mismatch"],
          "colour > 1e-5:", "%.1e" % d["img_color"]["frac_over_1e-5"])
    for lab in ("end_to_end", "isolated"):
        print("  ", lab, {k: "%.1e/%.1e" % (x["max_abs"], x["scale"]) for k, x in d[lab].items()})
PY
