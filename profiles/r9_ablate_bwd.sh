#!/bin/bash
# Where the mapping backward's time goes: builds of render_light.hip with one piece of render_bwd_light_kernel removed each
# (-DDGR_ABLATE_BWD=n: 1 no pair loop, 2 no reduction network and no delivery, 3 the network but no LDS atomics, 4 no exponential)
# -- wrong results, right cost of what is left -- with their vector / scalar / LDS instruction counts.
#   bash profiles/r9_ablate_bwd.sh build ; gpurun -- 'bash profiles/r9_ablate_bwd.sh run'
cd "$(dirname "$0")/../diff-gaussian-rasterization_amd"
# (the hooks are profiles/variants/ablate_bwd_hooks.patch: git apply it first)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize"
if [ "$1" = build ]; then
  for a in 1 2 3 4; do
    hipcc $FLAGS -DDGR_ABLATE_BWD=$a -c csrc/render_light.hip -o build/render_light_babl$a.o
    hipcc --offload-arch=gfx950 -shared -o lib/libdgr_hip_babl$a.so build/api.o build/preprocess.o build/binning.o build/segment_binning.o build/render_light_babl$a.o build/render_full.o build/optim.o build/slam.o
  done
  exit 0
fi
cd ..
mkdir -p gpurun_out/r9
P='import sys,json; d=json.loads(sys.stdin.read()); print({k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render")})'
for rep in 1 2; do
for a in "" _babl1 _babl2 _babl3 _babl4; do
  echo -n "lib$a: "; DGR_HIP_LIB=$PWD/diff-gaussian-rasterization_amd/lib/libdgr_hip$a.so timeout 300 python bench.py --no-cpu-baseline --steps 20 --views-in-flight 1 2>/dev/null | tail -1 | python -c "$P"
done; done 2>&1 | tee gpurun_out/r9/ablate_bwd.txt
cd /tmp && export TMPDIR=/tmp
for a in "" _babl1 _babl2 _babl3; do
  OUT=$GRAFT_REPO_ROOT/gpurun_out/r9/babl_pmc$a; mkdir -p $OUT
  DGR_HIP_LIB=$GRAFT_REPO_ROOT/diff-gaussian-rasterization_amd/lib/libdgr_hip$a.so timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS -d $OUT/pmc_sq1 -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 6 --warmup 2 --no-cpu-baseline --views-in-flight 1 > $OUT/log 2>&1
  (cd $GRAFT_REPO_ROOT && python profiles/summarize.py gpurun_out/r9/babl_pmc$a gpurun_out/r9/babl_pmc$a/sum > /dev/null 2>&1; echo "== lib$a"; grep "render_bwd_light" gpurun_out/r9/babl_pmc$a/sum_pmc.txt) | tee -a $GRAFT_REPO_ROOT/gpurun_out/r9/ablate_bwd.txt
  rm -rf $OUT/pmc_sq1
done
