#!/bin/bash
# Render records at a 64-byte stride (DGR_REC_STRIDE = 4: one L2 line per gather) against 48 bytes (3: 1.5 lines): HBM traffic of the
# blend kernels and of preprocess_fwd per launch, then stage times alternating.  The library file is swapped (lib_s3/, lib_s4/:
# builds of this tree with -DDGR_REC_STRIDE=3 / 4).   gpurun -- 'bash profiles/r9_rec_stride.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
R=$PWD; PKG=diff-gaussian-rasterization_amd
rm -f $R/gpurun_out/r9/rec_stride.txt
python -m pytest tests/test_hip_light_parity.py tests/test_hip_full_parity.py tests/test_golden.py tests/test_hip_front_end.py -x -q -m gpu -k "not config4 and not config5" 2>&1 | tail -2 | tee -a $R/gpurun_out/r9/rec_stride.txt
cd /tmp && export TMPDIR=/tmp
for v in s3 s4; do
  cp $R/$PKG/lib_$v/libdgr_hip.so $R/$PKG/lib/libdgr_hip.so
  for pass in "FETCH_SIZE TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
    OUT=$R/gpurun_out/r9/recs_pmc_$v; mkdir -p $OUT
    timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_${pass%% *} -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --views-in-flight 1 > $OUT/log 2>&1
  done
  (cd $R && python profiles/summarize.py gpurun_out/r9/recs_pmc_$v gpurun_out/r9/recs_pmc_$v/sum > /dev/null 2>&1; echo "== stride $v"; grep -E "render_(fwd|bwd)_light|preprocess_(fwd|bwd)" gpurun_out/r9/recs_pmc_$v/sum_pmc.txt | grep -E "FETCH_SIZE|WRITE_SIZE|TCC_HIT|TCC_MISS") | tee -a $R/gpurun_out/r9/rec_stride.txt
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
done
cd $R
P='import sys,json; d=json.loads(sys.stdin.read()); print("ms/view", round(d["ms_per_step"],4), "serial", round(d["config"]["ms_per_view_one_stream"] or 0,4), "strict", round(d["config"].get("ms_per_view_strict_one_stream") or 0,4), {k: round(v*1e3,1) for k,v in d["config"]["stage_ms"].items() if k.startswith("render") or k.startswith("pre")})'
run() { python bench.py --no-cpu-baseline --steps 100 "$@" 2>/dev/null | tail -1 | python -c "$P"; }
for rep in 1 2 3; do for v in s3 s4; do
cp $PKG/lib_$v/libdgr_hip.so $PKG/lib/libdgr_hip.so
echo -n "$v mapping  : "; run
echo -n "$v tracking : "; run --tracking
done; done 2>&1 | tee -a gpurun_out/r9/rec_stride.txt
cp $PKG/lib_s4/libdgr_hip.so $PKG/lib/libdgr_hip.so
