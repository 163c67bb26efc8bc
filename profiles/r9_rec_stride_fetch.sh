#!/bin/bash
# the FETCH_SIZE half of profiles/r9_rec_stride.sh (its first pass asked for too many TCC counters at once and timed out)
cd "$(dirname "$0")/.."
R=$PWD; PKG=diff-gaussian-rasterization_amd
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for v in s3 s4; do
  cp $R/$PKG/lib_$v/libdgr_hip.so $R/$PKG/lib/libdgr_hip.so
  OUT=$R/gpurun_out/r9/recf_pmc_$v; rm -rf $OUT; mkdir -p $OUT
  timeout 150 rocprofv3 --pmc FETCH_SIZE TCC_EA0_RDREQ_sum TCC_ATOMIC_sum -d $OUT/pmc_FETCH_SIZE -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --views-in-flight 1 > $OUT/log 2>&1
  (cd $R && python profiles/summarize.py gpurun_out/r9/recf_pmc_$v gpurun_out/r9/recf_pmc_$v/sum > /dev/null 2>&1; echo "== stride $v (run $rep)"; grep -E "render_(fwd|bwd)_light|preprocess_(fwd|bwd)" gpurun_out/r9/recf_pmc_$v/sum_pmc.txt | grep -E "FETCH_SIZE") | tee -a $R/gpurun_out/r9/rec_stride.txt
  rm -rf $OUT/pmc_FETCH_SIZE
done; done
cp $R/$PKG/lib_s4/libdgr_hip.so $R/$PKG/lib/libdgr_hip.so
