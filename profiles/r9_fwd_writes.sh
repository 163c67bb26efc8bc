#!/bin/bash
# What render_fwd's HBM traffic is made of (verdict round 5, item 8: 196 MB per launch against 147 algorithmic): builds of
# render_light.hip with one group of the forward's writes removed each (profiles/variants/ablate_fwd_writes_hooks.patch,
# -DDGR_ABLATE_FWDW: 1 no global atomics of the median statistics, 2 no tag write-back into point_list, 4 no tag bytes per half, 7 none
# of the three) -- wrong results for the backward, right traffic of what is left -- FETCH_SIZE and WRITE_SIZE per launch.
#   gpurun -- 'bash profiles/r9_fwd_writes.sh'
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/r9
R=$PWD
cd /tmp && export TMPDIR=/tmp
for a in "" _fw1 _fw2 _fw4 _fw7; do
  for pass in "FETCH_SIZE TCC_ATOMIC_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum" "WRITE_SIZE TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_ATOMIC_sum"; do
    OUT=$R/gpurun_out/r9/fww_pmc$a; mkdir -p $OUT
    DGR_HIP_LIB=$R/diff-gaussian-rasterization_amd/lib/libdgr_hip$a.so timeout 300 rocprofv3 --pmc $pass -d $OUT/pmc_${pass%% *} -o pmc -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --views-in-flight 1 > $OUT/log 2>&1
  done
  (cd $R && python profiles/summarize.py gpurun_out/r9/fww_pmc$a gpurun_out/r9/fww_pmc$a/sum > /dev/null 2>&1; echo "== lib$a"; grep "render_fwd_light" gpurun_out/r9/fww_pmc$a/sum_pmc.txt) | tee -a $R/gpurun_out/r9/fwd_writes.txt
  rm -rf $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE
done
